// params.cpp -- .param reader and the four typed bundles (see params.h for the reference map).
#include "params.h"

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <stdexcept>

namespace tolfg {

bool readparams(const std::string &filepath, std::vector<double> &out)
{
    out.clear();
    std::ifstream in(filepath);
    if (!in.is_open()) return false;
    std::string line;
    while (std::getline(in, line)) {
        const std::string head = line.substr(0, line.find('/'));
        const char *c = head.c_str();
        char *end = nullptr;
        errno = 0;
        const double v = std::strtod(c, &end);
        if (end == c || errno == ERANGE) continue;   // what std::stod rejects, the reference skips
        out.push_back(v);
    }
    return true;
}

namespace {
std::vector<double> exactly(const std::string &path, size_t want, const std::string &what)
{
    std::vector<double> v;
    readparams(path, v);   // an unreadable file yields zero values, as in the reference
    if (v.size() != want) throw std::length_error("Wrong number of parameters for " + what + ".param");
    return v;
}
std::string join(const std::string &root, const std::string &rel)
{
    if (root.empty() || root.back() == '/') return root + rel;
    return root + "/" + rel;
}
}  // namespace

aircraft::aircraft(const std::string &name, const std::string &root)
{
    const std::vector<double> p = exactly(join(root, "aircraft/" + name + ".param"), 15, name);
    const double d2r = M_PI / 180.0;
    mm = p[0]; b = p[1]; SS = p[2]; ee = p[3]; AR = p[4]; Cd0 = p[5]; CLmin = p[6]; CLmax = p[7];
    phimax = p[8] * d2r; Vamin = p[9]; Vamax = p[10]; gammamax = p[11] * d2r; phidotmax = p[12] * d2r;
    Tmin = p[13]; Tmax = p[14];
}

gain::gain(const std::string &mission, const std::string &root)
{
    const std::vector<double> p = exactly(join(root, "problems/" + mission + "/gains.param"), 5, mission);
    kT = p[0]; kp = p[1]; kv = p[2]; ka = p[3]; kdt = p[4];
}

limit::limit(const std::string &mission, const std::string &root)
{
    const std::vector<double> p = exactly(join(root, "problems/" + mission + "/limits.param"), 8, mission);
    dtmin = p[0]; dtmax = p[1]; xmin = p[2]; xmax = p[3]; ymin = p[4]; ymax = p[5]; zmin = p[6]; zmax = p[7];
}

snopt::snopt(const std::string &mission, const std::string &root)
{
    const std::vector<double> p = exactly(join(root, "problems/" + mission + "/snopt.param"), 6, mission);
    ts = (int)p[0]; numinp = (int)p[1]; numstates = (int)p[2]; numbounds = (int)p[3];
    opt_tol = p[4]; feas_tol = p[5];
}

}  // namespace tolfg
