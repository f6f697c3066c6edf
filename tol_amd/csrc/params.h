// params.h -- the four .param bundles of tol, host side.
// Mirrors the reference's types by name and field (ref: include/parameters.h:22-74) so that code
// written against the reference reads the same; the implementation is this repo's own.
#ifndef TOLFG_PARAMS_H_
#define TOLFG_PARAMS_H_

#include <string>
#include <vector>

namespace tolfg {

// ref: parameters::readparams, src/parameters.cpp:14-34.  One number per line; the text from the
// first '/' on is ignored; the leading float of the rest is taken (trailing junk such as the
// literal backslash-n and CRs of the shipped files is ignored); lines without one are skipped.
// Returns false when the file cannot be opened.
bool readparams(const std::string &filepath, std::vector<double> &out);

// ref: class aircraft, include/parameters.h:22-40; src/parameters.cpp:42-69.
// Throws std::length_error unless the file holds exactly 15 values.  Angles are converted from
// degrees to radians like the reference does (src/parameters.cpp:56,59,60).
struct aircraft {
    aircraft(const std::string &aircraftname, const std::string &root_path);
    double mm, b, SS, ee, AR, Cd0, CLmin, CLmax, phimax, Vamin, Vamax, gammamax, phidotmax, Tmin, Tmax;
};

// ref: class gain, include/parameters.h:42-50; 5 values
struct gain {
    gain(const std::string &problemtype, const std::string &root_path);
    double kT, kp, kv, ka, kdt;
};

// ref: class limit, include/parameters.h:52-63; 8 values
struct limit {
    limit(const std::string &problemtype, const std::string &root_path);
    double dtmin, dtmax, xmax, ymax, zmax, xmin, ymin, zmin;
};

// ref: class snopt, include/parameters.h:65-74; 6 values
struct snopt {
    snopt(const std::string &problemtype, const std::string &root_path);
    int ts, numinp, numstates, numbounds;
    double opt_tol, feas_tol;
};

}  // namespace tolfg
#endif
