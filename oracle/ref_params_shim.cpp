// ref_params_shim.cpp -- TEST INFRASTRUCTURE ONLY (see tolfg_oracle.h).
//
// A C-ABI window onto the REFERENCE's own .param reader, so tests can check this repo's readers
// (oracle/tolfg_oracle.c orc_read_params, tol_amd/csrc/params.cpp) against it.  The reference
// translation unit /root/reference/src/parameters.cpp depends on nothing but the C++ standard
// library, so `make ref` compiles it from where it lies together with this shim into
// oracle/_ref/libref_params.so (git-ignored; never copied, never shipped as source).
// Nothing else of the reference's hot path can be built here: every other translation unit includes
// include/problem.h, which needs the MongoDB C++ client headers (absent) -- see DESIGN.md.
#include "parameters.h"   // -I/root/reference/include

#include <exception>
#include <sstream>
#include <string>

namespace {
// the reference's constructors print progress to std::cout; keep test output quiet
struct quiet_cout {
    std::streambuf *old;
    std::ostringstream sink;
    quiet_cout() : old(std::cout.rdbuf(sink.rdbuf())) {}
    ~quiet_cout() { std::cout.rdbuf(old); }
};
}

extern "C" {

// out[15] = mm b SS ee AR Cd0 CLmin CLmax phimax Vamin Vamax gammamax phidotmax Tmin Tmax
// (angles already converted to radians by the reference, src/parameters.cpp:56-60).
// Returns 0, or -1 when the reference throws (wrong element count, src/parameters.cpp:65).
int ref_aircraft(const char *name, const char *root, double *out)
{
    quiet_cout q;
    try {
        aircraft a(name, root);
        const double v[15] = {a.mm, a.b, a.SS, a.ee, a.AR, a.Cd0, a.CLmin, a.CLmax, a.phimax,
                              a.Vamin, a.Vamax, a.gammamax, a.phidotmax, a.Tmin, a.Tmax};
        for (int i = 0; i < 15; i++) out[i] = v[i];
        return 0;
    } catch (std::exception &) { return -1; }
}

int ref_gain(const char *mission, const char *root, double *out)   // kT kp kv ka kdt
{
    quiet_cout q;
    try {
        gain g(mission, root);
        out[0] = g.kT; out[1] = g.kp; out[2] = g.kv; out[3] = g.ka; out[4] = g.kdt;
        return 0;
    } catch (std::exception &) { return -1; }
}

int ref_limit(const char *mission, const char *root, double *out)  // dtmin dtmax xmin xmax ymin ymax zmin zmax
{
    quiet_cout q;
    try {
        limit l(mission, root);
        out[0] = l.dtmin; out[1] = l.dtmax; out[2] = l.xmin; out[3] = l.xmax;
        out[4] = l.ymin; out[5] = l.ymax; out[6] = l.zmin; out[7] = l.zmax;
        return 0;
    } catch (std::exception &) { return -1; }
}

int ref_snopt(const char *mission, const char *root, double *out)  // ts numinp numstates numbounds opt_tol feas_tol
{
    quiet_cout q;
    try {
        snopt s(mission, root);
        out[0] = s.ts; out[1] = s.numinp; out[2] = s.numstates; out[3] = s.numbounds;
        out[4] = s.opt_tol; out[5] = s.feas_tol;
        return 0;
    } catch (std::exception &) { return -1; }
}

}
