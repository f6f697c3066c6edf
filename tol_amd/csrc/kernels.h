// kernels.h -- launch interface of the gfx950 kernels (host side; no HIP types leak past capi).
#ifndef TOLFG_KERNELS_H_
#define TOLFG_KERNELS_H_

#include <hip/hip_runtime_api.h>

namespace tolfg {

enum { MISSION_S10 = 0, MISSION_G7 = 1 };
enum { WIND_NONE = 0, WIND_SHEAR = 1, WIND_TABLE = 2 };   // kernel-side enumeration
enum { MAX_AIRCRAFT = 8 };
enum { PATTERN_REFERENCE = 0, PATTERN_COMPACT = 1 };   // slab_table.h

// Air-frame constants as the kernels want them (reciprocals taken once on the host).
// ref: the members of `aircraft` the path reads, include/parameters.h:25-30, and g/rho,
// include/problem.h:72-73.
struct AcCoef {
    double inv_m;      // 1/mm
    double qk;         // rho*SS/(2*mm)        so that q = qk*Va^2
    double Cd0;
    double kind;       // 1/(AR*pi*ee)         induced-drag factor
};

// Per-trajectory constants, one record per trajectory in device memory.
struct TrajDev {
    double shear;      // Vref/href: dWx/dz of the linear boundary layer (src/problem.cpp:521-524)
    double xg, yg, rg; // goal, NED (src/problem.cpp:24-27)
    double cchi, schi; // cos/sin of G7's chi_d (src/problemG7.cpp:524)
    int    ac;         // index into FgArgs::ac
    int    pad;
};

struct FgArgs {
    const void    *X;      long ldx;
    void          *F;      long ldf;
    void          *G;      long ldg;
    const void    *wind;   // [B][12][N+1] (ENU, reference member order) or nullptr
    const TrajDev *traj;   // [B]
    int  B, N, c0;
    int  tiles, nt;        // tiles per trajectory and nodes per tile, from plan_tiles()
    int  needF, needG;
    int  pattern;          // PATTERN_REFERENCE (104-entry slabs) | PATTERN_COMPACT (46-entry slabs)
    double *partial;       // [B*tiles][2] objective partials (sum T^2, sum (r-R)^2), device
    void   *obj;           // optional [B]: finalize_kernel also writes the objectives here, contiguous
    double kT, kp, kv, kdt;
    AcCoef ac[MAX_AIRCRAFT];
#ifdef TOLFG_STAMPS
    // diagnostic build only (tools/fgprobe.cpp): per-wave s_memtime stamps, 8 per workgroup, and a
    // variant selector for ablations.  The product library is never compiled with TOLFG_STAMPS.
    unsigned long long *stamps;
    int variant;
#endif
};

// Tiling of one trajectory's N dynamic nodes: `tiles` tiles of `nt` nodes (the last may be short).
void plan_tiles(int N, int dtype, int *tiles, int *nt);

// One evaluation = fg_kernel + finalize_kernel on stream s: F and G of B trajectories.  dtype: 0 = f64, 1 = f32.  vec = elements per
// 16-byte access the caller has verified alignment for (f64: 2 or 1; f32: 4 or 1).
// t0/t1 (may be null) are recorded on s immediately around fg_kernel, the dominant kernel.
hipError_t launch_fg(const FgArgs &a, int mission, int wind, int dtype, int vec, hipStream_t s,
                     hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);

// dObj[t] = F[t*ldf]
hipError_t launch_objectives(const void *F, long ldf, void *obj, int B, int dtype, hipStream_t s);

// LDS bytes per workgroup of the fg kernel (for DESIGN.md / occupancy reporting)
int fg_lds_bytes(int dtype);

}  // namespace tolfg
#endif
