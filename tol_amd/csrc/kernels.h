// kernels.h -- launch interface of the gfx950 kernels (host side; no HIP types leak past capi).
#ifndef TOLFG_KERNELS_H_
#define TOLFG_KERNELS_H_

#include <hip/hip_runtime_api.h>

namespace tolfg {

enum { MISSION_S10 = 0, MISSION_G7 = 1, MISSION_MIXED = 2 };   // MIXED: every trajectory carries its own (TrajDev::mission)
enum { WIND_NONE = 0, WIND_SHEAR = 1, WIND_TABLE = 2, WIND_GRID = 3 };   // kernel-side enumeration
enum { MAX_AIRCRAFT = 8 };
enum { PATTERN_REFERENCE = 0, PATTERN_COMPACT = 1 };   // slab_table.h
constexpr int kTileNodes = 64;                          // nodes per one-node-per-lane tile = wavefront width (kernels.hip TILE, plan.cpp)
constexpr unsigned kEmptySlotWord = 0xFFFBADADu;        // fused path: every 32-bit word of an empty partial slot

// Air-frame constants as the kernels want them (reciprocals taken once on the host).
// ref: the members of `aircraft` the path reads, include/parameters.h:25-30, and g/rho,
// include/problem.h:72-73.
struct AcCoef {
    double inv_m;      // 1/mm
    double qk;         // rho*SS/(2*mm)        so that q = qk*Va^2
    double Cd0;
    double kind;       // 1/(AR*pi*ee)         induced-drag factor
    double mm, SS, AR, ee;   // as read, for the initial-guess kernel (same operations as the host code)
};

// Bounds of the nodes k >= 1 and the step (ref: problem::setLimits, src/problem.cpp:272-285,267);
// lo/up in node-variable order x y z Va gam chi phi CL dphi dCL T.
struct AcBounds {
    double lo[11], up[11];
};

// Per-trajectory constants, one record per trajectory in device memory.
struct TrajDev {
    double shear;      // Vref/href: dWx/dz of the linear boundary layer (src/problem.cpp:521-524)
    double xg, yg, rg; // goal, NED (src/problem.cpp:24-27)
    double cchi, schi; // cos/sin of G7's chi_d (src/problemG7.cpp:524)
    int    ac;         // index into FgArgs::ac
    int    mission;    // MISSION_S10 | MISSION_G7 of this trajectory (read by MISSION_MIXED launches only)
    double xi, yi, zi; // start position (src/problem.cpp:83-85), for the initial-guess and bounds kernels
    double chi_d;
};

// Wind model 3, the gridded storm field (ref: problem::modelWind case 3, src/problem.cpp:544-695):
// regular ENU grid of the v (north) component -- the only one the reference interpolates.
struct GridDev {
    const void *v;             // [nx][ny][nz] elements of the batch dtype, device
    int    nx, ny, nz, pad;
    double x0, y0, z0;         // ENU coordinates of grid point (0,0,0)
    double dx, dy, dz;         // spacing (150 m in the reference, include/problem.h:90-92)
    double e0, n0, u0;         // EastFromDatum, NorthFromDatum, UpFromDatum (src/problem.cpp:411-413)
};

struct FgArgs {
    const void    *X;      long ldx;
    void          *F;      long ldf;
    void          *G;      long ldg;
    const void    *wind;   // [B][12][N+1] (ENU, reference member order) or nullptr
    GridDev        grid;   // WIND_GRID only
    const TrajDev *traj;   // [B]
    int  B, N;
    int  c0[2];            // position in G of node 0's slab, per mission
    int  tiles, nt;        // tiles per trajectory and nodes per tile, from plan_tiles()
    // Finer tiles for the trajectories the launch reaches last: the last tail_count trajectories (0 = none)
    // are cut into tail_tiles tiles of tail_nt <= nt nodes; their workgroup ids follow those of the body, so the
    // waves the machine runs while it drains are short-lived.  Tile numbering, with tb = B - tail_count: body
    // tile t of trajectory b < tb is b * tiles + t; tail tile t of b >= tb is tb * tiles + (b - tb) * tail_tiles + t.
    int  tail_count, tail_tiles, tail_nt;
    int  needF, needG;
    int  pattern;          // PATTERN_REFERENCE (104-entry slabs) | PATTERN_COMPACT (46-entry slabs)
    int  waves_per_cu;     // cap on resident tile waves per CU (0 = whatever fits); host-side launch hint
    int  single;           // 1 = one workgroup per trajectory, one launch (tiles <= 8, small B)
    int  fused;            // 1 = the tile wave that arrives last at its trajectory's counter finalizes (one launch);
                           // 0 = finalize_kernel follows fg_kernel
    int  stagger;          // 1 = waves take an issue priority from their slot on the SIMD (launches whose waves all start together)
    int  sub_nodes;        // 0 = a tile's Jacobian rows go through LDS all at once; 32 = in passes of 32 nodes (less LDS per
                           // wave: 16 instead of 10 resident fp64 tile waves per CU); fp64, tile-per-workgroup kernel only
    int  nt_stores;        // 1 = the slab stream carries the non-temporal hint (outputs beyond the Infinity Cache)
    int  store_shape;      // measurement aid: 1 = launch store_shape_kernel (this launch's grid, tile order, LDS request and
                           // store flavour around nothing but the slab stores) instead of the evaluation
    int  xcd_chunk;        // workgroup id < 8 * xcd_chunk -> tile (id % 8) * xcd_chunk + id / 8 (every XCD walks a contiguous
                           // run of xcd_chunk tiles); id >= 8 * xcd_chunk -> tile id.  0 <= xcd_chunk <= ceil(B*tiles/8)
    double *partial;       // [B*tiles][2] objective partials (sum T^2, sum (r-R)^2), device; on the fused path
                           // every slot is "empty" (kEmptySlotWord) between launches
    unsigned *counter;     // [B + 1]: [0, B) arrival counters of the fused path, [B] departures of the callback's
                           // completion word; all zero before a launch, put back to zero by the launch itself
    unsigned *status;      // host-mapped word (or nullptr): set to 1 by a finalizing wave that gave up waiting for an objective
                           // partial (fused path; only an input that carries the empty-slot marker itself can cause that)
    // Completion word for the SNOPT callback (host-mapped, or nullptr): when every wave's stores are visible
    // to the host, the last wave to leave writes done_seq there, so the caller can spin on it instead of
    // synchronising the stream.
    unsigned long long *done;
    unsigned long long done_seq;
    void   *obj;           // optional [B]: finalize_kernel also writes the objectives here, contiguous
    double kT[2], kp[2], kv[2], kdt[2];   // gains, per mission (problems/<mission>/gains.param)
    AcCoef ac[MAX_AIRCRAFT];
#if defined(TOLFG_STAMPS) || defined(TOLFG_ABLATE)
    // diagnostic builds only (tools/fgprobe.cpp, tools/fgbench.cpp -DTOLFG_ABLATE): per-wave s_memtime stamps, 8 per
    // workgroup, and a variant selector for ablations.  The product library is never compiled with either macro.
    unsigned long long *stamps;
    int variant;
#endif
};

// Tiling of one trajectory's N dynamic nodes: `tiles` tiles of `nt` <= max_nt nodes (the last may be
// short).  max_nt is a multiple of 4 in [4, 64] (fp32: up to 128, two nodes per lane); 0 = 64.
void plan_tiles(int N, int dtype, int max_nt, int *tiles, int *nt);
// Per-launch choices the host makes from the size of the outputs (plan.cpp holds the measurements)
struct LaunchPlan {
    int max_nt;            // upper bound on nodes per tile handed to plan_tiles()
    int waves_per_cu;      // cap on resident tile waves per CU, 0 = none
    int nt_stores;         // non-temporal slab stream
    int xcd;               // deal the tiles to the XCDs in contiguous eighths
    int fused;             // the last-arriving tile wave finalizes (one launch) vs finalize_kernel as a second launch
    int tail_count, tail_nt;   // the last tail_count trajectories in tiles of <= tail_nt nodes (0 = no tail)
    int stagger;           // issue priorities by SIMD slot (FgArgs::stagger)
    int sub_nodes;         // Jacobian rows through LDS in passes of this many nodes (FgArgs::sub_nodes), 0 = whole tile
    int single;            // one workgroup per trajectory, one launch (fg_single_kernel): a few short trajectories
};
// what the plan looks at: the launch's shape and the bytes it writes
struct LaunchShape {
    int B, N;              // trajectories, dynamic nodes per trajectory
    int dtype, pattern;    // 0 = f64 | 1 = f32; PATTERN_*
    int mission;           // MISSION_*
    int aligned;           // rows of X and F on 16-byte boundaries (16-byte window loads / defect stores possible)
    double out_bytes;      // F + G bytes this launch writes
    int needG = 1;         // the Jacobian is wanted (0: F alone)
    int cus = 256;         // compute units of the device (MI355X: 256; the batch asks the runtime)
};
LaunchPlan plan_launch(const LaunchShape &shape);

// One evaluation on stream s: F and G of B trajectories.  dtype: 0 = f64, 1 = f32.  vec = elements per
// 16-byte access the caller has verified alignment of X, F and their row strides for (f64: 2 or 1;
// f32: 4 or 1).  The Jacobian slabs are always streamed with 16-byte stores: wherever a slab region sits
// relative to a 16-byte boundary (odd c0, or c0 % 4 == 2 in fp32), the wave shifts its stream.
// t0/t1 (may be null) are recorded on s around the whole evaluation.
hipError_t launch_fg(const FgArgs &a, int mission, int wind, int dtype, int vec, hipStream_t s,
                     hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);

// dObj[t] = F[t*ldf]
hipError_t launch_objectives(const void *F, long ldf, void *obj, int B, int dtype, hipStream_t s);

// out[0] = sum of the B contiguous values at v, in double (v: elements of dtype)
hipError_t launch_sum(const void *v, int B, int dtype, double *out, hipStream_t s);

// Initial guess of B trajectories straight into device rows (ref: problemS10::InitialCond /
// problemG7::InitialCond); bounds likewise (ref: problem::setLimits).  One-time set-up kernels.
// tab: device table [2 missions][X0_FIELDS][N+1].  Field 0 of either mission holds the node times t_k = t_(k-1) + dt,
// added up on the host (S10: dt = 20/N, G7: dt = 10/N); launch_x0_table fills the other fields from them once (everything
// of a node's row that is the same for all trajectories of a mission); launch_x0 then runs the node-parallel kernel, one
// workgroup per trajectory.  tab == nullptr = the serial reference form, one thread per trajectory (bitwise the same rows).
constexpr int X0_FIELDS = 10;
hipError_t launch_x0_table(double *tab, int N, int mission, hipStream_t s);
hipError_t launch_x0(const FgArgs &a, int mission, int dtype, const double *tab, hipStream_t s);
struct BoundsArgs {
    void *xlow, *xupp; long ldx;
    void *Flow, *Fupp; long ldf;
    const TrajDev *traj;
    int B, N, mission;     // mission may be MISSION_MIXED; rows of a mixed batch are sized for the larger mission
    double dtmin[2], dtmax[2];          // per mission (problems/<mission>/limits.param)
    AcBounds ac[2][MAX_AIRCRAFT];       // [mission][air-frame]
};
hipError_t launch_bounds(const BoundsArgs &a, int dtype, hipStream_t s);

// *count += the number of 32-bit words in [p, p + bytes) that differ from `pattern` (alloc_kernels.hip; device_alloc's settle check)
hipError_t launch_count_not(const void *p, size_t bytes, unsigned pattern, unsigned long long *count, hipStream_t s);

// LDS bytes per workgroup of the fg kernel (for DESIGN.md / occupancy reporting)
int fg_lds_bytes(int dtype, int nt = 0, int sub_nodes = 0);        // nt = nodes per tile (0 = 64); sub_nodes: FgArgs::sub_nodes
// LDS bytes to request at launch so that at most waves_per_cu workgroups share a CU (0 = no cap)
int fg_lds_request(int dtype, int waves_per_cu, int nt = 0, int sub_nodes = 0);

}  // namespace tolfg
#endif
