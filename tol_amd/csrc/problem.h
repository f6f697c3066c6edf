// problem.h -- host side of the user-function path: GPU-backed counterparts of the reference's
// `problem` / `problemS10` / `problemG7` (ref: include/problem.h, include/problemS10.h,
// include/problemG7.h) and a batched evaluator with no reference counterpart.
#ifndef TOLFG_PROBLEM_H_
#define TOLFG_PROBLEM_H_

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/tolfg.h"
#include "kernels.h"
#include "params.h"
#include "setup.h"

namespace tolfg {

struct hip_failure : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// throws std::invalid_argument (ref: src/tol.cpp:19-23); "mixed" only where a batch may mix missions
int mission_from_name(const std::string &name, bool allow_mixed = false);
std::string default_root();

// Device memory as ONE address range backed by 2 MiB physical chunks (HIP virtual-memory management: hipMemAddressReserve,
// hipMemCreate + hipMemMap per chunk).  Why: where a launch's output buffer lands in HBM decides whether the kernel's ~2000
// concurrent store fronts run at 4.7 or at 6 TB/s (profiles/r04_allocation_classes.md) -- a plain hipMalloc of 1.4 GB lands in
// the slow class most of the time, this form in the fast one most of the time, and batch::alloc_outputs times candidates and
// keeps the best.  Falls back to hipMalloc where the runtime has no virtual-memory support.  Thread-safe registry.
void *device_alloc(int device, size_t bytes);
void device_free(void *ptr);           // no-op for nullptr; throws for a pointer device_alloc did not return

// ---------------------------------------------------------------------------------------------
// Device-resident evaluation of B independent trajectories that share ts; mission "S10", "G7" or
// "mixed" (every trajectory names its own mission; rows are sized for the larger one).
class batch {
public:
    batch(const std::string &mission, const std::string &root, const std::vector<std::string> &aircraft_names,
          int ts, int windmodel, int dtype, int device, int pattern = 0);
    ~batch();
    batch(const batch &) = delete;
    batch &operator=(const batch &) = delete;

    // row sizes to allocate for: the mission's own, or (mixed) the larger of the two per quantity
    const Sizes &sizes() const { return sz_; }
    // sizes and pattern of one mission of this batch (MISSION_S10 | MISSION_G7)
    const Sizes &sizes_of(int mission) const;
    const Sizes &sizes_of_traj(int t) const { return sizes_of(mission_of_traj(t)); }
    int mission() const { return mission_; }          // MISSION_S10 | MISSION_G7 | MISSION_MIXED
    int mission_of_traj(int t) const;
    double algorithmic_bytes(int B) const;             // elem_size * sum over trajectories of (n + neF + neG)
    int dtype() const { return dtype_; }
    int device() const { return device_; }
    int windmodel() const { return windmodel_; }
    void set_windmodel(int wm) { windmodel_ = wm; }
    size_t elem_size() const { return dtype_ == TOLFG_F64 ? 8 : 4; }
    const aircraft &airframe(int i) const { return acs_.at(i); }
    const gain &gains(int mission = -1) const { return gn_.at(slot(mission)); }
    const limit &limits(int mission = -1) const { return lm_.at(slot(mission)); }
    const snopt &snopt_params(int mission = -1) const { return sn_.at(slot(mission)); }

    void set_wind_grid(const tolfg_wind_grid &g);      // uploads; switches to TOLFG_WIND_GRID
    // true (once) when an evaluation since the last call lost an objective partial (FgArgs::status); meaningful after
    // the evaluations have completed.  eval() checks it on entry and refuses to go on (hip_failure).
    bool take_lost_partial();
    void set_trajectories(int B, const tolfg_traj *trajs);
    int trajectories() const { return ntraj_; }
    const tolfg_traj &trajectory(int t) const { return host_traj_.at(t); }
    double chi_d(int t) const;
    // wind model 3 at one NED point, on the host (debug dump only): v and dv/d(east, north, up)
    bool grid_wind_host(double pn, double pe, double pd, double *v, double *dve, double *dvn, double *dvu) const;

    // asynchronous on `stream`
    // dObj (optional, needs needF): the objectives F[t][0] are also written there, contiguous
    // done (optional, host-mapped word): written with done_seq once every store of this evaluation is
    // visible to the host -- the SNOPT callback spins on it instead of synchronising the stream
    void eval(int B, const void *dX, long ldx, void *dF, long ldf, void *dG, long ldg, const void *dWind,
              int needF, int needG, hipStream_t stream, void *dObj = nullptr,
              unsigned long long *done = nullptr, unsigned long long done_seq = 0);
    void objectives(int B, const void *dF, long ldf, void *dObj, hipStream_t stream);
    void x0_device(int B, void *dX, long ldx, hipStream_t stream);
    void bounds_device(int B, void *dXlow, void *dXupp, long ldx, void *dFlow, void *dFupp, long ldf, hipStream_t stream);
    // The G buffer of B trajectories ([B][ldg] elements, ldg = the row length rounded up to 16 bytes), placed for this
    // batch's launch: up to `tries` candidates from device_alloc, each timed with the bare store loop of the launch's own
    // shape (store_shape_kernel), the fastest kept, the rest freed.  Launches whose outputs fit the Infinity Cache do not
    // depend on placement: one candidate, not timed.  probe_us (optional, [tries]): the times, 0 where not timed.  The
    // caller frees the buffer with device_free.  Blocking (set-up time: tens of ms).
    void *alloc_outputs(int B, int tries, long *ldg, double *probe_us, int *tried);
    // measurement aid: HIP events recorded on the launch stream around fg_kernel of every eval
    void set_timing(bool on);
    int kernel_time(double *avg_ms, double *min_ms);   // launches averaged since the last call
    // measurement aid: while on, eval() launches the bare store loop of its own launch shape (store_shape_kernel)
    void set_store_shape(bool on) { store_shape_ = on; }

private:
    Sizes sz_;
    Sizes szm_[2];                      // per mission id; both filled for a mixed batch
    int mission_;
    int slot(int mission) const { return mission_ == MISSION_MIXED ? (mission < 0 ? 0 : mission) : 0; }
    std::vector<aircraft> acs_;
    std::vector<gain> gn_;              // one entry, or [S10, G7] for a mixed batch
    std::vector<limit> lm_;
    std::vector<snopt> sn_;
    int windmodel_, dtype_, device_;
    void upload(hipStream_t stream);    // trajectory table + x0 table, on `stream`, drained before it returns
    int cus_ = 0;                       // compute units of device_ (asked once, at the first upload)
    double plan_out_bytes_ = 0;         // > 0: eval() plans for this many output bytes (alloc_outputs' probe launches)
    // The *_forced_ members below are measurement overrides of the launch plan: the shipped library never sets them, the
    // measurement build (libtolfg_measure.so) takes them from the environment when the object is created (knobs.h)
    bool no_single_ = false, force_single_ = false, x0_serial_ = false;
    int waves_per_cu_ = 0;              // TOLFG_WAVES_PER_CU
    bool waves_forced_ = false;
    bool timing_ = false;
    bool store_shape_ = false;
    std::vector<hipEvent_t> ev_;
    size_t ev_used_ = 0;
    void *d_grid_ = nullptr;
    double *d_tgrid_ = nullptr;         // the initial guess's per-(mission, node) table, [2][X0_FIELDS][N+1]: field 0 = node times (x0_device)
    std::vector<double> grid_host_;
    double *d_partial_ = nullptr;
    long partial_cap_ = 0;
    unsigned *d_counter_ = nullptr;
    unsigned *h_status_ = nullptr, *d_status_ = nullptr;   // pinned, device-mapped: set by a launch that lost an objective partial
    int counter_cap_ = 0;
    int tile_nodes_forced_ = 0;         // TOLFG_TILE_NODES (measurements)
    int fused_forced_ = -1;             // TOLFG_FUSED=0/1 overrides one launch vs fg_kernel + finalize_kernel (measurements)
    int nt_forced_ = -1;                // TOLFG_NT_STORES=0/1 overrides the size-based choice (measurements)
    bool partial_dirty_ = false;        // the partial slots hold a two-launch evaluation's sums (not "empty")
    int tail_forced_ = -1, tail_nt_forced_ = 0;   // TOLFG_TAIL=count:nt overrides the finer-tiled tail (measurements)
    int xcd_forced_ = -1;               // TOLFG_XCD=0/1 overrides the tile order (measurements)
    int sub_forced_ = -1;               // TOLFG_SUB_NODES=0/32 overrides the LDS passes of a tile's rows (measurements)
    int stagger_forced_ = -1;           // TOLFG_STAGGER=0/1 overrides the issue-priority stagger (measurements)
    int ntraj_ = 0, cap_ = 0;
    hipStream_t last_stream_ = nullptr;  // stream of the last evaluation: moving to another one drains it first (stream contract)
    bool have_last_stream_ = false;
    bool uploaded_ = false;
    TrajDev *d_traj_ = nullptr;
    std::vector<tolfg_traj> host_traj_;
    std::vector<TrajDev> dev_traj_;
    FgArgs args_{};
};

// ---------------------------------------------------------------------------------------------
// ref: class problem, include/problem.h:17-140.  Public surface kept: modelWind / computeF /
// computeG (what DEFINEGusrfg_ calls, src/DefineFG.cpp:24-37) and the SNOPT companion data; the
// members the reference exposes to its missions keep their names (ac, gn, lm, sn, n, neF, neG,
// iGfun, jGvar, x, xlow, xupp, Flow, Fupp, xg, yg, rg).
class problem {
    std::unique_ptr<batch> eng_;      // declared first: the parameter references below bind to it
public:
    virtual ~problem();
    problem(const problem &) = delete;
    problem &operator=(const problem &) = delete;

    // ref: problem::modelWind(x), src/problem.cpp:475.  Wind models 0/1 are fused into the kernel,
    // so this stages x on the device and starts the evaluation; computeF/computeG collect.
    void modelWind(const double x[]);
    // ref: problem::computeF(x, F), src/problem.cpp:765
    void computeF(const double x[], double F[]);
    // ref: problem::computeG(x, G), src/problem.cpp:782
    void computeG(const double x[], double G[]);
    // what DEFINEGusrfg_ uses: one launch, only the requested outputs come back
    void evaluate(const double x[], bool needF, double F[], bool needG, double G[]);

    // Arrays used in place (include/tolfg.h): pin and map the caller's x / F / G (any may be null) for the kernel to
    // address directly until forget_arrays(); forget_arrays() waits for the stream, unregisters every array and
    // returns to the staging copies.  registered_arrays(): how many arrays are pinned right now.
    void register_arrays(double *x, double *F, double *G);
    void forget_arrays();
    int registered_arrays() const;

    void set_wind_table(const double *wind_enu);   // [12][ts+1], ENU, reference member order
    void set_wind_grid(const tolfg_wind_grid &g);  // wind model 3
    // ref: problem::writeJSON(filename), src/problem.cpp:1247 -- same keys, for the same consumers
    void writeJSON(const std::string &filename, const double *xsol, double final_cost) const;

    bool debug;                       // ref: problem::debug, include/problem.h:26 (default false here)

    // SNOPT companion data (ref: include/problem.h:109-136)
    int n, neF, neG;
    std::vector<int> iGfun, jGvar;
    std::vector<double> x, xlow, xupp, Flow, Fupp;
    int ObjRow = 0;
    double ObjAdd = 0;
    int neA = 0;
    const aircraft &ac;
    const gain &gn;
    const limit &lm;
    const snopt &sn;
    double xg, yg, zg, rg;            // goal, NED (ref: src/problem.cpp:24-27)
    double east, north, up;
    std::string mission, aircraft_type;

protected:
    problem(const tolfg_config &cfg, int mission_id);

private:
    void ensure_device();
    // Fuser / Guser: the caller's arrays when known at launch time (DEFINEGusrfg_), else nullptr
    void stage_and_launch(const double xin[], bool needF, bool needG, double *Fuser = nullptr, double *Guser = nullptr,
                          bool caller_keeps_x = false);
    void collect(bool wantF, double F[], bool wantG, double G[]);
    void wait_done();
    // Device address of a caller-owned host array the kernel may read / write in place, or nullptr (then the pinned
    // staging copies are used).  In-place use is a CONTRACT, never a guess (include/tolfg.h, "Arrays used in place"):
    // arrays named by register_arrays(), or -- only with tolfg_config.persistent_arrays -- an array handed to two
    // calls in a row (SNOPT hands the same sections of its workspace to every call: src/snoptProblem.cpp:468-477).
    void *device_view(void *p, size_t bytes);
    struct HostView { void *base; size_t bytes; void *dev; int seen; };   // seen: 1 = once, 2 = registration tried
    std::vector<HostView> views_;
    bool persistent_arrays_ = false;                // tolfg_config.persistent_arrays
    bool register_user_ = true;                     // TOLFG_NO_REGISTER=1 keeps the pinned staging copies (measurement build)
    bool use_flag_ = true;                          // TOLFG_NO_FLAG=1 synchronises the stream instead
    bool copy_x_ = false;                           // TOLFG_CALLBACK_COPY_X=1 always stages x (measurement build)
    unsigned long long *done_ = nullptr;            // pinned, device-mapped completion word
    unsigned long long seq_ = 0;
    bool flagged_ = false;                          // the evaluation in flight reports through done_
    double *landF_ = nullptr, *landG_ = nullptr;    // where its F and G arrive on the host
    void dump(const char *name, const double *v, int len);
    void dump_wind(const double *x);
    std::vector<double> wind_host_;                 // host copy of the table wind, for the dump

    hipStream_t stream_ = nullptr;
    static constexpr int kChunks = 6;               // pieces of G's device-to-host copy (staged path)
    hipEvent_t chunk_ev_[kChunks] = {};
    bool chunked_ = false;
    int nchunks_ = 2;               // measured at ts=2000: 1 / 2 / 4 / 6 pieces -> 213 / 169 / 177 / 237 us per call
    double *hx_ = nullptr, *hF_ = nullptr, *hG_ = nullptr;      // pinned
    double *dX_ = nullptr, *dF_ = nullptr, *dG_ = nullptr, *dW_ = nullptr;
    long ldx_, ldf_, ldg_;
    bool device_ready_ = false, staged_ = false, haveF_ = false, haveG_ = false;
    bool x_copied_ = true;              // hx_ holds the x of the staged evaluation (false: the kernel read the caller's own array)
    size_t zero_copy_limit_ = 64u << 20;    // bytes of x+F+G up to which the kernels address host memory directly
    bool zero_copy_ = true;           // TOLFG_CALLBACK_STAGING=1 selects explicit H2D/D2H copies instead
};

// ref: class problemS10 / problemG7 -- the two missions on the path
class problemS10 : public problem {
public:
    explicit problemS10(const tolfg_config &cfg);
};
class problemG7 : public problem {
public:
    explicit problemG7(const tolfg_config &cfg);
    double chi_d;                     // ref: include/problemG7.h:23
};

// ref: `problem *prob`, src/tol.cpp:3 -- the context the SNOPT callback evaluates
extern problem *prob;

}  // namespace tolfg
#endif
