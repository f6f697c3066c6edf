/*
 * tolfg.h -- C ABI of libtolfg.so, the MI355X-native implementation of tol's SNOPT user function.
 *
 * Everything here is plain C: pointers, sizes, opaque handles.  Citations "ref:" point into the
 * reference tree (lingaqing/tol) and name the interface each entry point replaces.
 *
 * Three groups:
 *   1. the SNOPT callback itself, DEFINEGusrfg_, byte-for-byte the reference's snFunA symbol;
 *   2. problem set-up: what the reference's `new problemS10(args)` / `new problemG7(args)` and
 *      problem::runSNOPT hand to SNOPT (sizes, sparsity pattern, initial guess, bounds);
 *   3. a batched, device-resident evaluation (no reference counterpart) for many independent
 *      trajectories per GPU.
 *
 * There is no CPU fallback: every evaluation entry point needs a gfx950 device and reports failure
 * (return code / *Status) when the HIP runtime or the device is unavailable.
 */
#ifndef TOLFG_H_
#define TOLFG_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ return codes */
enum {
    TOLFG_OK            = 0,
    TOLFG_ERR_ARG       = -1,   /* bad argument (null pointer, unknown mission, size mismatch)   */
    TOLFG_ERR_PARAM     = -2,   /* .param file missing or wrong element count
                                   (ref: std::length_error, src/parameters.cpp:45-67)            */
    TOLFG_ERR_HIP       = -3,   /* HIP runtime / device failure; tolfg_last_error() has the text */
    TOLFG_ERR_NOCURRENT = -4    /* DEFINEGusrfg_ called with no current problem                  */
};

/* thread-local text of the last failure in this library ("" when none) */
const char *tolfg_last_error(void);

/* wind models (ref: problem::modelWind, src/problem.cpp:475-531,732-735) */
enum {
    TOLFG_WIND_NONE  = 0,    /* case 0 */
    TOLFG_WIND_SHEAR = 1,    /* case 1, linear boundary layer: the offline fallback (src/problem.cpp:73-78) */
    TOLFG_WIND_GRID  = 3,    /* case 3, gridded storm field, trilinear interpolation of the v component
                                and its gradient (src/problem.cpp:544-695); the grid comes from
                                tolfg_set_wind_grid instead of the reference's MongoDB cache          */
    TOLFG_WIND_TABLE = 99    /* caller-supplied per-node wind; the reference's `default:` arm       */
};
/* Codes 2, 4 and 5 (thermal, two thermals, cyclic wind) are accepted and mean TOLFG_WIND_NONE: the bodies of those arms are
 * commented out in the reference (src/problem.cpp:534-542,698-730), so its modelWind leaves the wind vectors as the
 * constructor zeroed them.  Any other code is refused (TOLFG_ERR_ARG). */

/* A regular ENU grid of the north wind component v, the only one the reference interpolates
 * (src/problem.cpp:628-635,682-692).  v[(i*ny + j)*nz + k] is the value at east x0+i*dx, north
 * y0+j*dy, up z0+k*dz.  The aircraft's grid position is its NED position mapped to ENU plus
 * (east/north/up)_from_datum (ref: EastFromDatum..., src/problem.cpp:411-413).  Points outside
 * the grid are evaluated in the edge cell (the reference indexes out of bounds there).
 * nx != ny: the first index is the EAST index with its own extent nx, as the reference's cacheWind builds the cache
 * (src/problem.cpp:437-441).  The reference's search loops bound the east index by the NORTH count and the north index
 * by the east count (src/problem.cpp:556-566); inside the grid that changes nothing -- a loop that runs out of its
 * bound ends on the index it would have stopped at -- and the reference-evaluated fixture holds a 5 x 3 x 4 case that
 * this library reproduces; outside the grid the reference's reading is undefined and this library keeps the edge cell. */
typedef struct tolfg_wind_grid {
    int    nx, ny, nz;        /* >= 2 each */
    double x0, y0, z0;
    double dx, dy, dz;        /* 150 m in the reference (include/problem.h:90-92) */
    double east_from_datum, north_from_datum, up_from_datum;
    const double *v;          /* host array, nx*ny*nz values */
} tolfg_wind_grid;

enum { TOLFG_F64 = 0, TOLFG_F32 = 1 };

/* Jacobian sparsity pattern handed to SNOPT.
 * REFERENCE: exactly what problem::countG builds (src/problem.cpp:813-919), 104 entries per node of
 *            which 58 are structural zeros -- the drop-in choice and the default.
 * COMPACT:   the same rows without entries that are zero for every x (46 per node, and no dt entries
 *            in the boundary rows): 1.85x fewer bytes per evaluation.  Any driver that takes the
 *            pattern from tolfg_pattern() can use it; values of the kept entries are identical. */
enum { TOLFG_PATTERN_REFERENCE = 0, TOLFG_PATTERN_COMPACT = 1 };

/* ------------------------------------------------------------------ 2. problem set-up */

/* What the reference takes from argv (ref: arguments::arguments(char**), src/arguments.cpp:32-46)
 * plus the constants it hard-codes, exposed with the reference's values as defaults when the
 * struct is filled by tolfg_config_default(). */
typedef struct tolfg_config {
    const char *mission;      /* "S10" | "G7"                      ref: src/tol.cpp:9-23            */
    const char *aircraft;     /* basename of aircraft/<name>.param ref: src/parameters.cpp:42-43    */
    const char *root_path;    /* directory holding aircraft/ and problems/ (ref root_path,
                                 src/arguments.cpp:45); NULL = the data shipped with this library  */
    double east, north, up;                        /* argv[1..3], stored, unused on the path        */
    double east_goal, north_goal, up_goal;         /* argv[4..6]  (NED goal = north, east, -up)     */
    double radius_goal;                            /* argv[7]                                       */
    int    ts;                /* time segments; 0 = take snopt.param's (100 as shipped)            */
    int    windmodel;         /* TOLFG_WIND_*; reference offline behaviour = TOLFG_WIND_SHEAR      */
    double Vref, href;        /* shear wind constants, 2.4 and 10 (ref: src/problem.cpp:504-505)   */
    double xi, yi, zi;        /* start position, 0,0,0 (ref: src/problem.cpp:83-85,111-113)        */
    int    device;            /* HIP device ordinal                                                */
    int    debug_dumps;       /* 1 = also rewrite Xoutput/Foutput/Goutput.txt and Woutput.txt each call like
                                 the reference does (src/DefineFG.cpp:16-46, src/problem.cpp:740-756);
                                 default 0                                                          */
    int    pattern;           /* TOLFG_PATTERN_*; default REFERENCE                                */
    int    persistent_arrays; /* 1 = the caller promises that the x, F and G arrays it hands to DEFINEGusrfg_ stay
                                 allocated, at their addresses, until tolfg_forget_arrays() / tolfg_destroy() --
                                 true for snOptA, whose x, F, G are sections of its workspace for the whole solve
                                 (ref: src/snoptProblem.cpp:468-477).  See "Arrays used in place".  Default 0.     */
} tolfg_config;

void tolfg_config_default(tolfg_config *cfg);

typedef struct tolfg_problem tolfg_problem;

/* ref: `prob = new problemS10(args)` / `new problemG7(args)`, src/tol.cpp:9-17 -- reads the four
 * .param files, builds pattern, initial guess and bounds, allocates device and pinned buffers. */
int  tolfg_create(const tolfg_config *cfg, tolfg_problem **out);
/* ref: `delete prob`, src/tol.cpp:32 */
void tolfg_destroy(tolfg_problem *p);

/* Companion data the SNOPT driver needs (ref: problem::runSNOPT, src/problem.cpp:1223-1233).
 * n, neF: src/problem.cpp:151-152; neG: what countG counts (src/problem.cpp:813-919). */
int tolfg_sizes(const tolfg_problem *p, int *n, int *neF, int *neG);
/* iGfun/jGvar[neG], 0-based like the reference's arrays before snoptProblemA::solve shifts them
 * (ref: src/snoptProblem.cpp:460-465). */
int tolfg_pattern(const tolfg_problem *p, int *iGfun, int *jGvar);
/* ref: problemS10::InitialCond / problemG7::InitialCond */
int tolfg_x0(const tolfg_problem *p, double *x);
/* ref: problem::setLimits, src/problem.cpp:198-365 */
int tolfg_bounds(const tolfg_problem *p, double *xlow, double *xupp, double *Flow, double *Fupp);
/* ref: sn.opt_tol / sn.feas_tol handed to SNOPT, src/problem.cpp:1235-1236 */
int tolfg_tolerances(const tolfg_problem *p, double *opt_tol, double *feas_tol);
/* Table wind for TOLFG_WIND_TABLE: wind[f*(ts+1)+k], f = 0..11 in the order of the reference's
 * member vectors u v w du_dx du_dy du_dz dv_dx dv_dy dv_dz dw_dx dw_dy dw_dz, ENU convention
 * (ref: include/problem.h:103).  Copied to the device; switches the problem to table wind. */
int tolfg_set_wind_table(tolfg_problem *p, const double *wind_enu);
/* Gridded wind (TOLFG_WIND_GRID): copied to the device; switches the problem to wind model 3. */
int tolfg_set_wind_grid(tolfg_problem *p, const tolfg_wind_grid *grid);

/* Result file of one solved leg, with the keys the reference's writer emits and its consumers read
 * (ref: problem::writeJSON, src/problem.cpp:1247-1365; msl/mission.py:208-226;
 * matlab/@plotSNOPT/plotSNOPT.m:48-58): args, problem, FinalCost, dt, trajectory{time,x,y,z,Va,gam,
 * chi,phi,CL,dphi,dCL,T}, aircraft{...}, gains{...}, limits{...}, snopt{...}.  x is the solved
 * decision vector (n values), final_cost the objective F[0] SNOPT ended with. */
int tolfg_write_json(const tolfg_problem *p, const double *x, double final_cost, const char *filename);

/* ------------------------------------------------------------------ 1. the SNOPT callback */

/* ref: `problem *prob` global, src/tol.cpp:3 / include/global_objects.h:5.  The callback evaluates
 * the current problem.  (If iu != NULL and *leniu >= 2 and iu[0] == TOLFG_IU_MAGIC the callback
 * instead uses the handle index in iu[1] -- see tolfg_handle_index -- so several problems can be
 * solved from one process; tol itself never sets user workspace.) */
void           tolfg_set_current(tolfg_problem *p);
tolfg_problem *tolfg_get_current(void);
#define TOLFG_IU_MAGIC 0x70F6
int            tolfg_handle_index(const tolfg_problem *p);

/* ref: include/DefineFG.h:9-14, assignable to snFunA (include/snopt/snopt.h:60-66).
 * Reads x[0..*n), writes F[0..*neF) iff *needF > 0 and G[0..*neG) iff *needG > 0, complete on
 * return.  The reference never touches *Status; this build sets *Status = -2 (snOptA: terminate)
 * on a HIP failure or size mismatch and leaves it untouched otherwise. */
void DEFINEGusrfg_(int *Status, int *n, double x[],
                   int *needF, int *neF, double F[],
                   int *needG, int *neG, double G[],
                   char *cu, int *lencu,
                   int iu[], int *leniu,
                   double ru[], int *lenru);

/* Arrays used in place.
 * By default every DEFINEGusrfg_ call copies x into, and F and G out of, pinned staging buffers of the library
 * (17 + 190 KB at ts = 200): safe for any caller, whatever it does with its arrays between calls.
 * The kernel can instead read x and write F and G where they lie in the caller's memory: no copies, 16.8 us per call at
 * ts = 200 against 23.8 us staged (BENCH_r04, driver's box) -- i.e. 1.19e7 against 8.4e6 collocation-node evaluations
 * per second: the >= 1e7 callback rate is met ONLY under the in-place contract.  That needs the arrays pinned and mapped into the GPU's address space
 * (hipHostRegister), and a pinning is tied to the ADDRESS, not to the allocation: an array that is freed and
 * re-allocated -- even at the same address -- is no longer the memory the GPU writes.  So in-place use is a contract
 * the caller enters explicitly, one of two ways:
 *   * tolfg_register_arrays(p, x, F, G): these arrays (16-byte aligned; any may be NULL) are used in place from
 *     now on, whenever a call passes exactly these pointers; or
 *   * tolfg_config.persistent_arrays = 1: an array passed to two DEFINEGusrfg_ calls in a row is taken to be one
 *     the caller keeps and is registered on that second call (the drop-in choice for snOptA, which hands sections
 *     of its own workspace that the driver never sees).
 * Either way the caller must call tolfg_forget_arrays(p) BEFORE freeing or re-allocating any such array (it waits
 * for the evaluation in flight, unpins everything and returns to the staging copies); tolfg_destroy does the same.
 * tolfg_registered_arrays(p) tells how many arrays are pinned at the moment. */
int tolfg_register_arrays(tolfg_problem *p, double *x, double *F, double *G);
int tolfg_forget_arrays(tolfg_problem *p);
int tolfg_registered_arrays(const tolfg_problem *p);

/* Measurement aid: enter DEFINEGusrfg_ `calls` times from native code through an snFunA function pointer,
 * the way snOptA does (x, F, G: caller arrays of the problem's sizes, the same ones every call, used in place for
 * the duration of this function where they are 16-byte aligned, and forgotten again before it returns), after
 * `warm` untimed calls; *us_per_call receives the mean wall time of one call.  needF / needG are passed through
 * (snOptA asks for F alone during its line searches).  Returns the last *Status the callback left (1 = untouched)
 * or a negative TOLFG_ERR_*. */
int tolfg_time_callback(tolfg_problem *p, const double *x, double *F, double *G, int needF, int needG, int warm, int calls,
                        double *us_per_call);
/* The same with the array contract named: in_place = 1 is tolfg_time_callback (arrays registered for the duration);
 * in_place = 0 times the DEFAULT contract -- nothing registered, every call copies x into and F, G out of the library's
 * pinned staging buffers. */
int tolfg_time_callback_as(tolfg_problem *p, const double *x, double *F, double *G, int needF, int needG, int in_place, int warm,
                           int calls, double *us_per_call);

/* The three public methods DEFINEGusrfg_ dispatches to in the reference
 * (ref: problem::modelWind / computeF / computeG, src/problem.cpp:475,765,782), for callers that
 * drive them separately.  modelWind stages x on the device; computeF / computeG evaluate on demand
 * (one fused launch serves both when they follow the same modelWind). */
int tolfg_modelWind(tolfg_problem *p, const double *x);
int tolfg_computeF(tolfg_problem *p, const double *x, double *F);
int tolfg_computeG(tolfg_problem *p, const double *x, double *G);

/* ------------------------------------------------------------------ 3. batched evaluation */

/* One trajectory of a batch.  All trajectories of a batch share ts; they share the mission too unless
 * the batch was created with mission "mixed", in which case `mission` selects it per trajectory. */
enum { TOLFG_MISSION_S10 = 0, TOLFG_MISSION_G7 = 1 };
typedef struct tolfg_traj {
    int    aircraft;          /* index into tolfg_batch_config.aircraft[]                          */
    int    mission;           /* TOLFG_MISSION_*; read only by "mixed" batches                     */
    double Vref, href;        /* shear wind of this trajectory (TOLFG_WIND_SHEAR)                  */
    double north_goal, east_goal, radius_goal;
    double xi, yi;            /* start position: G7's course chi_d = atan2(yg-yi, xg-xi)           */
    double zi;                /* start height (NED, so negative up), used by the device-side set-up */
} tolfg_traj;

typedef struct tolfg_batch_config {
    const char        *mission;       /* "S10" | "G7" | "mixed" (BASELINE configs[4]: G7 and S10 trajectories
                                         evaluated by ONE launch; every row of X/F/G keeps its own mission's
                                         SNOPT layout, rows are strided for the larger mission)          */
    const char        *root_path;     /* NULL = shipped data                                      */
    const char *const *aircraft;      /* aircraft table: names of .param files                     */
    int                n_aircraft;    /* 1..8                                                      */
    int                ts;            /* 0 = snopt.param's                                         */
    int                windmodel;
    int                dtype;         /* TOLFG_F64 | TOLFG_F32: element type of X, F, G, wind      */
    int                device;
    int                pattern;       /* TOLFG_PATTERN_*; default REFERENCE                        */
} tolfg_batch_config;

typedef struct tolfg_batch tolfg_batch;

int  tolfg_batch_create(const tolfg_batch_config *cfg, tolfg_batch **out);
void tolfg_batch_destroy(tolfg_batch *b);
/* Row sizes to allocate for (a mixed batch: the larger of the two missions per quantity) and the
 * pattern of a single-mission batch. */
int  tolfg_batch_sizes(const tolfg_batch *b, int *n, int *neF, int *neG);
int  tolfg_batch_pattern(const tolfg_batch *b, int *iGfun, int *jGvar);
/* The same per mission (TOLFG_MISSION_*), for the rows of a mixed batch. */
int  tolfg_batch_mission_sizes(const tolfg_batch *b, int mission, int *n, int *neF, int *neG);
int  tolfg_batch_mission_pattern(const tolfg_batch *b, int mission, int *iGfun, int *jGvar);
/* Describe (or re-describe) the B trajectories; uploads a small per-trajectory table. */
int  tolfg_batch_set_trajectories(tolfg_batch *b, int B, const tolfg_traj *trajs);
/* one gridded wind field for the whole batch (TOLFG_WIND_GRID); copied to the device */
int  tolfg_batch_set_wind_grid(tolfg_batch *b, const tolfg_wind_grid *grid);
/* initial guess of trajectory t (host, double) -- the reference's InitialCond with (xi,yi,zi) */
int  tolfg_batch_x0(const tolfg_batch *b, int t, double zi, double *x);
/* Set-up on the device for batched warm starts: the initial guess of
 * trajectories [0,B) written straight into the rows of dX (ref: InitialCond with the trajectory's
 * (xi,yi,zi)), and the bounds into dXlow/dXupp [B][ldx] and dFlow/dFupp [B][ldf] (ref: setLimits).
 * Element type = the batch dtype.  Asynchronous on stream. */
int  tolfg_batch_x0_device(tolfg_batch *b, int B, void *dX, long ldx, void *stream);
int  tolfg_batch_bounds_device(tolfg_batch *b, int B, void *dXlow, void *dXupp, long ldx,
                               void *dFlow, void *dFupp, long ldf, void *stream);
/* bounds of trajectory t (ref: problem::setLimits), any pointer may be NULL */
int  tolfg_batch_bounds(const tolfg_batch *b, int t, double zi,
                        double *xlow, double *xupp, double *Flow, double *Fupp);

/* Stream contract: a batch owns per-launch workspace (objective partials, arrival counters), so its evaluations run one
 * at a time.  Evaluations issued on ONE stream are ordered by the stream; when an evaluation names a different stream than
 * the one before it, the library drains the previous stream first (blocking: use one stream per batch, several batches for
 * several streams).  A stream synchronisation must separate the last evaluation from tolfg_batch_set_trajectories /
 * tolfg_batch_set_wind_grid (they copy on the null stream).
 * The first evaluation of a batch (and one with a larger B than before) allocates and uploads; do not
 * issue it inside a hipGraph capture -- warm up once, then capture (tests/test_gpu_parity.py does). */
/* Evaluate F and G of trajectories [0,B) in one launch.  dX, dF, dG, dWind are DEVICE pointers to
 * elements of the batch dtype; row t of X/F/G starts ldx/ldf/ldg elements after row t-1
 * (ld >= n / neF / neG; dX and dF on 16-byte boundaries with ldx, ldf even -- multiples of 4 for fp32 -- keep the
 * 16-byte window loads and defect stores; G may sit anywhere, its slabs are always streamed with 16-byte stores).  dWind is NULL unless windmodel is
 * TOLFG_WIND_TABLE, then [B][12][ts+1].  dObj is NULL or a device array of B elements that also
 * receives the objectives F[t][0], contiguous (needs needF).  stream is a hipStream_t (NULL =
 * default stream).  Asynchronous: returns after enqueueing. */
int  tolfg_batch_eval(tolfg_batch *b, int B,
                      const void *dX, long ldx, void *dF, long ldf, void *dG, long ldg,
                      const void *dWind, int needF, int needG, void *dObj, void *stream);
/* Health of the evaluations issued so far, to be asked after they have completed (stream synchronisation): TOLFG_OK,
 * or TOLFG_ERR_HIP when a launch lost an objective partial -- the one-launch evaluation hands the tiles' objective
 * terms to the finalizing wave through polled slots whose "empty" marker is a NaN bit pattern (0xFFFBADADFFFBADAD); an x
 * that carries exactly that NaN makes a slot look empty for good, the wave gives up after a bounded wait, F[0] of that
 * trajectory is a NaN and this call says so (and clears the condition).  tolfg_batch_eval itself refuses to start
 * (TOLFG_ERR_HIP) while the condition is pending; DEFINEGusrfg_ reports it as *Status = -2. */
int  tolfg_batch_status(tolfg_batch *b);
/* dObj[t] = F[t][0] for t in [0,B): the per-trajectory objectives, contiguous, ready for the
 * RCCL all-gather across GPUs (a separate small kernel; tolfg_batch_eval's dObj does it for free). */
int  tolfg_batch_objectives(tolfg_batch *b, int B, const void *dF, long ldf, void *dObj, void *stream);

/* Where the outputs live.  A launch beyond the Infinity Cache streams F and G from ~2000 concurrent store fronts, and WHERE
 * the G buffer landed in HBM decides whether they run at 4.7 or at 6 TB/s: one and the same launch takes 308 us on one 1.4 GB
 * allocation and 281 us on the next (profiles/r04_allocation_classes.md; the vendor's fill runs at the same speed on both).  A
 * plain hipMalloc of that size lands in the slow class most of the time; one address range backed by 2 MiB physical chunks
 * (HIP virtual-memory management) lands in the fast one most of the time.  So the library offers:
 *   tolfg_device_alloc / tolfg_device_free: device memory in that form (falls back to hipMalloc where the runtime has no
 *     virtual-memory support) -- for X, F, G or anything else; a pointer from it is an ordinary device pointer;
 *   tolfg_batch_alloc_outputs: the G buffer of B trajectories ([B][*ldg] elements of the batch dtype, *ldg = the row length
 *     rounded up to 16 bytes), PLACED for this batch's launch: up to `tries` candidates (1..16, and never more than fit side by side into half of the free device memory; the Python layer and tolfg_multi take 12: ~0.2 s each at 1.4 GB, most of it the settling of the fresh block) from
 *     tolfg_device_alloc, held side by side, each timed with the bare store loop of the launch's own shape
 *     (tolfg_batch_set_store_shape); the search ends early once a candidate is 18 % faster than the slowest seen (fast and slow
 *     class are ~20 % apart); the fastest is kept, the rest freed.  probe_us (NULL or [tries]) receives the candidates' times in us, *tried (NULL or int)
 *     how many were timed -- 1 and no timing when the launch's outputs fit the cache, where placement does not matter.
 *     Blocking, up to a few seconds for a dozen gigabyte-sized candidates: set-up time.  Free the buffer with tolfg_device_free.
 * Buffers from anywhere else keep working; they just take the class they land in. */
/* A block comes SETTLED and zeroed: the driver wipes video memory it gets back with a copy job of its own and may hand the chunks out
 * again before that job has run, so that a fresh block was seen to go back to 0.0, chunk by chunk, milliseconds after a kernel had
 * written it (profiles/r05_fresh_vmm_blocks.md).  tolfg_device_alloc therefore fills the block with a pattern and returns only once the
 * pattern has held through a quiet period (>= 1 ms; ~0.1 s per GB), synchronising the device while it does: call it at set-up time. */
int  tolfg_device_alloc(int device, size_t bytes, void **ptr);
int  tolfg_device_free(void *ptr);      /* waits for the device first, like hipFree does */
int  tolfg_batch_alloc_outputs(tolfg_batch *b, int B, int tries, void **dG, long *ldg, double *probe_us, int *tried);

/* Measurement aid.  While enabled, every tolfg_batch_eval attaches a start and a stop HIP event to its dispatches
 * (hipExtLaunchKernelGGL): around the whole evaluation -- fg_kernel alone in the single-launch form, fg_kernel through
 * finalize_kernel in the two-launch form.  tolfg_batch_kernel_time waits for the recorded launches, returns how many
 * there were and their average / minimum duration in ms, and resets the record.  Dispatch profiling is not free: every
 * launch takes 4-16 us longer while it is on (profiles/r02_event_cost.md), so bench.py keeps it out of its timed region. */
int  tolfg_batch_set_timing(tolfg_batch *b, int enable);
int  tolfg_batch_kernel_time(tolfg_batch *b, double *avg_ms, double *min_ms);
/* Measurement aid (calibration of a box).  While enabled, tolfg_batch_eval launches -- instead of the evaluation -- a
 * bare store loop in the evaluation's own launch shape: the same grid (one wave per tile), tile order over the XCDs,
 * resident-wave cap and store flavour (non-temporal or plain), every wave writing the 16-byte vectors of its tile's
 * Jacobian slab region with a constant; no loads, no arithmetic.  Its rate is a reference point for THIS stream shape on THIS
 * box, measured by bench.py in the same process as the evaluation (roofline.box_stream_shape_GBs, vs_bare_store_loop) -- not a
 * ceiling: the evaluation has read 1.00-1.03 x of it.  F and G hold garbage while it is on. */
int  tolfg_batch_set_store_shape(tolfg_batch *b, int enable);

/* algorithmic bytes one evaluation of trajectories [0,B) moves: elemsize * sum of (n + neF + neG)
 * (SURVEY.md section 8d), with each trajectory's own mission sizes and the batch's pattern */
double tolfg_batch_algorithmic_bytes(const tolfg_batch *b, int B);

/* ------------------------------------------------------------------ 4. several GPUs of one node, one process */

/* A batch sharded over the devices of one node (no reference counterpart; BASELINE north star: "a batch of independent
 * trajectories shards embarrassingly across the 8 GPUs of one node, RCCL over xGMI only for the final objective
 * gather").  One process: per device one tolfg_batch, one HIP stream and one issuing host thread, so that the launches
 * reach the devices side by side.  Trajectory t of `total` lives on device i with lo_i <= t < hi_i, where
 * [lo_i, hi_i) = tolfg_shard_bounds(total, i, n_devices): contiguous shards, the first total % n_devices one longer.
 * The only collective is ncclAllGather of the objectives F[t][0] (plus ncclAllReduce for their mean); the RCCL library is
 * resolved at run time (the copy the process already holds, else the one beside the HIP runtime in use, else
 * TOLFG_RCCL_LIBRARY), so libtolfg.so has no link-time dependency on it.  cfg->device is ignored (devices[] rules). */
typedef struct tolfg_multi tolfg_multi;
int  tolfg_multi_create(const tolfg_batch_config *cfg, const int *devices, int n_devices, tolfg_multi **out);
void tolfg_multi_destroy(tolfg_multi *m);
int  tolfg_multi_sizes(const tolfg_multi *m, int *n, int *neF, int *neG);
/* describe all `total` trajectories in global order; device i keeps its shard and (re)allocates X, F, G for it */
int  tolfg_multi_set_trajectories(tolfg_multi *m, long total, const tolfg_traj *trajs);
int  tolfg_multi_shard(const tolfg_multi *m, int device_index, long *lo, long *hi);
/* device pointers (on devices[device_index]) of the shard's rows, SNOPT layout, strides in elements of the batch dtype */
int  tolfg_multi_buffers(const tolfg_multi *m, int device_index, void **dX, long *ldx, void **dF, long *ldf, void **dG, long *ldg);
/* wind: one gridded field for every device (wind model 3), or -- batches created with TOLFG_WIND_TABLE -- the per-trajectory
 * tables of ALL trajectories, [total][12][ts+1] doubles in global order (each device receives its shard's rows) */
int  tolfg_multi_set_wind_grid(tolfg_multi *m, const tolfg_wind_grid *grid);
int  tolfg_multi_set_wind_tables(tolfg_multi *m, const double *wind_enu);
/* initial guesses of every trajectory, generated on the devices (ref: InitialCond with each trajectory's start) */
int  tolfg_multi_x0(tolfg_multi *m);
/* one evaluation of every shard: one launch per device, asynchronous */
int  tolfg_multi_eval(tolfg_multi *m, int needF, int needG);
/* the same, every device reading its shard's x rows from dX[device_index] (a device pointer ON that device, row stride as
 * tolfg_multi_buffers reports) instead of from the library's own X: warm starts, inputs in rotation */
int  tolfg_multi_eval_from(tolfg_multi *m, const void *const *dX, int n_devices, int needF, int needG);
/* ncclAllGather of the objectives over the devices, then waits for all of them; host_obj (optional): `total` values of
 * the batch dtype in global trajectory order.  Reports a lost objective partial of any device (tolfg_batch_status). */
int  tolfg_multi_gather_objectives(tolfg_multi *m, void *host_obj);
/* The gather without the wait.  Every device has a launch stream and a (high-priority) gather stream, and
 * TOLFG_MULTI_SLOTS = 4 objective buffers in rotation.  tolfg_multi_gather_begin enqueues the all-gather of the objectives of
 * the evaluation issued last on the gather streams, behind an event on the launch streams, and returns a ticket at once;
 * the next tolfg_multi_eval writes the next buffer and runs BESIDE the gather (a launch only ever waits for the gather four
 * back, and puts nothing into its stream when that one is done, as it normally is).  tolfg_multi_gather_wait(ticket)
 * waits for that gather on every device, reports a lost objective partial, and (host_obj != NULL) delivers the `total`
 * objectives in global trajectory order.  A ticket is valid until four further gathers have begun (TOLFG_ERR_ARG after).
 * tolfg_multi_step = eval (dX as in tolfg_multi_eval_from, or NULL for the library's X; F always) + gather_begin. */
enum { TOLFG_MULTI_SLOTS = 4 };
int  tolfg_multi_gather_begin(tolfg_multi *m, unsigned long *ticket);
int  tolfg_multi_gather_wait(tolfg_multi *m, unsigned long ticket, void *host_obj);
int  tolfg_multi_step(tolfg_multi *m, const void *const *dX, int n_devices, int needG, unsigned long *ticket);
/* How the collective is issued.  GROUPED (default): one ncclGroupStart / ncclGroupEnd bracket around the devices' calls,
 * from the caller's thread.  THREADS: every device's issuing thread calls for its own communicator, no group (NCCL's
 * one-thread-per-device form); a tolfg_multi_step then needs no rendezvous of the host threads.  Same results.  The bracket is
 * serial work of ~18 us per device for the calling thread (150 us per step at 8 devices, against <= 45 us per thread):
 * choose THREADS -- or TOLFG_MULTI_GATHER_HOST -- when a device's launch is shorter than that (profiles/r05_native_multi.md). */
enum { TOLFG_MULTI_ISSUE_GROUPED = 0, TOLFG_MULTI_ISSUE_THREADS = 1 };
int  tolfg_multi_set_issue(tolfg_multi *m, int mode);
/* Where the objectives are gathered.  RCCL (default): ncclAllGather into a device vector on every device.  HOST: no collective
 * at all -- every device's finalizing waves store their trajectories' objectives straight into ONE pinned, device-mapped host
 * vector, each shard at its global offset, so that a gather is an event behind the launch (gather_begin) and a wait for it
 * (gather_wait): for consumers on the host (Monte-Carlo statistics, one SQP driver per trajectory) that takes the collective's
 * launch and its stream hand-over out of every step (one device: 20.4 -> 14 us per step at 128 trajectories,
 * profiles/r05_native_multi.md); the devices do not receive each other's objectives.  Results are the same numbers.
 * Call while nothing is in flight (it waits). */
enum { TOLFG_MULTI_GATHER_RCCL = 0, TOLFG_MULTI_GATHER_HOST = 1 };
int  tolfg_multi_set_gather(tolfg_multi *m, int mode);
/* Candidates of the per-device placement search for the G buffers (tolfg_batch_alloc_outputs) that the NEXT
 * tolfg_multi_set_trajectories runs, the devices searching side by side on their own threads: default 12 (each device holds
 * its candidates within half of its free memory, ~0.2 s per 1.4 GB candidate); 0 or 1 = one allocation, no search. */
int  tolfg_multi_set_placement(tolfg_multi *m, int tries);
/* Measurement aid (bench.py --native-multi): `warm` untimed steps, then `steps` steps issued from native code between two
 * full synchronisations.  A step = one launch per device (+ the asynchronous gather when gather != 0).  dX: n_x sets of
 * per-device X pointers, dX[j * n_devices + i], used in rotation (n_x = 0: the library's own X).  launch_us_per_device
 * (NULL or [n_devices]): (HIP event after the device's last launch - before its first) / steps. */
typedef struct tolfg_multi_timing {
    double wall_us_per_step;     /* host clock, first issue to everything complete, / steps                           */
    double launch_us_per_step;   /* the slowest device's launch_us_per_device                                         */
    double issue_us_per_step;    /* host time spent issuing, / steps                                                  */
    double gather_us;            /* one synchronous gather (begin + wait), nothing else in flight; 0 without gather   */
    int    devices, steps, issue, gather;
} tolfg_multi_timing;
int  tolfg_multi_time_steps(tolfg_multi *m, int n_x, const void *const *dX, int needF, int needG, int gather, int warm, int steps,
                            tolfg_multi_timing *out, double *launch_us_per_device);
/* Monte-Carlo mean of the objectives: one ncclAllReduce(sum) of the per-device partial sums */
int  tolfg_multi_mean_objective(tolfg_multi *m, double *mean);
int  tolfg_multi_sync(tolfg_multi *m);
/* which librccl was loaded ("" before the first tolfg_multi_create) and its ncclGetVersion code (0 = unknown) */
const char *tolfg_multi_rccl_library(void);
int  tolfg_multi_rccl_version(void);

/* The sharding rule and the re-ordering of an all-gather's equally sized blocks into global trajectory order, for
 * callers that run their own collectives (one process per GPU: tol_amd/distributed.py uses the same rule). */
int  tolfg_shard_bounds(long total, int rank, int world, long *lo, long *hi);
int  tolfg_compact_gathered(const void *padded, size_t elem_size, long total, int world, void *out);

/* ------------------------------------------------------------------ misc */
/* directory of the .param data shipped with the library (tol_amd/data) */
const char *tolfg_default_root(void);
/* .param reader (ref: parameters::readparams, src/parameters.cpp:14-34).  Returns the number of
 * values found (may exceed maxn; only maxn are stored) or TOLFG_ERR_PARAM if unreadable. */
int tolfg_read_params(const char *path, double *out, int maxn);
const char *tolfg_version(void);
/* 1 in tol_amd/lib/libtolfg_measure.so (the same sources compiled with -DTOLFG_MEASURE: it also reads the measurement
 * variables tabled in tol_amd/csrc/knobs.h), 0 in the shipped library */
int tolfg_measurement_build(void);

/* ------------------------------------------------------------------ Environment
 * The shipped library reads exactly three environment variables, once per process, and none of them changes which kernel
 * an evaluation runs (tol_amd/csrc/knobs.h holds the table, measurement build included):
 *   TOLFG_RCCL_LIBRARY=path       the collective library tolfg_multi_create loads, instead of its own search (the copy
 *                                 the process already holds, else the one beside the HIP runtime in use); final: that
 *                                 library or an error
 *   TOLFG_TRACE=1                 DEFINEGusrfg_ prints one line per call on stderr: stage+launch / wait / copy-out times
 *   TOLFG_MULTI_SHARED_DEVICES=1  test seam: tolfg_multi_create accepts a device ordinal more than once.  Honoured only
 *                                 together with TOLFG_RCCL_LIBRARY (RCCL itself refuses such a list), and announced on
 *                                 stderr.  Never set in production.
 */

#ifdef __cplusplus
}
#endif
#endif /* TOLFG_H_ */
