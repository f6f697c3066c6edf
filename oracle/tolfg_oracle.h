/*
 * tolfg_oracle.h -- CPU restatement of tol's SNOPT user-function path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle for the HIP path in tol_amd/.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link, load or call anything in this directory; the product
 * library (tol_amd/lib/libtolfg.so) never does and has no CPU fallback.
 *
 * PARITY STATUS: "partially pinned".  The reference (/root/reference, lingaqing/tol) ships no tests,
 * golden vectors or result files for this path, and its hot-path translation
 * units cannot be compiled in this image without writing stand-ins for the absent MongoDB client
 * headers and the commercial SNOPT library (include/problem.h:6 pulls mongo/client/dbclient.h into
 * every one of them), which the build rules forbid.  What pins this restatement:
 *   (1) the known-answer values SURVEY.md section 8(c) recorded from the reference's own compiled code
 *       (tests/golden/survey_known_answers.json; tests/test_oracle_known_answers.py),
 *   (2) the reference's own .param reader, src/parameters.cpp, which does compile from its own file
 *       and is built in place into oracle/_ref/ (oracle/Makefile target `ref`) to check the reader,
 *   (3) an independent central finite-difference check of G against F (tests/test_oracle_fd.py),
 *       which must fail exactly on the reference quirks SURVEY.md Appendix B lists and nowhere else,
 *   (4) numbers obtained by EXECUTING the text of the reference's own functions -- constructors included since
 *       round 3 -- with this repository's C-subset interpreter (tools/make_ref_vectors.py, tools/refeval/;
 *       tests/golden/ref_eval_*.npz; tests/test_ref_eval_*.py): x0, bounds, pattern, F and G for both missions,
 *       five air-frames, wind models 0 / 1 / 3 / table, shear wind with seeded (Vref, href), ts = 6 ... 2000.
 *       The executor is this repository's own, so by the build rules this does not count as the compiled reference:
 *       agreement with the reference's BINARY beyond (1) stays unpinned.
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef TOLFG_ORACLE_H_
#define TOLFG_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_S10 = 0, ORC_G7 = 1 };
enum { ORC_WIND_NONE = 0, ORC_WIND_SHEAR = 1, ORC_WIND_GRID = 3, ORC_WIND_TABLE = 99 };

/* numinp / numstates: problems/{S10,G7}/snopt.param:3-4 */
enum { ORC_NI = 11, ORC_NS = 8 };

typedef struct {
    int    mission;            /* ORC_S10 | ORC_G7                        src/tol.cpp:9-17 */
    int    N;                  /* ts, number of time segments             include/parameters.h:66 */
    /* aircraft coefficients used on the path                             include/parameters.h:25-30 */
    double mm, SS, Cd0, AR, ee;
    /* gains                                                              include/parameters.h:45-49 */
    double kT, kp, kv, kdt;
    /* goal in NED: xg=north_goal, yg=east_goal, rg=radius_goal           src/problem.cpp:24-27 */
    double xg, yg, rg;
    /* G7 desired course = atan2(yg-yi, xg-xi)                            src/problemG7.cpp:524 */
    double chi_d;
    /* wind: model 0/1 (src/problem.cpp:480-531) or an injected per-node table (the `default: break`
     * arm, src/problem.cpp:732-735, leaves the member arrays as the caller filled them) */
    int    windmodel;
    double Vref, href;         /* 2.4, 10 in the reference                src/problem.cpp:504-505 */
    /* table wind, ENU convention, field-major like the reference's 12 std::vectors
     * (include/problem.h:103): wind[f*(N+1)+k], f = 0..11 in the order
     * u v w du_dx du_dy du_dz dv_dx dv_dy dv_dz dw_dx dw_dy dw_dz */
    const double *wind;
    /* wind model 3, the gridded storm field (src/problem.cpp:544-695): a regular ENU grid of the
     * v (north) component, the only one the reference interpolates; gv[(i*gny + j)*gnz + k] is the
     * value at east gx0+i*gdx, north gy0+j*gdy, up gz0+k*gdz; aircraft position in the grid frame =
     * NED position mapped to ENU plus (gE, gN, gU) (EastFromDatum..., src/problem.cpp:411-413).
     * PARITY UNPINNED: no wind data and no reference output exist for this model. */
    int    gnx, gny, gnz;
    double gx0, gy0, gz0, gdx, gdy, gdz, gE, gN, gU;
    const double *gv;
} orc_problem;

/* .param reader: src/parameters.cpp:14-34.  Returns the number of values parsed (<= maxn) or -1
 * when the file cannot be opened (the reference then sees zero values and throws length_error). */
int  orc_read_params(const char *path, double *out, int maxn);

/* sizes: src/problem.cpp:151-152; neG closed forms SURVEY.md section 8 */
int  orc_nb (int mission);
int  orc_n  (int N);
int  orc_neF(int mission, int N);
int  orc_neG(int mission, int N);
int  orc_c0 (int mission, int N);   /* index in G of node 0's 104-double slab */

/* Jacobian sparsity pattern, 0-based, in the order countG emits it (src/problem.cpp:813-919). */
void orc_pattern_closed(int mission, int N, int *iGfun, int *jGvar);
/* Same pattern by walking every (row, col) like countG does -- O(neF*n), small N only.
 * Also returns the per-entry dispatch data computeG uses (src/problem.cpp:870-875,905-908);
 * any of Fs/xs/tfs/txs may be NULL.  Returns neG. */
int  orc_pattern_walk(int mission, int N, int *iGfun, int *jGvar,
                      int *Fs, int *xs, int *tfs, int *txs);

/* initial guess: src/problemS10.cpp:19-219, src/problemG7.cpp:19-217.  (xi,yi,zi) is the start
 * position the reference hard-codes to 0 (src/problem.cpp:83-85,111-113). */
void orc_x0(const orc_problem *p, double xi, double yi, double zi, double *x);

/* bounds: src/problem.cpp:198-365 with the node-0 constants of src/problem.cpp:80-134.
 * ac15 = the 15 aircraft values as read (angles still in degrees), lim8 = limits.param values. */
void orc_bounds(int mission, int N, const double *ac15, const double *lim8,
                double xi, double yi, double zi,
                double *xlow, double *xupp, double *Flow, double *Fupp);

/* One evaluation of the user function, fused per node (same math, evaluated once per node). */
void orc_eval(const orc_problem *p, const double *x, int needF, double *F, int needG, double *G);

/* One evaluation in the reference's own evaluation order: computeF then computeG's loop over the
 * neG entries, each entry rebuilding its whole row (src/problem.cpp:782-806,1035-1208).
 * Fs/xs/tfs/txs come from orc_pattern_walk or orc_dispatch_closed. */
void orc_eval_entrywise(const orc_problem *p, const double *x, int needF, double *F,
                        int needG, double *G, int neG,
                        const int *Fs, const int *xs, const int *tfs, const int *txs);
/* closed-form dispatch data (what the walk yields), O(neG) */
void orc_dispatch_closed(int mission, int N, int *Fs, int *xs, int *tfs, int *txs);

/* Batch of B independent trajectories sharing mission/N (problem b = probs[b]); rows of X/F/G are
 * ldx/ldf/ldg doubles apart.  nthreads > 1 uses OpenMP over trajectories.  Returns threads used. */
int  orc_eval_batch(const orc_problem *probs, int B, const double *X, int ldx,
                    double *F, int ldf, double *G, int ldg, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
