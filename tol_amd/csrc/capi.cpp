// capi.cpp -- the extern "C" surface declared in include/tolfg.h.  No exception leaves this file:
// the SNOPT callback is entered from a Fortran frame (ref: f_snkera, src/snoptProblem.cpp:468).
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/tolfg.h"
#include "knobs.h"
#include "problem.h"
#include "multi.h"

using namespace tolfg;

struct tolfg_problem {
    problem *p;
    int index;
};
struct tolfg_batch {
    batch *b;
};
struct tolfg_multi {
    multi *m;
    int place_tries;
};

namespace {

thread_local std::string g_err;
std::mutex g_reg_mu;
std::vector<tolfg_problem *> g_registry;     // handle index -> problem, for the iu[] route
tolfg_problem *g_current = nullptr;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

template <typename Fn>
int guarded(Fn &&fn)
{
    try {
        g_err.clear();
        fn();
        return TOLFG_OK;
    } catch (const std::length_error &e) {
        return fail(TOLFG_ERR_PARAM, e.what());
    } catch (const hip_failure &e) {
        return fail(TOLFG_ERR_HIP, e.what());
    } catch (const std::invalid_argument &e) {
        return fail(TOLFG_ERR_ARG, e.what());
    } catch (const std::out_of_range &e) {
        return fail(TOLFG_ERR_ARG, e.what());
    } catch (const std::bad_alloc &) {
        return fail(TOLFG_ERR_HIP, "out of host memory");
    } catch (const std::exception &e) {
        return fail(TOLFG_ERR_HIP, e.what());
    } catch (...) {
        return fail(TOLFG_ERR_HIP, "unknown failure");
    }
}

}  // namespace

extern "C" {

const char *tolfg_last_error(void) { return g_err.c_str(); }
const char *tolfg_version(void) { return measurement_build() ? "tolfg-mi355x 0.2 (gfx950, measurement build)" : "tolfg-mi355x 0.2 (gfx950)"; }
int tolfg_measurement_build(void) { return measurement_build() ? 1 : 0; }

const char *tolfg_default_root(void)
{
    static const std::string root = default_root();
    return root.c_str();
}

int tolfg_read_params(const char *path, double *out, int maxn)
{
    if (!path || (maxn > 0 && !out)) return fail(TOLFG_ERR_ARG, "tolfg_read_params: null argument");
    std::vector<double> v;
    if (!readparams(path, v)) return fail(TOLFG_ERR_PARAM, std::string("cannot open ") + path);
    for (int i = 0; i < maxn && i < (int)v.size(); ++i) out[i] = v[i];
    return (int)v.size();
}

void tolfg_config_default(tolfg_config *cfg)
{
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->mission = "S10";
    cfg->aircraft = "tempest";
    cfg->root_path = nullptr;
    cfg->windmodel = TOLFG_WIND_SHEAR;   // what the reference runs offline (src/problem.cpp:73-78)
    cfg->Vref = 2.4;                     // src/problem.cpp:504
    cfg->href = 10.0;                    // src/problem.cpp:505
}

int tolfg_create(const tolfg_config *cfg, tolfg_problem **out)
{
    if (!cfg || !out) return fail(TOLFG_ERR_ARG, "tolfg_create: null argument");
    *out = nullptr;
    return guarded([&] {
        if (!cfg->mission) throw std::invalid_argument("mission is required");
        problem *p = nullptr;
        switch (mission_from_name(cfg->mission)) {      // ref: mission_select, src/tol.cpp:5-24
        case MISSION_S10: p = new problemS10(*cfg); break;
        case MISSION_G7:  p = new problemG7(*cfg); break;
        }
        tolfg_problem *h = new tolfg_problem{p, -1};
        std::lock_guard<std::mutex> lk(g_reg_mu);
        h->index = (int)g_registry.size();
        g_registry.push_back(h);
        *out = h;
    });
}

void tolfg_destroy(tolfg_problem *h)
{
    if (!h) return;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        if (h->index >= 0 && h->index < (int)g_registry.size()) g_registry[h->index] = nullptr;
        if (g_current == h) g_current = nullptr;
    }
    delete h->p;
    delete h;
}

int tolfg_sizes(const tolfg_problem *h, int *n, int *neF, int *neG)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null problem");
    if (n) *n = h->p->n;
    if (neF) *neF = h->p->neF;
    if (neG) *neG = h->p->neG;
    return TOLFG_OK;
}

int tolfg_pattern(const tolfg_problem *h, int *iGfun, int *jGvar)
{
    if (!h || !iGfun || !jGvar) return fail(TOLFG_ERR_ARG, "null argument");
    std::memcpy(iGfun, h->p->iGfun.data(), sizeof(int) * h->p->neG);
    std::memcpy(jGvar, h->p->jGvar.data(), sizeof(int) * h->p->neG);
    return TOLFG_OK;
}

int tolfg_x0(const tolfg_problem *h, double *x)
{
    if (!h || !x) return fail(TOLFG_ERR_ARG, "null argument");
    std::memcpy(x, h->p->x.data(), sizeof(double) * h->p->n);
    return TOLFG_OK;
}

int tolfg_bounds(const tolfg_problem *h, double *xlow, double *xupp, double *Flow, double *Fupp)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null problem");
    if (xlow) std::memcpy(xlow, h->p->xlow.data(), sizeof(double) * h->p->n);
    if (xupp) std::memcpy(xupp, h->p->xupp.data(), sizeof(double) * h->p->n);
    if (Flow) std::memcpy(Flow, h->p->Flow.data(), sizeof(double) * h->p->neF);
    if (Fupp) std::memcpy(Fupp, h->p->Fupp.data(), sizeof(double) * h->p->neF);
    return TOLFG_OK;
}

int tolfg_tolerances(const tolfg_problem *h, double *opt_tol, double *feas_tol)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null problem");
    if (opt_tol) *opt_tol = h->p->sn.opt_tol;
    if (feas_tol) *feas_tol = h->p->sn.feas_tol;
    return TOLFG_OK;
}

int tolfg_set_wind_table(tolfg_problem *h, const double *wind_enu)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null problem");
    return guarded([&] { h->p->set_wind_table(wind_enu); });
}

int tolfg_set_wind_grid(tolfg_problem *h, const tolfg_wind_grid *grid)
{
    if (!h || !grid) return fail(TOLFG_ERR_ARG, "null argument");
    return guarded([&] { h->p->set_wind_grid(*grid); });
}

int tolfg_write_json(const tolfg_problem *h, const double *x, double final_cost, const char *filename)
{
    if (!h || !x || !filename) return fail(TOLFG_ERR_ARG, "null argument");
    return guarded([&] { h->p->writeJSON(filename, x, final_cost); });
}

void tolfg_set_current(tolfg_problem *h)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_current = h;
    prob = h ? h->p : nullptr;
}

tolfg_problem *tolfg_get_current(void) { return g_current; }

int tolfg_handle_index(const tolfg_problem *h) { return h ? h->index : -1; }

void DEFINEGusrfg_(int *Status, int *n, double x[], int *needF, int *neF, double F[], int *needG, int *neG,
                   double G[], char *cu, int *lencu, int iu[], int *leniu, double ru[], int *lenru)
{
    (void)cu; (void)lencu; (void)ru; (void)lenru;
    problem *p = prob;
    if (iu && leniu && *leniu >= 2 && iu[0] == TOLFG_IU_MAGIC) {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        const int idx = iu[1];
        p = (idx >= 0 && idx < (int)g_registry.size() && g_registry[idx]) ? g_registry[idx]->p : nullptr;
    }
    const bool wantF = needF && *needF > 0, wantG = needG && *needG > 0;
    int rc;
    if (!p) {
        rc = fail(TOLFG_ERR_NOCURRENT, "DEFINEGusrfg_: no current problem (call tolfg_set_current)");
    } else if (!n || !x || *n != p->n || (wantF && (!neF || !F || *neF != p->neF)) ||
               (wantG && (!neG || !G || *neG != p->neG))) {
        rc = fail(TOLFG_ERR_ARG, "DEFINEGusrfg_: array sizes do not match the current problem");
    } else {
        rc = guarded([&] { p->evaluate(x, wantF, F, wantG, G); });
    }
    if (rc != TOLFG_OK) {
        std::fprintf(stderr, "tolfg: %s\n", g_err.c_str());
        if (Status) *Status = -2;   // snOptA: a value <= -2 asks SNOPT to terminate
    }
}

int tolfg_register_arrays(tolfg_problem *h, double *x, double *F, double *G)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null problem");
    return guarded([&] { h->p->register_arrays(x, F, G); });
}

int tolfg_forget_arrays(tolfg_problem *h)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null problem");
    return guarded([&] { h->p->forget_arrays(); });
}

int tolfg_registered_arrays(const tolfg_problem *h)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null problem");
    return h->p->registered_arrays();
}

int tolfg_time_callback(tolfg_problem *h, const double *x, double *F, double *G, int needF, int needG, int warm, int calls,
                        double *us_per_call)
{
    return tolfg_time_callback_as(h, x, F, G, needF, needG, 1, warm, calls, us_per_call);
}

int tolfg_time_callback_as(tolfg_problem *h, const double *x, double *F, double *G, int needF, int needG, int in_place, int warm,
                           int calls, double *us_per_call)
{
    if (!h || !x || !F || !G || calls < 1 || !us_per_call) return fail(TOLFG_ERR_ARG, "tolfg_time_callback: bad argument");
    typedef void (*snFunA)(int *, int *, double *, int *, int *, double *, int *, int *, double *, char *, int *, int *, int *,
                           double *, int *);     // ref: include/snopt/snopt.h:60-66
    volatile snFunA usrfun = DEFINEGusrfg_;
    tolfg_problem *keep = g_current;
    tolfg_set_current(h);
    int Status = 1, n = h->p->n, neF = h->p->neF, neG = h->p->neG, zero = 0;
    int wantF = needF ? 1 : 0, wantG = needG ? 1 : 0;
    // the caller's arrays, the same ones every call, as snOptA's are: used in place while this function runs (where
    // aligned) and forgotten before it returns, so nothing stays pinned that the caller may free
    double *xs = const_cast<double *>(x);          // snFunA takes double x[]; the callback only reads it
    auto al = [](const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    // in_place = 0: the default contract -- nothing is registered, every call copies x into and F, G out of the library's
    // pinned staging buffers
    const int rc = in_place ? guarded([&] { h->p->register_arrays(al(xs) ? xs : nullptr, al(F) ? F : nullptr, al(G) ? G : nullptr); })
                            : guarded([&] { h->p->forget_arrays(); });
    if (rc != TOLFG_OK) { tolfg_set_current(keep); return rc; }
    for (int i = 0; i < warm; ++i) usrfun(&Status, &n, xs, &wantF, &neF, F, &wantG, &neG, G, nullptr, &zero, nullptr, &zero, nullptr, &zero);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < calls; ++i) usrfun(&Status, &n, xs, &wantF, &neF, F, &wantG, &neG, G, nullptr, &zero, nullptr, &zero, nullptr, &zero);
    const auto t1 = std::chrono::steady_clock::now();
    *us_per_call = std::chrono::duration<double, std::micro>(t1 - t0).count() / calls;
    (void)guarded([&] { h->p->forget_arrays(); });
    tolfg_set_current(keep);
    return Status;
}

int tolfg_modelWind(tolfg_problem *h, const double *x)
{
    if (!h || !x) return fail(TOLFG_ERR_ARG, "null argument");
    return guarded([&] { h->p->modelWind(x); });
}

int tolfg_computeF(tolfg_problem *h, const double *x, double *F)
{
    if (!h || !x || !F) return fail(TOLFG_ERR_ARG, "null argument");
    return guarded([&] { h->p->computeF(x, F); });
}

int tolfg_computeG(tolfg_problem *h, const double *x, double *G)
{
    if (!h || !x || !G) return fail(TOLFG_ERR_ARG, "null argument");
    return guarded([&] { h->p->computeG(x, G); });
}

// ------------------------------------------------------------------------------------ batch

int tolfg_batch_create(const tolfg_batch_config *cfg, tolfg_batch **out)
{
    if (!cfg || !out) return fail(TOLFG_ERR_ARG, "tolfg_batch_create: null argument");
    *out = nullptr;
    return guarded([&] {
        if (!cfg->mission || !cfg->aircraft || cfg->n_aircraft < 1)
            throw std::invalid_argument("mission and at least one aircraft are required");
        std::vector<std::string> names;
        for (int i = 0; i < cfg->n_aircraft; ++i) {
            if (!cfg->aircraft[i]) throw std::invalid_argument("null aircraft name");
            names.emplace_back(cfg->aircraft[i]);
        }
        const std::string root = cfg->root_path ? std::string(cfg->root_path) : default_root();
        *out = new tolfg_batch{new batch(cfg->mission, root, names, cfg->ts, cfg->windmodel, cfg->dtype, cfg->device,
                                        cfg->pattern)};
    });
}

void tolfg_batch_destroy(tolfg_batch *h)
{
    if (!h) return;
    delete h->b;
    delete h;
}

int tolfg_batch_sizes(const tolfg_batch *h, int *n, int *neF, int *neG)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null batch");
    const Sizes &s = h->b->sizes();
    if (n) *n = s.n;
    if (neF) *neF = s.neF;
    if (neG) *neG = s.neG;
    return TOLFG_OK;
}

int tolfg_batch_pattern(const tolfg_batch *h, int *iGfun, int *jGvar)
{
    if (!h || !iGfun || !jGvar) return fail(TOLFG_ERR_ARG, "null argument");
    if (h->b->mission() == MISSION_MIXED) return fail(TOLFG_ERR_ARG, "a mixed batch has one pattern per mission: tolfg_batch_mission_pattern");
    make_pattern(h->b->sizes(), iGfun, jGvar);
    return TOLFG_OK;
}

int tolfg_batch_mission_sizes(const tolfg_batch *h, int mission, int *n, int *neF, int *neG)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null batch");
    return guarded([&] {
        const Sizes &s = h->b->sizes_of(mission);
        if (n) *n = s.n;
        if (neF) *neF = s.neF;
        if (neG) *neG = s.neG;
    });
}

int tolfg_batch_mission_pattern(const tolfg_batch *h, int mission, int *iGfun, int *jGvar)
{
    if (!h || !iGfun || !jGvar) return fail(TOLFG_ERR_ARG, "null argument");
    return guarded([&] { make_pattern(h->b->sizes_of(mission), iGfun, jGvar); });
}

int tolfg_batch_set_trajectories(tolfg_batch *h, int B, const tolfg_traj *trajs)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null batch");
    return guarded([&] { h->b->set_trajectories(B, trajs); });
}

int tolfg_batch_set_wind_grid(tolfg_batch *h, const tolfg_wind_grid *grid)
{
    if (!h || !grid) return fail(TOLFG_ERR_ARG, "null argument");
    return guarded([&] { h->b->set_wind_grid(*grid); });
}

int tolfg_batch_x0(const tolfg_batch *h, int t, double zi, double *x)
{
    if (!h || !x) return fail(TOLFG_ERR_ARG, "null argument");
    return guarded([&] {
        const tolfg_traj &tr = h->b->trajectory(t);
        initial_guess(h->b->sizes_of_traj(t), h->b->airframe(tr.aircraft), Start{tr.xi, tr.yi, zi}, h->b->chi_d(t), x);
    });
}

int tolfg_batch_x0_device(tolfg_batch *h, int B, void *dX, long ldx, void *stream)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null batch");
    return guarded([&] { h->b->x0_device(B, dX, ldx, static_cast<hipStream_t>(stream)); });
}

int tolfg_batch_bounds_device(tolfg_batch *h, int B, void *dXlow, void *dXupp, long ldx, void *dFlow, void *dFupp,
                              long ldf, void *stream)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null batch");
    return guarded([&] { h->b->bounds_device(B, dXlow, dXupp, ldx, dFlow, dFupp, ldf, static_cast<hipStream_t>(stream)); });
}

int tolfg_batch_bounds(const tolfg_batch *h, int t, double zi, double *xlow, double *xupp, double *Flow,
                       double *Fupp)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null batch");
    return guarded([&] {
        const Sizes &s = h->b->sizes_of_traj(t);
        const tolfg_traj &tr = h->b->trajectory(t);
        std::vector<double> xl(s.n), xu(s.n), Fl(s.neF), Fu(s.neF);
        set_limits(s, h->b->airframe(tr.aircraft), h->b->limits(s.mission), Start{tr.xi, tr.yi, zi}, xl.data(), xu.data(),
                   Fl.data(), Fu.data());
        if (xlow) std::memcpy(xlow, xl.data(), sizeof(double) * s.n);
        if (xupp) std::memcpy(xupp, xu.data(), sizeof(double) * s.n);
        if (Flow) std::memcpy(Flow, Fl.data(), sizeof(double) * s.neF);
        if (Fupp) std::memcpy(Fupp, Fu.data(), sizeof(double) * s.neF);
    });
}

int tolfg_batch_eval(tolfg_batch *h, int B, const void *dX, long ldx, void *dF, long ldf, void *dG, long ldg,
                     const void *dWind, int needF, int needG, void *dObj, void *stream)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null batch");
    if (dObj && !needF) return fail(TOLFG_ERR_ARG, "dObj needs needF");
    return guarded([&] {
        h->b->eval(B, dX, ldx, dF, ldf, dG, ldg, dWind, needF, needG, static_cast<hipStream_t>(stream), dObj);
    });
}

int tolfg_batch_objectives(tolfg_batch *h, int B, const void *dF, long ldf, void *dObj, void *stream)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null batch");
    return guarded([&] { h->b->objectives(B, dF, ldf, dObj, static_cast<hipStream_t>(stream)); });
}

int tolfg_batch_status(tolfg_batch *h)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null batch");
    if (h->b->take_lost_partial())
        return fail(TOLFG_ERR_HIP, "an evaluation of this batch lost an objective partial: its F[0] is not a number (does x carry "
                                   "NaNs?); the outputs of that evaluation must not be used");
    return TOLFG_OK;
}

int tolfg_batch_set_timing(tolfg_batch *h, int enable)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null batch");
    return guarded([&] { h->b->set_timing(enable != 0); });
}

int tolfg_device_alloc(int device, size_t bytes, void **ptr)
{
    if (!ptr) return fail(TOLFG_ERR_ARG, "tolfg_device_alloc: null argument");
    *ptr = nullptr;
    return guarded([&] { *ptr = device_alloc(device, bytes); });
}

int tolfg_device_free(void *ptr)
{
    return guarded([&] { device_free(ptr); });
}

int tolfg_batch_alloc_outputs(tolfg_batch *h, int B, int tries, void **dG, long *ldg, double *probe_us, int *tried)
{
    if (!h || !dG) return fail(TOLFG_ERR_ARG, "tolfg_batch_alloc_outputs: null argument");
    *dG = nullptr;
    return guarded([&] { *dG = h->b->alloc_outputs(B, tries, ldg, probe_us, tried); });
}

int tolfg_batch_set_store_shape(tolfg_batch *h, int enable)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null batch");
    return guarded([&] { h->b->set_store_shape(enable != 0); });
}

int tolfg_batch_kernel_time(tolfg_batch *h, double *avg_ms, double *min_ms)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null batch");
    int n = 0;
    const int rc = guarded([&] { n = h->b->kernel_time(avg_ms, min_ms); });
    return rc == TOLFG_OK ? n : rc;
}

double tolfg_batch_algorithmic_bytes(const tolfg_batch *h, int B)
{
    if (!h || B < 0) return 0.0;
    return h->b->algorithmic_bytes(B);
}

// ------------------------------------------------------------------------------------ several GPUs, one process

int tolfg_multi_create(const tolfg_batch_config *cfg, const int *devices, int n_devices, tolfg_multi **out)
{
    if (!cfg || !out || !devices || n_devices < 1) return fail(TOLFG_ERR_ARG, "tolfg_multi_create: null argument or no device");
    *out = nullptr;
    return guarded([&] {
        if (!cfg->mission || !cfg->aircraft || cfg->n_aircraft < 1)
            throw std::invalid_argument("mission and at least one aircraft are required");
        std::vector<std::string> names;
        for (int i = 0; i < cfg->n_aircraft; ++i) {
            if (!cfg->aircraft[i]) throw std::invalid_argument("null aircraft name");
            names.emplace_back(cfg->aircraft[i]);
        }
        const std::string root = cfg->root_path ? std::string(cfg->root_path) : default_root();
        *out = new tolfg_multi{new multi(cfg->mission, root, names, cfg->ts, cfg->windmodel, cfg->dtype, cfg->pattern,
                                        std::vector<int>(devices, devices + n_devices)), 12};
    });
}

void tolfg_multi_destroy(tolfg_multi *h)
{
    if (!h) return;
    delete h->m;
    delete h;
}

int tolfg_multi_sizes(const tolfg_multi *h, int *n, int *neF, int *neG)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null handle");
    const Sizes &s = h->m->sizes();
    if (n) *n = s.n;
    if (neF) *neF = s.neF;
    if (neG) *neG = s.neG;
    return TOLFG_OK;
}

int tolfg_multi_set_trajectories(tolfg_multi *h, long total, const tolfg_traj *trajs)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null handle");
    return guarded([&] { h->m->set_trajectories(total, trajs, h->place_tries); });
}

int tolfg_multi_shard(const tolfg_multi *h, int i, long *lo, long *hi)
{
    if (!h || !lo || !hi || i < 0 || i >= h->m->devices()) return fail(TOLFG_ERR_ARG, "tolfg_multi_shard: bad argument");
    h->m->shard(i, lo, hi);
    return TOLFG_OK;
}

int tolfg_multi_buffers(const tolfg_multi *h, int i, void **dX, long *ldx, void **dF, long *ldf, void **dG, long *ldg)
{
    if (!h || i < 0 || i >= h->m->devices()) return fail(TOLFG_ERR_ARG, "tolfg_multi_buffers: bad argument");
    h->m->buffers(i, dX, ldx, dF, ldf, dG, ldg);
    return TOLFG_OK;
}

int tolfg_multi_set_wind_grid(tolfg_multi *h, const tolfg_wind_grid *grid)
{
    if (!h || !grid) return fail(TOLFG_ERR_ARG, "null argument");
    return guarded([&] { h->m->set_wind_grid(*grid); });
}

int tolfg_multi_set_wind_tables(tolfg_multi *h, const double *wind_enu)
{
    if (!h || !wind_enu) return fail(TOLFG_ERR_ARG, "null argument");
    return guarded([&] { h->m->set_wind_tables(wind_enu); });
}

int tolfg_multi_x0(tolfg_multi *h)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null handle");
    return guarded([&] { h->m->x0(); });
}

int tolfg_multi_eval(tolfg_multi *h, int needF, int needG)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null handle");
    return guarded([&] { h->m->eval(needF != 0, needG != 0); });
}

int tolfg_multi_set_placement(tolfg_multi *h, int tries)
{
    if (!h || tries < 0) return fail(TOLFG_ERR_ARG, "tolfg_multi_set_placement: bad argument");
    h->place_tries = tries;
    return TOLFG_OK;
}

int tolfg_multi_set_issue(tolfg_multi *h, int mode)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null handle");
    return guarded([&] { h->m->set_issue(mode); });
}

int tolfg_multi_set_gather(tolfg_multi *h, int mode)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null handle");
    return guarded([&] { h->m->set_gather(mode); });
}

int tolfg_multi_eval_from(tolfg_multi *h, const void *const *dX, int n_devices, int needF, int needG)
{
    if (!h || !dX || n_devices != h->m->devices()) return fail(TOLFG_ERR_ARG, "tolfg_multi_eval_from: one X pointer per device is required");
    return guarded([&] { h->m->eval(needF != 0, needG != 0, dX); });
}

int tolfg_multi_gather_begin(tolfg_multi *h, unsigned long *ticket)
{
    if (!h || !ticket) return fail(TOLFG_ERR_ARG, "null argument");
    return guarded([&] { *ticket = h->m->gather_begin(); });
}

int tolfg_multi_gather_wait(tolfg_multi *h, unsigned long ticket, void *host_obj)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null handle");
    return guarded([&] { h->m->gather_wait(ticket, host_obj); });
}

int tolfg_multi_step(tolfg_multi *h, const void *const *dX, int n_devices, int needG, unsigned long *ticket)
{
    if (!h || !ticket || (dX && n_devices != h->m->devices())) return fail(TOLFG_ERR_ARG, "tolfg_multi_step: bad argument");
    return guarded([&] { *ticket = h->m->step(true, needG != 0, dX); });
}

int tolfg_multi_time_steps(tolfg_multi *h, int n_x, const void *const *dX, int needF, int needG, int gather, int warm, int steps,
                           tolfg_multi_timing *out, double *launch_us_per_device)
{
    if (!h || !out) return fail(TOLFG_ERR_ARG, "null argument");
    return guarded([&] {
        const multi::Timing t = h->m->time_steps(n_x, dX, needF != 0, needG != 0, gather != 0, warm, steps, launch_us_per_device);
        out->wall_us_per_step = t.wall_us_per_step;
        out->launch_us_per_step = t.launch_us_per_step;
        out->issue_us_per_step = t.issue_us_per_step;
        out->gather_us = t.gather_us;
        out->devices = h->m->devices();
        out->steps = steps;
        out->issue = h->m->issue();
        out->gather = h->m->gather();
    });
}

int tolfg_multi_rccl_version(void)
{
    int v = 0;
    try {
        const rccl_api &nc = rccl_api::get();
        if (!nc.GetVersion || nc.GetVersion(&v) != 0) v = 0;
    } catch (const std::exception &) {
        v = 0;
    }
    return v;
}

int tolfg_multi_gather_objectives(tolfg_multi *h, void *host_obj)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null handle");
    return guarded([&] { h->m->gather_objectives(host_obj); });
}

int tolfg_multi_mean_objective(tolfg_multi *h, double *mean)
{
    if (!h || !mean) return fail(TOLFG_ERR_ARG, "null argument");
    return guarded([&] { *mean = h->m->mean_objective(); });
}

int tolfg_multi_sync(tolfg_multi *h)
{
    if (!h) return fail(TOLFG_ERR_ARG, "null handle");
    return guarded([&] { h->m->sync(); });
}

const char *tolfg_multi_rccl_library(void)
{
    static std::string path;
    try {
        path = rccl_api::get().path;
    } catch (const std::exception &) {
        path.clear();
    }
    return path.c_str();
}

int tolfg_shard_bounds(long total, int rank, int world, long *lo, long *hi)
{
    if (total < 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi) return fail(TOLFG_ERR_ARG, "tolfg_shard_bounds: bad argument");
    shard_bounds(total, rank, world, lo, hi);
    return TOLFG_OK;
}

int tolfg_compact_gathered(const void *padded, size_t elem_size, long total, int world, void *out)
{
    if (!padded || !out || elem_size == 0 || total < 0 || world < 1) return fail(TOLFG_ERR_ARG, "tolfg_compact_gathered: bad argument");
    compact_gathered(padded, elem_size, total, world, out);
    return TOLFG_OK;
}

}  // extern "C"
