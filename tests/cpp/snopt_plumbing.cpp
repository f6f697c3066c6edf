// snopt_plumbing.cpp -- BASELINE configs[0] "plumbing": a C++ SNOPT-side driver that knows nothing
// of this repo except include/tolfg.h enters the callback exactly the way tol's vendored wrapper
// does (ref: snoptProblemA::solve, src/snoptProblem.cpp:448-488) -- function pointer of type snFunA,
// 1-based pattern while "inside SNOPT", needF = needG = 1, user workspace empty.
// SNOPT itself is commercial and absent; this stands where f_snkera would call usrfun once.
// Built and run by tests/test_cpp_plumbing.py:  g++ ... -ltolfg -lamdhip64
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tolfg.h"

extern "C" {
// ref: include/snopt/snopt.h:60-66
typedef void (*snFunA)(int *Status, int *n, double x[], int *needF, int *neF, double F[], int *needG, int *neG,
                       double G[], char cu[], int *lencu, int iu[], int *leniu, double ru[], int *lenru);
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    tolfg_config cfg;
    tolfg_config_default(&cfg);
    cfg.mission = argv[1];
    cfg.aircraft = argv[2];
    cfg.ts = std::atoi(argv[3]);
    cfg.east_goal = 400; cfg.north_goal = 0; cfg.up_goal = 70; cfg.up = 100;
    cfg.radius_goal = cfg.mission[0] == 'S' ? 100 : 0;
    cfg.persistent_arrays = 1;                // like snOptA: the same x, F, G every call, until tolfg_forget_arrays
    tolfg_problem *p = nullptr;
    if (tolfg_create(&cfg, &p) != TOLFG_OK) { std::fprintf(stderr, "create: %s\n", tolfg_last_error()); return 1; }
    tolfg_set_current(p);

    int n, neF, neG;
    tolfg_sizes(p, &n, &neF, &neG);
    std::vector<int> iGfun(neG), jGvar(neG);
    tolfg_pattern(p, iGfun.data(), jGvar.data());
    std::vector<double> x(n), F(neF, -7.0), G(neG, -7.0);
    tolfg_x0(p, x.data());

    snFunA usrfun = DEFINEGusrfg_;            // must be assignable without a cast
    for (int i = 0; i < neG; i++) { iGfun[i]++; jGvar[i]++; }       // src/snoptProblem.cpp:460-465
    int Status = 1, needF = 1, needG = 1, lencu = 0, leniu = 0, lenru = 0;
    for (int call = 0; call < 3; call++)      // the second call in a row pins the arrays, the third runs in place
        usrfun(&Status, &n, x.data(), &needF, &neF, F.data(), &needG, &neG, G.data(), nullptr, &lencu, nullptr, &leniu,
               nullptr, &lenru);
    if (Status == 1 && tolfg_registered_arrays(p) < 2) { std::fprintf(stderr, "arrays were not used in place\n"); return 4; }
    if (tolfg_forget_arrays(p) != TOLFG_OK || tolfg_registered_arrays(p) != 0) return 5;
    for (int i = 0; i < neG; i++) { iGfun[i]--; jGvar[i]--; }
    std::printf("status %d n %d neF %d neG %d\n", Status, n, neF, neG);
    for (int i = 0; i < neF; i++) std::printf("F %.17g\n", F[i]);
    for (int i = 0; i < neG; i++) std::printf("G %.17g\n", G[i]);
    tolfg_destroy(p);
    return Status == 1 ? 0 : 3;
}
