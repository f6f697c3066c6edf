/* batch_montecarlo.c -- plain C99 use of the batched ABI (include/tolfg.h): evaluate F and G of B
 * loiter trajectories that differ in their shear wind, on one GPU, and print the spread of the
 * objective.  Device memory comes from the HIP runtime directly; no Python, no C++.
 *
 *   gcc -std=c99 -O2 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ examples/batch_montecarlo.c \
 *       -L tol_amd/lib -ltolfg -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/tol_amd/lib -Wl,-rpath,/opt/rocm/lib \
 *       -o batch_montecarlo && ./batch_montecarlo 256
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "tolfg.h"

#define CK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP failure at %s\n", #x); return 1; } } while (0)
#define TK(x) do { if ((x) != TOLFG_OK) { fprintf(stderr, "%s: %s\n", #x, tolfg_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 256;
    const char *airframes[1] = {"tempest"};
    tolfg_batch_config cfg;
    cfg.mission = "S10"; cfg.root_path = NULL; cfg.aircraft = airframes; cfg.n_aircraft = 1;
    cfg.ts = 200; cfg.windmodel = TOLFG_WIND_SHEAR; cfg.dtype = TOLFG_F64; cfg.device = 0;
    cfg.pattern = TOLFG_PATTERN_REFERENCE;
    tolfg_batch *bt = NULL;
    TK(tolfg_batch_create(&cfg, &bt));
    int n, neF, neG;
    TK(tolfg_batch_sizes(bt, &n, &neF, &neG));

    tolfg_traj *tr = (tolfg_traj *)calloc((size_t)B, sizeof(tolfg_traj));
    for (int t = 0; t < B; t++) {
        tr[t].aircraft = 0;
        tr[t].Vref = 5.0 * t / (B > 1 ? B - 1 : 1);      /* 0 .. 5 m/s at href */
        tr[t].href = 10.0;
        tr[t].north_goal = 0.0; tr[t].east_goal = 400.0; tr[t].radius_goal = 100.0;
        tr[t].zi = -50.0;
    }
    TK(tolfg_batch_set_trajectories(bt, B, tr));

    const long ldx = (n + 1) & ~1L, ldf = (neF + 1) & ~1L;
    long ldg = 0;
    double *dX, *dF, *dG, *dObj;
    CK(hipMalloc((void **)&dX, sizeof(double) * B * ldx));
    CK(hipMalloc((void **)&dF, sizeof(double) * B * ldf));
    /* G, nine tenths of what a launch writes, from the library: where it lands in HBM decides how fast a launch beyond the cache runs
       (include/tolfg.h, "Where the outputs live"); up to 12 candidates are timed, the best kept */
    TK(tolfg_batch_alloc_outputs(bt, B, 12, (void **)&dG, &ldg, NULL, NULL));
    CK(hipMalloc((void **)&dObj, sizeof(double) * B));
    TK(tolfg_batch_x0_device(bt, B, dX, ldx, NULL));                       /* initial guess, on the device */
    TK(tolfg_batch_eval(bt, B, dX, ldx, dF, ldf, dG, ldg, NULL, 1, 1, dObj, NULL));
    double *obj = (double *)malloc(sizeof(double) * B);
    CK(hipMemcpy(obj, dObj, sizeof(double) * B, hipMemcpyDeviceToHost));    /* synchronises */

    double lo = obj[0], hi = obj[0], sum = 0;
    for (int t = 0; t < B; t++) { if (obj[t] < lo) lo = obj[t]; if (obj[t] > hi) hi = obj[t]; sum += obj[t]; }
    printf("B %d n %d neF %d neG %d objective min %.17g mean %.17g max %.17g first %.17g last %.17g\n", B, n, neF, neG, lo, sum / B, hi,
           obj[0], obj[B - 1]);
    free(obj); free(tr);
    (void)hipFree(dX); (void)hipFree(dF); (void)hipFree(dObj);
    TK(tolfg_device_free(dG));
    tolfg_batch_destroy(bt);
    return 0;
}
