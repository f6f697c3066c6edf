/* batch_montecarlo_multi.c -- plain C99 use of the several-GPUs-one-process ABI (include/tolfg.h, section 4):
 * a Monte-Carlo batch of B loiter trajectories that differ in their shear wind, sharded over the GPUs named on the
 * command line, evaluated with one launch per device, the objectives all-gathered over RCCL -- once synchronously, then in a
 * pipelined loop (tolfg_multi_step / tolfg_multi_gather_wait).
 *
 *   gcc -std=c99 -O2 -I include examples/batch_montecarlo_multi.c -L tol_amd/lib -ltolfg -L /opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/tol_amd/lib -Wl,-rpath,/opt/rocm/lib -o mc_multi && ./mc_multi 1024 0 1 2 3 4 5 6 7
 *
 * The program itself makes no HIP call: buffers, streams, threads and communicators belong to the library (librccl is
 * resolved at run time, beside the libamdhip64 the executable links).
 */
#include <stdio.h>
#include <stdlib.h>

#include "tolfg.h"

#define TK(x) do { if ((x) != TOLFG_OK) { fprintf(stderr, "%s: %s\n", #x, tolfg_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
    const long B = argc > 1 ? atol(argv[1]) : 1024;
    int devices[64], nd = 0, i;
    long t;
    for (i = 2; i < argc && nd < 64; i++) devices[nd++] = atoi(argv[i]);
    if (nd == 0) devices[nd++] = 0;

    const char *airframes[1] = {"tempest"};
    tolfg_batch_config cfg;
    cfg.mission = "S10"; cfg.root_path = NULL; cfg.aircraft = airframes; cfg.n_aircraft = 1;
    cfg.ts = 200; cfg.windmodel = TOLFG_WIND_SHEAR; cfg.dtype = TOLFG_F64; cfg.device = 0;
    cfg.pattern = TOLFG_PATTERN_REFERENCE;
    tolfg_multi *m = NULL;
    TK(tolfg_multi_create(&cfg, devices, nd, &m));

    tolfg_traj *tr = (tolfg_traj *)calloc((size_t)B, sizeof(tolfg_traj));
    for (t = 0; t < B; t++) {
        tr[t].aircraft = 0;
        tr[t].Vref = 5.0 * (double)t / (double)(B > 1 ? B - 1 : 1);      /* 0 .. 5 m/s at href */
        tr[t].href = 10.0;
        tr[t].north_goal = 0.0; tr[t].east_goal = 400.0; tr[t].radius_goal = 100.0;
        tr[t].zi = -50.0;
    }
    TK(tolfg_multi_set_trajectories(m, B, tr));
    for (i = 0; i < nd; i++) {
        long lo, hi;
        TK(tolfg_multi_shard(m, i, &lo, &hi));
        printf("device %d: trajectories [%ld, %ld)\n", devices[i], lo, hi);
    }
    TK(tolfg_multi_x0(m));                         /* initial guesses, generated on the devices */
    TK(tolfg_multi_eval(m, 1, 1));                 /* F, G and the objectives of every shard: one launch per device */
    double *obj = (double *)malloc(sizeof(double) * (size_t)B), mean = 0.0;
    TK(tolfg_multi_gather_objectives(m, obj));     /* ncclAllGather over the devices; global trajectory order */
    TK(tolfg_multi_mean_objective(m, &mean));      /* ncclAllReduce of the per-device partial sums */

    /* The same as a loop a Monte-Carlo driver would run: step i+1 is launched before the objectives of step i are asked for, so
     * the gather of one step runs beside the launch of the next (four objective buffers in rotation inside the library). */
    {
        const int steps = 6;
        unsigned long ticket[2];
        double *last = (double *)malloc(sizeof(double) * (size_t)B);
        int s, same = 1;
        TK(tolfg_multi_step(m, NULL, 0, 1, &ticket[0]));
        for (s = 1; s < steps; s++) {
            TK(tolfg_multi_step(m, NULL, 0, 1, &ticket[s & 1]));
            TK(tolfg_multi_gather_wait(m, ticket[(s - 1) & 1], last));      /* step s is already running */
        }
        TK(tolfg_multi_gather_wait(m, ticket[(steps - 1) & 1], last));
        for (t = 0; t < B; t++) same = same && last[t] == obj[t];          /* the same x every step: the same objectives, bit for bit */
        printf("pipelined %d steps, objectives %s\n", steps, same ? "identical" : "DIFFERENT");
        free(last);
        if (!same) return 2;
    }

    double lo = obj[0], hi = obj[0];
    for (t = 0; t < B; t++) { if (obj[t] < lo) lo = obj[t]; if (obj[t] > hi) hi = obj[t]; }
    printf("B %ld devices %d rccl %s objective min %.17g mean %.17g max %.17g first %.17g last %.17g\n", B, nd,
           tolfg_multi_rccl_library(), lo, mean, hi, obj[0], obj[B - 1]);
    free(obj); free(tr);
    tolfg_multi_destroy(m);
    return 0;
}
