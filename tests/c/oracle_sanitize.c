/* oracle_sanitize.c -- runs the oracle's entry points under -fsanitize=address,undefined (CPU only). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "tolfg_oracle.h"

int main(void)
{
    int rc = 0;
    for (int mission = 0; mission < 2; mission++)
        for (int wi = 0; wi < 4; wi++) {
            const int Ns[4] = {1, 7, 64, 203};
            for (int ni = 0; ni < 4; ni++) {
                const int N = Ns[ni];
                orc_problem p = {0};
                p.mission = mission; p.N = N;
                p.mm = 6.1228; p.SS = 0.6316; p.Cd0 = 0.03; p.AR = 16.4457; p.ee = 0.9693;
                p.kT = 0.3; p.kp = 8; p.kv = 0.2; p.kdt = 1;
                p.xg = 0; p.yg = 400; p.rg = mission == ORC_S10 ? 100 : 0; p.chi_d = atan2(400.0, 0.0);
                p.Vref = 2.4; p.href = 10;
                const int n = orc_n(N), neF = orc_neF(mission, N), neG = orc_neG(mission, N);
                double *x = malloc(sizeof(double) * n), *F = malloc(sizeof(double) * neF), *G = malloc(sizeof(double) * neG);
                double *F2 = malloc(sizeof(double) * neF), *G2 = malloc(sizeof(double) * neG);
                double *wind = malloc(sizeof(double) * 12 * (N + 1)), *grid = malloc(sizeof(double) * 5 * 4 * 3);
                int *iG = malloc(sizeof(int) * (neG + 64)), *jG = malloc(sizeof(int) * (neG + 64));
                int *d[4];
                for (int q = 0; q < 4; q++) d[q] = malloc(sizeof(int) * (neG + 64));
                for (int i = 0; i < 12 * (N + 1); i++) wind[i] = 0.01 * (i % 17) - 0.05;
                for (int i = 0; i < 60; i++) grid[i] = 0.3 * (i % 11) - 1.0;
                const int wm[4] = {ORC_WIND_NONE, ORC_WIND_SHEAR, ORC_WIND_TABLE, ORC_WIND_GRID};
                p.windmodel = wm[wi]; p.wind = wind;
                p.gnx = 5; p.gny = 4; p.gnz = 3; p.gx0 = -300; p.gy0 = -300; p.gz0 = -100; p.gdx = p.gdy = p.gdz = 150;
                p.gE = 10; p.gN = 20; p.gU = 30; p.gv = grid;
                orc_x0(&p, 1.0, -2.0, -30.0, x);
                orc_pattern_closed(mission, N, iG, jG);
                if (N <= 7 && orc_pattern_walk(mission, N, iG, jG, d[0], d[1], d[2], d[3]) != neG) rc = 1;
                orc_dispatch_closed(mission, N, d[0], d[1], d[2], d[3]);
                orc_eval(&p, x, 1, F, 1, G);
                orc_eval_entrywise(&p, x, 1, F2, 1, G2, neG, d[0], d[1], d[2], d[3]);
                for (int i = 0; i < neF; i++) if (F[i] != F2[i] && !(isnan(F[i]) && isnan(F2[i]))) rc = 2;
                for (int i = 0; i < neG; i++) if (G[i] != G2[i] && !(isnan(G[i]) && isnan(G2[i]))) rc = 3;
                free(x); free(F); free(G); free(F2); free(G2); free(wind); free(grid); free(iG); free(jG);
                for (int q = 0; q < 4; q++) free(d[q]);
            }
        }
    double out[64];
    if (orc_read_params("/nonexistent/file.param", out, 64) != -1) rc = 4;
    printf("oracle sanitize run rc=%d\n", rc);
    return rc;
}
