/*
 * tolfg_oracle.c -- CPU restatement of tol's SNOPT user-function path.  TEST INFRASTRUCTURE ONLY.
 * See tolfg_oracle.h for who may use this and for the parity status ("partially pinned").
 *
 * The reference evaluates F in three passes (cost, dynamicConstraints, boundaryConstraints;
 * src/problem.cpp:765-774) and G one entry at a time (src/problem.cpp:782-806).  This file states
 * the same mathematics in vector form:
 *
 *   e_a = (cx cg, sx cg, -sg)   e_g = (cx sg, sx sg, cg)   e_x = (-sx, cx, 0)   e_h = (cx, sx, 0)
 *   v   = W + Va e_a                      ground velocity                 src/problem.cpp:1003-1005
 *   A = JW^T e_a, B = JW^T e_g, C = JW^T e_x, H = JW^T e_h,  JW[i][j] = dW_i/dx_j (NED, frozen)
 *   q   = rho S Va^2 / (2 m),  CD = Cd0 + CL^2/(pi AR e)
 *   f4  = T/m - v.A - g sg - q CD                                         src/problem.cpp:1006
 *   f5  = (v.B - g cg + q CL cphi)/Va                                     src/problem.cpp:1007
 *   f6  = (q CL sphi - v.C)/(Va cg)                                       src/problem.cpp:1008
 *   defect_r = s_{k+1,r} - dt f_r - s_{k,r}                               src/problem.cpp:1012-1019
 *
 * and the Jacobian rows as the derivatives of those with W and JW held constant, which is what the
 * reference's tabulated entries are (src/problem.cpp:1080-1186).
 *
 * Citations are relative to /root/reference/.
 */
#include "tolfg_oracle.h"

#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GRAV 9.81      /* include/problem.h:72 */
#define RHO  1.2682    /* include/problem.h:73 */

/* ------------------------------------------------------------------ .param reader */

int orc_read_params(const char *path, double *out, int maxn)
{
    /* src/parameters.cpp:14-34: per line, the text before the first '/' (the delimiter literal
     * '//' truncates to '/'), std::stod of it; a line stod rejects is skipped. */
    FILE *fp = fopen(path, "r");
    if (!fp) return -1;
    char  *line = NULL;
    size_t cap = 0;
    int    cnt = 0;
    while (getline(&line, &cap, fp) >= 0) {
        size_t len = strlen(line);
        if (len && line[len - 1] == '\n') line[len - 1] = '\0';
        char *slash = strchr(line, '/');
        if (slash) *slash = '\0';
        char *end = NULL;
        errno = 0;
        double v = strtod(line, &end);
        if (end == line) continue;         /* stod: invalid_argument */
        if (errno == ERANGE) continue;     /* stod: out_of_range     */
        if (cnt < maxn) out[cnt] = v;
        cnt++;
    }
    free(line);
    fclose(fp);
    return cnt;
}

/* ------------------------------------------------------------------ sizes and pattern */

int orc_nb(int mission) { return mission == ORC_S10 ? 11 : 12; }     /* snopt.param:5 */
int orc_n(int N) { return ORC_NI * (N + 1) + 1; }                    /* src/problem.cpp:151 */
int orc_neF(int mission, int N) { return ORC_NS * N + 1 + orc_nb(mission); } /* :152 */
int orc_neG(int mission, int N) { return mission == ORC_S10 ? 107 * N + 37 : 105 * N + 48; }
int orc_c0(int mission, int N) { return mission == ORC_S10 ? 3 * N + 4 : N + 6; }

void orc_pattern_closed(int mission, int N, int *iG, int *jG)
{
    int e = 0, k, r, m, b;
    const int nb = orc_nb(mission);
    /* cost row */
    iG[e] = 0; jG[e++] = 0;
    if (mission == ORC_S10) {
        for (k = 0; k <= N; k++) {
            iG[e] = 0; jG[e++] = 11 * k + 1;
            iG[e] = 0; jG[e++] = 11 * k + 2;
            iG[e] = 0; jG[e++] = 11 * k + 11;
        }
    } else {
        iG[e] = 0; jG[e++] = 1;
        iG[e] = 0; jG[e++] = 2;
        for (k = 0; k < N; k++) { iG[e] = 0; jG[e++] = 11 * k + 11; }
        iG[e] = 0; jG[e++] = 11 * N + 1;
        iG[e] = 0; jG[e++] = 11 * N + 2;
        iG[e] = 0; jG[e++] = 11 * N + 11;
    }
    /* dynamics rows: dt, the 11 variables of node k, the matching state of node k+1 */
    for (k = 0; k < N; k++)
        for (r = 1; r <= 8; r++) {
            int row = 8 * k + r;
            iG[e] = row; jG[e++] = 0;
            for (m = 0; m < 11; m++) { iG[e] = row; jG[e++] = 11 * k + 1 + m; }
            iG[e] = row; jG[e++] = 11 * (k + 1) + r;
        }
    /* boundary rows */
    for (b = 0; b < nb; b++) {
        int row = 8 * N + 1 + b;
        iG[e] = row; jG[e++] = 0;
        if (mission == ORC_G7 && (b == 0 || b == 1 || b == 11)) {
            iG[e] = row; jG[e++] = 1;
            iG[e] = row; jG[e++] = 2;
            iG[e] = row; jG[e++] = 11 * N + 1;
            iG[e] = row; jG[e++] = 11 * N + 2;
        } else {
            iG[e] = row; jG[e++] = 1 + b;
            iG[e] = row; jG[e++] = 11 * N + 1 + b;
        }
    }
}

/* Does the gradient routine for (Fnum, xnum, tf, tx) raise the reference's Gnonzero flag?
 * S10: src/problemS10.cpp:346-383,400-412.  G7: src/problemG7.cpp:341-381,407-511.
 * dynamics: src/problem.cpp:1074,1197-1206. */
static int raises_nonzero(int mission, int N, int Fnum, int xnum, int tf, int tx)
{
    if (Fnum == 0) {
        if (mission == ORC_S10) return xnum == 0 || xnum == 1 || xnum == 10 || xnum == 11;
        return ((xnum == 0 || xnum == 1) && (tx == 0 || tx == N)) || xnum == 10 || xnum == 11;
    }
    if (Fnum <= ORC_NS) {
        if (tx == tf) return 1;
        return tx == tf + 1 && xnum == Fnum - 1;
    }
    if (!(tx == 0 || tx == N)) return 0;
    if (mission == ORC_S10) return xnum == Fnum - 9 && xnum <= 10;
    if (Fnum >= 11 && Fnum <= 19) return xnum == Fnum - 9;
    return xnum == 0 || xnum == 1;   /* Fnum 9, 10, 20 */
}

int orc_pattern_walk(int mission, int N, int *iG, int *jG, int *Fs, int *xs, int *tfs, int *txs)
{
    /* src/problem.cpp:813-919 */
    const int pF = ORC_NS, px = ORC_NI, nb = orc_nb(mission);
    const int n = orc_n(N), neF = orc_neF(mission, N);
    int neG = 0, reserve_dt = -1;
    for (int ii = 0; ii < neF; ii++)
        for (int jj = 0; jj < n; jj++) {
            int Fnum = ii % pF, xnum, tf, tx;
            if (Fnum == 0 && ii != 0) Fnum = pF;
            tf = (ii - 1) / pF;                       /* C division: (0-1)/8 == 0 */
            if (ii >= neF - nb) Fnum = pF + nb - (neF - 1 - ii);
            if (jj == 0) { xnum = px; reserve_dt = neG; }
            else xnum = (jj - 1) % px;
            tx = (jj - 1) / px;                       /* (0-1)/11 == 0 */
            if (raises_nonzero(mission, N, Fnum, xnum, tf, tx) || xnum == px) {
                if (iG) iG[neG] = ii;
                if (jG) jG[neG] = jj;
                if (Fs) Fs[neG] = Fnum;
                if (xs) xs[neG] = xnum;
                if (tfs) tfs[neG] = tf;
                if (txs) txs[neG] = tx;
                neG++;
            }
            /* the row's reserved dt slot is re-aimed at node tf once the diagonal is met (:883-910) */
            if (xnum == Fnum - 1 && tf == tx && Fnum > 0 && Fnum <= pF) {
                if (Fs) Fs[reserve_dt] = Fnum;
                if (xs) xs[reserve_dt] = px;
                if (tfs) tfs[reserve_dt] = tf;
                if (txs) txs[reserve_dt] = tx;
                reserve_dt = -1;
            }
        }
    return neG;
}

void orc_dispatch_closed(int mission, int N, int *Fs, int *xs, int *tfs, int *txs)
{
    const int neG = orc_neG(mission, N), neF = orc_neF(mission, N), nb = orc_nb(mission);
    int *iG = (int *)malloc(sizeof(int) * (size_t)neG), *jG = (int *)malloc(sizeof(int) * (size_t)neG);
    orc_pattern_closed(mission, N, iG, jG);
    for (int e = 0; e < neG; e++) {
        int ii = iG[e], jj = jG[e];
        int Fnum = ii % 8;
        if (Fnum == 0 && ii != 0) Fnum = 8;
        int tf = (ii - 1) / 8;
        if (ii >= neF - nb) Fnum = 8 + nb - (neF - 1 - ii);
        int xnum = jj == 0 ? 11 : (jj - 1) % 11;
        int tx = (jj - 1) / 11;
        if (jj == 0 && Fnum >= 1 && Fnum <= 8 && ii < neF - nb) tx = tf;   /* :905-908 */
        Fs[e] = Fnum; xs[e] = xnum; tfs[e] = tf; txs[e] = tx;
    }
    free(iG); free(jG);
}

/* ------------------------------------------------------------------ wind */

typedef struct {
    double W[3];      /* NED */
    double J[3][3];   /* J[i][j] = dW_i/dx_j, NED */
} ned_wind;

static void wind_at(const orc_problem *p, const double *x, int k, ned_wind *w)
{
    double enu[12] = {0};
    if (p->windmodel == ORC_WIND_SHEAR) {
        /* src/problem.cpp:521-524: zs = -z_NED; v = -Vref*zs/href; dv_dz = -Vref/href */
        double zs = -x[k * ORC_NI + 3];
        enu[1] = -p->Vref * zs / p->href;
        enu[8] = -p->Vref / p->href;
    } else if (p->windmodel == ORC_WIND_TABLE) {
        for (int f = 0; f < 12; f++) enu[f] = p->wind[(size_t)f * (p->N + 1) + k];
    } else if (p->windmodel == ORC_WIND_GRID) {
        /* src/problem.cpp:551-692.  ENU <- NED, then the cell whose lower corner is the first grid
         * coordinate within one spacing below the point (the reference's three search loops), then
         * the eight-node trilinear shape functions and their derivatives; only v is interpolated.
         * Points outside the grid use the edge cell (the reference indexes out of bounds there). */
        const double xs = x[k * ORC_NI + 2] + p->gE, ys = x[k * ORC_NI + 1] + p->gN, zs = -x[k * ORC_NI + 3] + p->gU;
        int xi, yi, zi;
        for (xi = 0; xi < p->gnx - 2; xi++) if (xs - (p->gx0 + xi * p->gdx) < p->gdx) break;
        for (yi = 0; yi < p->gny - 2; yi++) if (ys - (p->gy0 + yi * p->gdy) < p->gdy) break;
        for (zi = 0; zi < p->gnz - 2; zi++) if (zs - (p->gz0 + zi * p->gdz) < p->gdz) break;
        double vc[8];
        for (int c = 0; c < 8; c++) {
            const int i = xi + (c & 1), j = yi + ((c >> 1) & 1), l = zi + (c >> 2);
            vc[c] = p->gv[((size_t)i * p->gny + j) * p->gnz + l];
        }
        const double dx = p->gdx, dy = p->gdy, dz = p->gdz;
        const double xrel = xs - (p->gx0 + xi * dx), yrel = ys - (p->gy0 + yi * dy), zrel = zs - (p->gz0 + zi * dz);
        const double ze = xrel / dx, et = yrel / dy, mu = zrel / dz;
        const double Nf[8] = {(1 - ze) * (1 - et) * (1 - mu), ze * (1 - et) * (1 - mu), (1 - ze) * et * (1 - mu), ze * et * (1 - mu),
                              (1 - ze) * (1 - et) * mu,       ze * (1 - et) * mu,       (1 - ze) * et * mu,       ze * et * mu};
        const double NX[8] = {-(1 - et) * (1 - mu) / dx, (1 - et) * (1 - mu) / dx, -et * (1 - mu) / dx, et * (1 - mu) / dx,
                              -(1 - et) * mu / dx,       (1 - et) * mu / dx,       -et * mu / dx,       et * mu / dx};
        const double NY[8] = {-(1 - ze) * (1 - mu) / dy, -ze * (1 - mu) / dy, (1 - ze) * (1 - mu) / dy, ze * (1 - mu) / dy,
                              -(1 - ze) * mu / dy,       -ze * mu / dy,       (1 - ze) * mu / dy,       ze * mu / dy};
        const double NZ[8] = {-(1 - ze) * (1 - et) / dz, -ze * (1 - et) / dz, -(1 - ze) * et / dz, -ze * et / dz,
                              (1 - ze) * (1 - et) / dz,  ze * (1 - et) / dz,  (1 - ze) * et / dz,  ze * et / dz};
        for (int c = 0; c < 8; c++) {
            enu[1] += Nf[c] * vc[c];
            enu[6] += NX[c] * vc[c];
            enu[7] += NY[c] * vc[c];
            enu[8] += NZ[c] * vc[c];
        }
    }
    /* NED <- ENU, src/problem.cpp:970-981 (== :1061-1072) */
    const double u = enu[0], v = enu[1], ww = enu[2];
    w->W[0] = v;  w->W[1] = u;  w->W[2] = -ww;
    w->J[0][0] = enu[7];  w->J[0][1] = enu[6];  w->J[0][2] = -enu[8];    /* dWx: dv_dy, dv_dx, -dv_dz */
    w->J[1][0] = enu[4];  w->J[1][1] = enu[3];  w->J[1][2] = -enu[5];    /* dWy: du_dy, du_dx, -du_dz */
    w->J[2][0] = -enu[10]; w->J[2][1] = -enu[9]; w->J[2][2] = enu[11];   /* dWz: -dw_dy, -dw_dx, dw_dz */
}

/* ------------------------------------------------------------------ per-node dynamics */

static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* tJ[j] = sum_i e[i] J[i][j]  (JW^T e) */
static void jt_mul(const ned_wind *w, const double *e, double *out)
{
    for (int j = 0; j < 3; j++) out[j] = e[0] * w->J[0][j] + e[1] * w->J[1][j] + e[2] * w->J[2][j];
}

/* s = the 11 variables of one node (x y z Va gam chi phi CL dphi dCL T).
 * f[0..7]   : state rates (f[6] = dphi, f[7] = dCL)
 * tab[r][c] : d defect_r / d (s[0..10], dt), c = 11 is the dt column  (the reference's tabG) */
static void node_rates_and_rows(const orc_problem *p, const double *s, double dt, const ned_wind *w,
                                double f[8], double tab[8][12])
{
    const double Va = s[3], gam = s[4], chi = s[5], phi = s[6], CL = s[7], T = s[10];
    const double sg = sin(gam), cg = cos(gam), sx = sin(chi), cx = cos(chi), sp = sin(phi), cp = cos(phi);
    const double ea[3] = {cx * cg, sx * cg, -sg};
    const double eg[3] = {cx * sg, sx * sg, cg};
    const double ex[3] = {-sx, cx, 0.0};
    const double eh[3] = {cx, sx, 0.0};
    const double v[3] = {w->W[0] + Va * ea[0], w->W[1] + Va * ea[1], w->W[2] + Va * ea[2]};
    double A[3], B[3], C[3], H[3];
    jt_mul(w, ea, A); jt_mul(w, eg, B); jt_mul(w, ex, C); jt_mul(w, eh, H);

    const double m = p->mm, S = p->SS;
    const double CD = p->Cd0 + CL * CL / (p->AR * M_PI * p->ee);
    const double q = RHO * S * Va * Va / (2.0 * m);
    const double vA = dot3(v, A), vB = dot3(v, B), vC = dot3(v, C), vH = dot3(v, H);
    const double N5 = vB - GRAV * cg + q * CL * cp;      /* Va * f5 */
    const double N6 = q * CL * sp - vC;                  /* Va cg * f6 */

    f[0] = v[0]; f[1] = v[1]; f[2] = v[2];
    f[3] = T / m - vA - GRAV * sg - q * CD;
    f[4] = N5 / Va;
    f[5] = N6 / (Va * cg);
    f[6] = s[8];
    f[7] = s[9];

    if (!tab) return;
    memset(tab, 0, sizeof(double) * 8 * 12);
    /* rows 1-3 (src/problem.cpp:1084-1115) */
    tab[0][0] = -1.0; tab[0][3] = -dt * ea[0]; tab[0][4] = dt * Va * eg[0]; tab[0][5] = dt * Va * ea[1];  tab[0][11] = -v[0];
    tab[1][1] = -1.0; tab[1][3] = -dt * ea[1]; tab[1][4] = dt * Va * eg[1]; tab[1][5] = -dt * Va * ea[0]; tab[1][11] = -v[1];
    tab[2][2] = -1.0; tab[2][3] = dt * sg;     tab[2][4] = dt * Va * cg;                                  tab[2][11] = -v[2];
    /* row 4 (:1125-1130) */
    tab[3][3]  = dt * (dot3(ea, A) + RHO * S * Va * CD / m) - 1.0;
    tab[3][4]  = -dt * (vB - GRAV * cg + Va * dot3(eg, A));
    tab[3][5]  = dt * cg * (vC + Va * dot3(ex, A));
    tab[3][7]  = dt * RHO * S * Va * Va * CL / (p->AR * M_PI * p->ee * m);
    tab[3][10] = -dt / m;
    tab[3][11] = -f[3];
    /* row 5 (:1140-1145) */
    tab[4][3]  = dt * N5 / (Va * Va) - dt * (dot3(ea, B) + RHO * S * Va * CL * cp / m) / Va;
    tab[4][4]  = -dt * (vA + GRAV * sg - Va * dot3(eg, B)) / Va - 1.0;
    tab[4][5]  = -dt * (sg * vC + Va * cg * dot3(ex, B)) / Va;
    tab[4][6]  = dt * RHO * S * Va * CL * sp / (2.0 * m);
    tab[4][7]  = -dt * RHO * S * Va * cp / (2.0 * m);
    tab[4][11] = -f[4];
    /* row 6 (:1155-1160) */
    tab[5][3]  = -dt * (RHO * S * Va * CL * sp / m - dot3(ea, C)) / (Va * cg) + dt * N6 / (Va * Va * cg);
    tab[5][4]  = -dt * sg * N6 / (Va * cg * cg) - dt * dot3(eg, C) / cg;
    tab[5][5]  = -dt * (vH - Va * cg * dot3(ex, C)) / (Va * cg) - 1.0;
    tab[5][6]  = -dt * RHO * S * Va * CL * cp / (2.0 * m * cg);
    tab[5][7]  = -dt * RHO * S * Va * sp / (2.0 * m * cg);
    tab[5][11] = -f[5];
    /* rows 7-8 (:1170-1184) */
    tab[6][6] = -1.0; tab[6][8] = -dt; tab[6][11] = -s[8];
    tab[7][7] = -1.0; tab[7][9] = -dt; tab[7][11] = -s[9];
}

/* ------------------------------------------------------------------ cost and boundary pieces */

static double s10_radius(const orc_problem *p, const double *x, int k)
{
    const double xs = x[k * ORC_NI + 1], ys = x[k * ORC_NI + 2];
    return sqrt((xs - p->xg) * (xs - p->xg) + (ys - p->yg) * (ys - p->yg));
}

static double g7_dist(const orc_problem *p, const double *x)
{
    const double dxf = x[p->N * ORC_NI + 1] - x[1], dyf = x[p->N * ORC_NI + 2] - x[2];
    return sqrt(dxf * dxf + dyf * dyf);
}

static double cost_value(const orc_problem *p, const double *x)
{
    const int N = p->N;
    const double dt = x[0];
    double sumT = 0.0, sump = 0.0;
    for (int k = 0; k <= N; k++) {
        const double T = x[k * ORC_NI + 11];
        sumT = sumT + T * T;
        if (p->mission == ORC_S10) {
            const double r = s10_radius(p, x, k);
            sump = sump + (r - p->rg) * (r - p->rg);
        }
    }
    if (p->mission == ORC_S10)                       /* src/problemS10.cpp:264 */
        return 0.5 * p->kT * sumT + 0.5 * p->kp * sump + p->kdt * dt;
    return p->kT * 0.5 * sumT + p->kv * N * dt / g7_dist(p, x);   /* src/problemG7.cpp:249 */
}

/* one entry of the cost row, addressed like the reference does (xnum 0..10 at node tx, 11 = dt) */
static double cost_grad_entry(const orc_problem *p, const double *x, int xnum, int tx)
{
    const int N = p->N;
    const double dt = x[0], T = x[tx * ORC_NI + 11];
    if (p->mission == ORC_S10) {                     /* src/problemS10.cpp:336-383 */
        if (xnum == 11) return p->kdt;
        if (xnum == 10) return p->kT * T;
        const double r = s10_radius(p, x, tx);
        const double d = (xnum == 0) ? x[tx * ORC_NI + 1] - p->xg : x[tx * ORC_NI + 2] - p->yg;
        return p->kp * (r - p->rg) * d / r;
    }
    /* src/problemG7.cpp:330-381; the gradient is written with kp where the value uses kv */
    const double dist = g7_dist(p, x);
    if (xnum == 11) return p->kp * N / dist;
    if (xnum == 10) return p->kT * T;
    const double d = (xnum == 0) ? x[N * ORC_NI + 1] - x[1] : x[N * ORC_NI + 2] - x[2];
    const double g = p->kp * N * dt * d / (dist * dist * dist);
    return tx == 0 ? g : -g;
}

static void boundary_values(const orc_problem *p, const double *x, double *Fb)
{
    const int N = p->N;
    if (p->mission == ORC_S10) {                     /* src/problemS10.cpp:292-303 */
        for (int b = 0; b < 11; b++) {
            double d = x[N * ORC_NI + 1 + b] - x[1 + b];
            if (b == 5) d = d - 2.0 * M_PI;
            Fb[b] = d;
        }
        return;
    }
    /* src/problemG7.cpp:274-294 */
    const double xf = x[N * ORC_NI + 1], x0 = x[1], yf = x[N * ORC_NI + 2], y0 = x[2];
    const double dist = sqrt((xf - x0) * (xf - x0) + (yf - y0) * (yf - y0));
    const double dmax = sqrt((p->xg - x0) * (p->xg - x0) + (p->yg - y0) * (p->yg - y0));
    Fb[0] = xf - x0 - dist * cos(p->chi_d);
    Fb[1] = yf - y0 - dist * sin(p->chi_d);
    for (int b = 2; b < 11; b++) Fb[b] = x[N * ORC_NI + 1 + b] - x[1 + b];
    Fb[11] = dist - dmax;
}

/* one entry of a boundary row (Fnum 9..), addressed like the reference does */
static double boundary_grad_entry(const orc_problem *p, const double *x, int Fnum, int xnum, int tx)
{
    const int N = p->N;
    /* dt column: undefined in the reference for S10 (src/problemS10.cpp:397, never assigned),
     * 0 for G7 (src/problemG7.cpp:404).  This build defines both as 0.0. */
    if (xnum == 11) return 0.0;
    const double sgn = (tx == 0) ? -1.0 : 1.0;
    if (p->mission == ORC_S10 || (Fnum >= 11 && Fnum <= 19)) return sgn;
    const double dxf = x[N * ORC_NI + 1] - x[1], dyf = x[N * ORC_NI + 2] - x[2];
    const double dist = sqrt(dxf * dxf + dyf * dyf);
    const double d = (xnum == 0) ? dxf : dyf;
    if (Fnum == 20) return sgn * d / dist;           /* src/problemG7.cpp:484-511 */
    const double trig = (Fnum == 9) ? cos(p->chi_d) : sin(p->chi_d);
    const int diag = (Fnum == 9 && xnum == 0) || (Fnum == 10 && xnum == 1);
    /* src/problemG7.cpp:423-481: node 0 gets (-1 if diagonal) + (d/dist) trig, node N the negative */
    const double g0 = (diag ? -1.0 : 0.0) + (d / dist) * trig;
    return tx == 0 ? g0 : -g0;
}

/* ------------------------------------------------------------------ fused evaluation */

void orc_eval(const orc_problem *p, const double *x, int needF, double *F, int needG, double *G)
{
    const int N = p->N, nb = orc_nb(p->mission);
    const double dt = x[0];
    if (needF) F[0] = cost_value(p, x);
    int e = 0;
    if (needG) {
        /* cost row, in pattern order */
        G[e++] = cost_grad_entry(p, x, 11, 0);
        if (p->mission == ORC_S10) {
            for (int k = 0; k <= N; k++) {
                G[e++] = cost_grad_entry(p, x, 0, k);
                G[e++] = cost_grad_entry(p, x, 1, k);
                G[e++] = cost_grad_entry(p, x, 10, k);
            }
        } else {
            G[e++] = cost_grad_entry(p, x, 0, 0);
            G[e++] = cost_grad_entry(p, x, 1, 0);
            for (int k = 0; k < N; k++) G[e++] = cost_grad_entry(p, x, 10, k);
            G[e++] = cost_grad_entry(p, x, 0, N);
            G[e++] = cost_grad_entry(p, x, 1, N);
            G[e++] = cost_grad_entry(p, x, 10, N);
        }
    }
    for (int k = 0; k < N; k++) {
        const double *s = x + k * ORC_NI + 1, *sn = s + ORC_NI;
        ned_wind w;
        double f[8], tab[8][12];
        wind_at(p, x, k, &w);
        node_rates_and_rows(p, s, dt, &w, f, needG ? tab : NULL);
        if (needF)
            for (int r = 0; r < 8; r++) F[8 * k + 1 + r] = sn[r] - f[r] * dt - s[r];
        if (needG)
            for (int r = 0; r < 8; r++) {
                G[e++] = tab[r][11];
                for (int m = 0; m < 11; m++) G[e++] = tab[r][m];
                G[e++] = 1.0;                        /* src/problem.cpp:1200-1205 */
            }
    }
    if (needF) boundary_values(p, x, F + 8 * N + 1);
    if (needG)
        for (int b = 0; b < nb; b++) {
            const int Fnum = 9 + b;
            G[e++] = boundary_grad_entry(p, x, Fnum, 11, 0);
            if (p->mission == ORC_G7 && (b == 0 || b == 1 || b == 11)) {
                G[e++] = boundary_grad_entry(p, x, Fnum, 0, 0);
                G[e++] = boundary_grad_entry(p, x, Fnum, 1, 0);
                G[e++] = boundary_grad_entry(p, x, Fnum, 0, N);
                G[e++] = boundary_grad_entry(p, x, Fnum, 1, N);
            } else {
                G[e++] = boundary_grad_entry(p, x, Fnum, b, 0);
                G[e++] = boundary_grad_entry(p, x, Fnum, b, N);
            }
        }
}

/* ------------------------------------------------------------------ reference evaluation order */

void orc_eval_entrywise(const orc_problem *p, const double *x, int needF, double *F,
                        int needG, double *G, int neG,
                        const int *Fs, const int *xs, const int *tfs, const int *txs)
{
    const int N = p->N;
    const double dt = x[0];
    if (needF) {                                      /* src/problem.cpp:765-774 */
        F[0] = cost_value(p, x);
        for (int k = 0; k < N; k++) {
            const double *s = x + k * ORC_NI + 1, *sn = s + ORC_NI;
            ned_wind w;
            double f[8];
            wind_at(p, x, k, &w);
            node_rates_and_rows(p, s, dt, &w, f, NULL);
            for (int r = 0; r < 8; r++) F[8 * k + 1 + r] = sn[r] - f[r] * dt - s[r];
        }
        boundary_values(p, x, F + 8 * N + 1);
    }
    if (!needG) return;
    for (int e = 0; e < neG; e++) {                   /* src/problem.cpp:785-802 */
        const int Fnum = Fs[e], xnum = xs[e], tf = tfs[e], tx = txs[e];
        if (Fnum == 0) {
            G[e] = cost_grad_entry(p, x, xnum, tx);
        } else if (Fnum <= ORC_NS) {
            /* src/problem.cpp:1035-1208: the whole 12-wide row at node tx is rebuilt, one entry kept */
            if (tx == tf) {
                ned_wind w;
                double f[8], tab[8][12];
                wind_at(p, x, tx, &w);
                node_rates_and_rows(p, x + tx * ORC_NI + 1, dt, &w, f, tab);
                G[e] = tab[Fnum - 1][xnum];
            } else {
                G[e] = 1.0;
            }
        } else {
            G[e] = boundary_grad_entry(p, x, Fnum, xnum, tx);
        }
    }
}

int orc_eval_batch(const orc_problem *probs, int B, const double *X, int ldx,
                   double *F, int ldf, double *G, int ldg, int nthreads)
{
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 1) {
        used = nthreads;
#pragma omp parallel for num_threads(nthreads) schedule(static)
        for (int b = 0; b < B; b++)
            orc_eval(&probs[b], X + (size_t)b * ldx, 1, F + (size_t)b * ldf, 1, G + (size_t)b * ldg);
        return used;
    }
#endif
    (void)nthreads;
    for (int b = 0; b < B; b++)
        orc_eval(&probs[b], X + (size_t)b * ldx, 1, F + (size_t)b * ldf, 1, G + (size_t)b * ldg);
    return used;
}

/* ------------------------------------------------------------------ initial guess and bounds */

void orc_x0(const orc_problem *p, double xi, double yi, double zi, double *x)
{
    /* src/problemS10.cpp:38-211 (circle, tfinal 20 s, 100 m amplitudes) and
     * src/problemG7.cpp:39-203 (40 m straight line over 10 s, rotated onto chi_d). */
    const int N = p->N, s10 = p->mission == ORC_S10;
    const double tfinal = s10 ? 20.0 : 10.0;
    const double dt = tfinal / N;
    const double xAmp = s10 ? 100.0 : 40.0, yAmp = s10 ? 100.0 : 0.0, zAmp = 0.0;
    const double ws = 2.0 * M_PI / tfinal;
    double t = 0.0, pre_chi = 0.0, pre_phi = 0.0, pre_CL = 0.0;
    for (int k = 0; k <= N; k++) {
        double pos[3], vel[3], acc[3];
        const double swt = sin(ws * t), cwt = cos(ws * t);
        if (s10) {
            pos[0] = xAmp * swt - xAmp + xi;  pos[1] = -yAmp * cwt + yi;        pos[2] = zAmp * cwt - zAmp + zi;
            vel[0] = ws * xAmp * cwt;         vel[1] = ws * yAmp * swt;         vel[2] = -ws * zAmp * swt;
            acc[0] = -ws * ws * xAmp * swt;   acc[1] = ws * ws * yAmp * cwt;    acc[2] = -ws * ws * zAmp * cwt;
        } else {
            pos[0] = xAmp / tfinal * t + xi;  pos[1] = -yAmp * cwt + yAmp + yi; pos[2] = zAmp * cwt - zAmp + zi;
            vel[0] = xAmp / tfinal;           vel[1] = yAmp * ws * swt;         vel[2] = -zAmp * ws * swt;
            acc[0] = 0.0;                     acc[1] = yAmp * ws * ws * cwt;    acc[2] = -zAmp * ws * ws * cwt;
            /* RotateYaw, src/problemG7.cpp:520-542: positions only */
            const double c = cos(p->chi_d), s = sin(p->chi_d);
            const double px = c * pos[0] + -s * pos[1] + 0.0 * pos[2];
            const double py = s * pos[0] + c * pos[1] + 0.0 * pos[2];
            pos[0] = px; pos[1] = py;
        }
        /* air-relative velocity with zero wind on the guess */
        const double va = sqrt(vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2]);
        double chi = atan2(vel[1], vel[0]) + (s10 ? 0.0 : p->chi_d);
        const double gam = atan2(-vel[2], sqrt(vel[0] * vel[0] + vel[1] * vel[1]));
        if (k > 0) {          /* unwrap against the previous node */
            double diff = chi - pre_chi;
            while (diff < -M_PI || diff > M_PI) {
                if (diff < -M_PI) chi = chi + 2.0 * M_PI * ceil((-M_PI - diff) / (2.0 * M_PI));
                if (diff > M_PI) chi = chi + 2.0 * M_PI * floor((M_PI - diff) / (2.0 * M_PI));
                diff = chi - pre_chi;
            }
        }
        const double r1[3] = {vel[0] / va, vel[1] / va, vel[2] / va};
        const double ag[3] = {acc[0], acc[1], acc[2] - GRAV};
        /* component of (a - g) normal to the flight path */
        double an[3];
        an[0] = -ag[0] * (r1[0] * r1[0] - 1.0) - r1[0] * r1[1] * ag[1] - r1[0] * r1[2] * ag[2];
        an[1] = -ag[1] * (r1[1] * r1[1] - 1.0) - r1[0] * r1[1] * ag[0] - r1[1] * r1[2] * ag[2];
        an[2] = -ag[2] * (r1[2] * r1[2] - 1.0) - r1[0] * r1[2] * ag[0] - r1[1] * r1[2] * ag[1];
        const double man = sqrt(an[0] * an[0] + an[1] * an[1] + an[2] * an[2]);
        const double r3[3] = {-an[0] / man, -an[1] / man, -an[2] / man};
        const double r2z = r3[0] * r1[1] - r3[1] * r1[0];
        const double phi = atan2(r2z, r3[2]);
        const double L = p->mm * man;
        const double CL = 2.0 * L / (RHO * va * va * p->SS);
        const double D = 0.5 * RHO * va * va * p->SS * (p->Cd0 + CL * CL / (M_PI * p->AR * p->ee));
        const double T = p->mm * (r1[0] * ag[0] + r1[1] * ag[1] + r1[2] * ag[2]) + D;
        double *s = x + k * ORC_NI + 1;
        s[0] = pos[0]; s[1] = pos[1]; s[2] = pos[2];
        s[3] = va; s[4] = gam; s[5] = chi; s[6] = phi; s[7] = CL;
        s[8] = k == 0 ? 0.0 : (phi - pre_phi) / dt;
        s[9] = k == 0 ? 0.0 : (CL - pre_CL) / dt;
        s[10] = T;
        pre_phi = phi; pre_CL = CL; pre_chi = chi;
        t = t + dt;
    }
    x[0] = dt;
    if (s10) {   /* src/problemS10.cpp:210-211 */
        x[9] = x[N * ORC_NI + 9];
        x[10] = x[N * ORC_NI + 10];
    }
}

void orc_bounds(int mission, int N, const double *ac, const double *lim,
                double xi, double yi, double zi,
                double *xlow, double *xupp, double *Flow, double *Fupp)
{
    const double d2r = M_PI / 180.0;
    const double CLmin = ac[6], CLmax = ac[7], phimax = ac[8] * d2r, Vamin = ac[9], Vamax = ac[10];
    const double gammamax = ac[11] * d2r, phidotmax = ac[12] * d2r, Tmin = ac[13], Tmax = ac[14];
    const double dtmin = lim[0], dtmax = lim[1];
    const double xmin = lim[2], xmax = lim[3], ymin = lim[4], ymax = lim[5], zmin = lim[6], zmax = lim[7];
    const int nb = orc_nb(mission), neF = orc_neF(mission, N);
    /* node 0: src/problem.cpp:80-134 (constants), :254-268 */
    xlow[0] = dtmin; xupp[0] = dtmax;
    double *lo = xlow + 1, *up = xupp + 1;
    lo[0] = xi; up[0] = xi; lo[1] = yi; up[1] = yi; lo[2] = zi; up[2] = zi;
    lo[3] = 4.0; up[3] = 50.0;
    lo[4] = 0.0; up[4] = 0.0;
    if (mission == ORC_S10) { lo[5] = -1.7453292519943296e+18; up[5] = 1.7453292519943296e+18; }
    else                    { lo[5] = -1e20 * M_PI / 180.0;    up[5] = 1e20 * M_PI / 180.0; }
    if (mission == ORC_S10) { lo[6] = -1.5707963267948966;     up[6] = 1.5707963267948966; }
    else                    { lo[6] = -90.0 * M_PI / 180.0;    up[6] = 90.0 * M_PI / 180.0; }
    lo[7] = -0.5; up[7] = 3.0;
    lo[8] = -3.4906585039886591; up[8] = 3.4906585039886591;
    lo[9] = -200.0; up[9] = 200.0;
    lo[10] = 0.0; up[10] = 1e20;
    /* nodes 1..N: src/problem.cpp:272-285 (CLdot is bounded by phidotmax there) */
    for (int k = 1; k <= N; k++) {
        lo = xlow + 1 + k * ORC_NI; up = xupp + 1 + k * ORC_NI;
        lo[0] = xmin; up[0] = xmax; lo[1] = ymin; up[1] = ymax; lo[2] = zmin; up[2] = zmax;
        lo[3] = Vamin; up[3] = Vamax;
        lo[4] = -gammamax; up[4] = gammamax;
        lo[5] = -1e20; up[5] = 1e20;
        lo[6] = -phimax; up[6] = phimax;
        lo[7] = CLmin; up[7] = CLmax;
        lo[8] = -phidotmax; up[8] = phidotmax;
        lo[9] = -phidotmax; up[9] = phidotmax;
        lo[10] = Tmin; up[10] = Tmax;
    }
    /* rows: objective free, defects and boundary rows equalities, G7's last row dist <= dmax
     * (src/problem.cpp:297-358) */
    Flow[0] = -1e20; Fupp[0] = 1e20;
    for (int i = 1; i < neF; i++) { Flow[i] = 0.0; Fupp[i] = 0.0; }
    if (mission == ORC_G7) Flow[neF - 1] = -1e20;
    (void)nb;
}
