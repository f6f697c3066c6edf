// setup.cpp -- sizes, closed-form sparsity pattern, initial guess, bounds (host, one-time).
#include "setup.h"

#include <cmath>

#include "kernels.h"
#include "slab_table.h"

namespace tolfg {

namespace {
constexpr double kGrav = 9.81;    // ref: include/problem.h:72
constexpr double kRho = 1.2682;   // ref: include/problem.h:73
}

Sizes make_sizes(int mission, int N, int pattern)
{
    Sizes s;
    s.mission = mission;
    s.pattern = pattern;
    s.N = N;
    s.nb = mission == MISSION_S10 ? 11 : 12;
    s.n = 11 * (N + 1) + 1;
    s.neF = 8 * N + 1 + s.nb;
    s.c0 = mission == MISSION_S10 ? 3 * N + 4 : N + 6;          // objective row
    if (pattern == PATTERN_COMPACT) {
        // 46 entries per node; boundary rows without their (always zero) dt entries
        s.slab = SLAB_COMPACT;
        s.neG = s.c0 + SLAB_COMPACT * N + (mission == MISSION_S10 ? 22 : 30);
    } else {
        s.slab = SLAB_FULL;
        s.neG = mission == MISSION_S10 ? 107 * N + 37 : 105 * N + 48;
    }
    return s;
}

void make_pattern(const Sizes &sz, int *iG, int *jG)
{
    const int N = sz.N;
    int e = 0;
    auto put = [&](int row, int col) { iG[e] = row; jG[e] = col; ++e; };
    // objective row: dt, then the nodes' (x, y, T) for S10 / (x0, y0), every T, (xN, yN) for G7
    put(0, 0);
    if (sz.mission == MISSION_S10) {
        for (int k = 0; k <= N; ++k) { put(0, 11 * k + 1); put(0, 11 * k + 2); put(0, 11 * k + 11); }
    } else {
        put(0, 1); put(0, 2);
        for (int k = 0; k < N; ++k) put(0, 11 * k + 11);
        put(0, 11 * N + 1); put(0, 11 * N + 2); put(0, 11 * N + 11);
    }
    // defect rows: 13 entries each -> one contiguous 104-entry slab per node; the compact pattern
    // keeps the entries the slab table marks as structurally non-zero (46 per node)
    constexpr SlabTableFull table = make_slab_table();
    const bool compact = sz.pattern == PATTERN_COMPACT;
    for (int k = 0; k < N; ++k)
        for (int r = 1; r <= 8; ++r)
            for (int c = 0; c < 13; ++c) {
                if (compact && table.c[13 * (r - 1) + c] == SL_ZERO) continue;
                const int col = c == 0 ? 0 : (c == 12 ? 11 * (k + 1) + r : 11 * k + c);
                put(8 * k + r, col);
            }
    // boundary rows
    for (int b = 0; b < sz.nb; ++b) {
        const int row = 8 * N + 1 + b;
        if (!compact) put(row, 0);
        if (sz.mission == MISSION_G7 && (b == 0 || b == 1 || b == 11)) {
            put(row, 1); put(row, 2); put(row, 11 * N + 1); put(row, 11 * N + 2);
        } else {
            put(row, 1 + b); put(row, 11 * N + 1 + b);
        }
    }
}

void initial_guess(const Sizes &sz, const aircraft &ac, const Start &st, double chi_d, double *x)
{
    const bool loiter = sz.mission == MISSION_S10;
    const int N = sz.N;
    // S10: one 20 s lap of a 100 m circle; G7: 40 m along the course line in 10 s
    const double tfinal = loiter ? 20.0 : 10.0;
    const double ax = loiter ? 100.0 : 40.0, ay = loiter ? 100.0 : 0.0, az = 0.0;
    const double dt = tfinal / N;
    const double w = 2.0 * M_PI / tfinal;
    const double cd = std::cos(chi_d), sd = std::sin(chi_d);
    double t = 0.0, chi_prev = 0.0, phi_prev = 0.0, CL_prev = 0.0;
    for (int k = 0; k <= N; ++k, t = t + dt) {
        const double s = std::sin(w * t), c = std::cos(w * t);
        double p[3], v[3], acc[3];
        if (loiter) {
            p[0] = ax * s - ax + st.xi;     p[1] = -ay * c + st.yi;        p[2] = az * c - az + st.zi;
            v[0] = w * ax * c;              v[1] = w * ay * s;             v[2] = -w * az * s;
            acc[0] = -w * w * ax * s;       acc[1] = w * w * ay * c;       acc[2] = -w * w * az * c;
        } else {
            const double px = ax / tfinal * t + st.xi, py = -ay * c + ay + st.yi;
            p[0] = cd * px - sd * py;       // yaw the position onto the course (velocity: chi += chi_d)
            p[1] = sd * px + cd * py;
            p[2] = az * c - az + st.zi;
            v[0] = ax / tfinal;             v[1] = ay * w * s;             v[2] = -az * w * s;
            acc[0] = 0.0;                   acc[1] = ay * w * w * c;       acc[2] = -az * w * w * c;
        }
        const double Va = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        double chi = std::atan2(v[1], v[0]) + (loiter ? 0.0 : chi_d);
        const double gam = std::atan2(-v[2], std::sqrt(v[0] * v[0] + v[1] * v[1]));
        if (k > 0) {   // keep the course continuous from node to node
            double jump = chi - chi_prev;
            while (jump < -M_PI || jump > M_PI) {
                if (jump < -M_PI) chi = chi + 2.0 * M_PI * std::ceil((-M_PI - jump) / (2.0 * M_PI));
                if (jump > M_PI) chi = chi + 2.0 * M_PI * std::floor((M_PI - jump) / (2.0 * M_PI));
                jump = chi - chi_prev;
            }
        }
        // unit tangent, specific force a - g, and its part normal to the path
        const double u[3] = {v[0] / Va, v[1] / Va, v[2] / Va};
        const double sf[3] = {acc[0], acc[1], acc[2] - kGrav};
        const double nrm[3] = {
            -sf[0] * (u[0] * u[0] - 1.0) - u[0] * u[1] * sf[1] - u[0] * u[2] * sf[2],
            -sf[1] * (u[1] * u[1] - 1.0) - u[0] * u[1] * sf[0] - u[1] * u[2] * sf[2],
            -sf[2] * (u[2] * u[2] - 1.0) - u[0] * u[2] * sf[0] - u[1] * u[2] * sf[1]};
        const double nmag = std::sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
        const double lift[3] = {-nrm[0] / nmag, -nrm[1] / nmag, -nrm[2] / nmag};
        const double phi = std::atan2(lift[0] * u[1] - lift[1] * u[0], lift[2]);
        const double CL = 2.0 * (ac.mm * nmag) / (kRho * Va * Va * ac.SS);
        const double drag = 0.5 * kRho * Va * Va * ac.SS * (ac.Cd0 + CL * CL / (M_PI * ac.AR * ac.ee));
        const double thrust = ac.mm * (u[0] * sf[0] + u[1] * sf[1] + u[2] * sf[2]) + drag;
        double *nd = x + 11 * k + 1;
        nd[0] = p[0]; nd[1] = p[1]; nd[2] = p[2];
        nd[3] = Va; nd[4] = gam; nd[5] = chi; nd[6] = phi; nd[7] = CL;
        nd[8] = k ? (phi - phi_prev) / dt : 0.0;
        nd[9] = k ? (CL - CL_prev) / dt : 0.0;
        nd[10] = thrust;
        chi_prev = chi; phi_prev = phi; CL_prev = CL;
    }
    x[0] = dt;
    if (loiter) {   // S10 copies the last node's control rates into node 0 (src/problemS10.cpp:210-211)
        x[9] = x[11 * N + 9];
        x[10] = x[11 * N + 10];
    }
}

void set_limits(const Sizes &sz, const aircraft &ac, const limit &lm, const Start &st,
                double *xlow, double *xupp, double *Flow, double *Fupp)
{
    const bool loiter = sz.mission == MISSION_S10;
    auto node = [&](int k, int m, double lo, double up) { xlow[11 * k + 1 + m] = lo; xupp[11 * k + 1 + m] = up; };
    xlow[0] = lm.dtmin; xupp[0] = lm.dtmax;
    // node 0 is pinned to the start position with the wide constants of src/problem.cpp:80-134
    node(0, 0, st.xi, st.xi); node(0, 1, st.yi, st.yi); node(0, 2, st.zi, st.zi);
    node(0, 3, 4.0, 50.0);
    node(0, 4, 0.0, 0.0);
    if (loiter) {
        node(0, 5, -1.7453292519943296e+18, 1.7453292519943296e+18);
        node(0, 6, -1.5707963267948966, 1.5707963267948966);
    } else {
        node(0, 5, -1e20 * M_PI / 180.0, 1e20 * M_PI / 180.0);
        node(0, 6, -90.0 * M_PI / 180.0, 90.0 * M_PI / 180.0);
    }
    node(0, 7, -0.5, 3.0);
    node(0, 8, -3.4906585039886591, 3.4906585039886591);
    node(0, 9, -200.0, 200.0);
    node(0, 10, 0.0, 1e20);
    for (int k = 1; k <= sz.N; ++k) {   // src/problem.cpp:272-285
        node(k, 0, lm.xmin, lm.xmax); node(k, 1, lm.ymin, lm.ymax); node(k, 2, lm.zmin, lm.zmax);
        node(k, 3, ac.Vamin, ac.Vamax);
        node(k, 4, -ac.gammamax, ac.gammamax);
        node(k, 5, -1e20, 1e20);
        node(k, 6, -ac.phimax, ac.phimax);
        node(k, 7, ac.CLmin, ac.CLmax);
        node(k, 8, -ac.phidotmax, ac.phidotmax);
        node(k, 9, -ac.phidotmax, ac.phidotmax);   // the reference bounds CLdot by phidotmax too
        node(k, 10, ac.Tmin, ac.Tmax);
    }
    Flow[0] = -1e20; Fupp[0] = 1e20;
    for (int i = 1; i < sz.neF; ++i) { Flow[i] = 0.0; Fupp[i] = 0.0; }
    if (!loiter) Flow[sz.neF - 1] = -1e20;   // G7: dist <= dmax (src/problem.cpp:345-349)
}

}  // namespace tolfg
