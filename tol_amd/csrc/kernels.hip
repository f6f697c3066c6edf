// kernels.hip -- hand-written gfx950 (CDNA4) kernels for tol's SNOPT user function.
//
// One launch evaluates F and G of a batch of trajectories.  Work decomposition:
//
//   * one 64-lane wavefront = one workgroup = either
//       - a DYNAMICS TILE: 64 consecutive collocation nodes of one trajectory, lane = node:
//         defects F[8k+1..8k+8] and the contiguous 104-element Jacobian slab of each node
//         (ref: problem::dynamicConstraints src/problem.cpp:929-1021, problem::dynamicsGradients
//         src/problem.cpp:1035-1208, wind models 0/1 src/problem.cpp:480-531), or
//       - the trajectory's EPILOGUE: objective F[0] and its gradient row, boundary rows and their
//         gradients (ref: problemS10::cost/costGradient/boundaryConstraints/boundaryGradients
//         src/problemS10.cpp:227-415, problemG7::... src/problemG7.cpp:225-513).
//   * the SNOPT-facing layouts are node-major (x[11k+1+m], G slab c0+104k), so a lane-per-node
//     access is 88 B / 832 B strided.  Every global access is therefore made wave-cooperative
//     through LDS: the tile's x window is loaded with contiguous 16-byte loads, and the slabs are
//     written with contiguous 16-byte stores (1 KiB per wave instruction).  Of the 104 slab
//     elements only 32 are computed per node; the 58 structural zeros and the +-1 constants are
//     injected from a compile-time table while streaming out, so the LDS exchange is 43 elements
//     per node (22 KB per wave in fp64 -> 7 waves per CU).
//   * air-frame coefficients and per-trajectory constants are block-uniform: they arrive through
//     the kernarg segment / scalar loads and live in SGPRs (cheaper than an LDS copy).
//   * the objective is a fixed-order butterfly over the wave (deterministic, no atomics).
//   * no MFMA: ~1 flop/byte, HBM-bound.
//
// The maths is the vector form stated in oracle/tolfg_oracle.c; W and grad W are held constant in
// the Jacobian exactly as the reference's tabulated entries do.
#include "kernels.h"

#include <hip/hip_runtime.h>

namespace tolfg {

namespace {

constexpr int TILE = 64;           // nodes per dynamics tile = wavefront width
constexpr int NI = 11;             // variables per node   (problems/*/snopt.param:3)
constexpr int SLAB = 104;          // 8 rows x 13 pattern entries per node
constexpr double kGrav = 9.81;     // include/problem.h:72
constexpr double kTwoPi = 6.283185307179586476925286766559;

// LDS row of one node (elements): 32 computed Jacobian values, 3 constants, 8 defects.
constexpr int SL_ZERO = 32, SL_ONE = 33, SL_MONE = 34, SL_F = 35;
constexpr int RS = 43;             // odd stride: conflict-free ds_write_b64 across lanes

// ---- which LDS slot feeds slab element e = 13*(row-1) + col,  col = [dt | x y z Va gam chi phi CL dphi dCL T | next]
struct SlabTable { unsigned char c[SLAB]; };
constexpr SlabTable make_slab_table()
{
    SlabTable t{};
    for (int i = 0; i < SLAB; i++) t.c[i] = SL_ZERO;
    for (int r = 0; r < 8; r++) t.c[13 * r + 12] = SL_ONE;          // d defect_r / d s_{k+1,r}
    // row 1 (x)                          row 2 (y)                           row 3 (z)
    t.c[0] = 0;  t.c[1] = SL_MONE;        t.c[13] = 4; t.c[15] = SL_MONE;     t.c[26] = 8; t.c[29] = SL_MONE;
    t.c[4] = 1;  t.c[5] = 2; t.c[6] = 3;  t.c[17] = 5; t.c[18] = 6; t.c[19] = 7;  t.c[30] = 9; t.c[31] = 10;
    // row 4 (Va): dt Va gam chi CL T
    t.c[39] = 11; t.c[43] = 12; t.c[44] = 13; t.c[45] = 14; t.c[47] = 15; t.c[50] = 16;
    // row 5 (gam): dt Va gam chi phi CL
    t.c[52] = 17; t.c[56] = 18; t.c[57] = 19; t.c[58] = 20; t.c[59] = 21; t.c[60] = 22;
    // row 6 (chi): dt Va gam chi phi CL
    t.c[65] = 23; t.c[69] = 24; t.c[70] = 25; t.c[71] = 26; t.c[72] = 27; t.c[73] = 28;
    // row 7 (phi): dt, phi = -1, dphi = -dt        row 8 (CL): dt, CL = -1, dCL = -dt
    t.c[78] = 29; t.c[85] = SL_MONE; t.c[87] = 31;  t.c[91] = 30; t.c[99] = SL_MONE; t.c[101] = 31;
    return t;
}
__device__ constexpr SlabTable kSlab = make_slab_table();

template <typename T, int VEC> struct Vec { typedef T type __attribute__((ext_vector_type(VEC))); };
template <typename T> struct Vec<T, 1> { typedef T type; };

__device__ __forceinline__ void sincos_t(double a, double &s, double &c) { sincos(a, &s, &c); }
__device__ __forceinline__ void sincos_t(float a, float &s, float &c) { sincosf(a, &s, &c); }
__device__ __forceinline__ double sqrt_t(double a) { return sqrt(a); }
__device__ __forceinline__ float sqrt_t(float a) { return sqrtf(a); }

// Copy len contiguous elements, starting at a 16-byte aligned address when VEC > 1, into LDS.
template <typename T, int VEC>
__device__ __forceinline__ void stage_window(T *lds, const T *src, int len, int lane)
{
    typedef typename Vec<T, VEC>::type vec;
    const int nvec = len / VEC;
    for (int i = lane; i < nvec; i += TILE)
        *reinterpret_cast<vec *>(lds + i * VEC) = *reinterpret_cast<const vec *>(src + i * VEC);
    if (VEC > 1) {
        const int i = nvec * VEC + lane;
        if (i < len) lds[i] = src[i];
    }
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int m = 1; m < TILE; m <<= 1) v += __shfl_xor(v, m, TILE);
    return v;
}

// e . V where V = JW^T(some unit vector) has the sparsity the wind model implies
template <int WIND, typename T>
__device__ __forceinline__ T dotw(T e0, T e1, T e2, const T (&V)[3])
{
    if constexpr (WIND == WIND_NONE) return T(0);
    else if constexpr (WIND == WIND_SHEAR) return e2 * V[2];
    else return e0 * V[0] + e1 * V[1] + e2 * V[2];
}

// State rates f[0..7] and the 32 computed Jacobian values g[] of one node.
// s = x y z Va gam chi phi CL dphi dCL T;  we = the node's 12 ENU wind values (WIND_TABLE only).
template <typename T, int WIND>
__device__ __forceinline__ void node_eval(const T (&s)[NI], T dt, T shear, const T (&we)[12],
                                          T inv_m, T qk, T Cd0, T kind, T (&f)[8], T (&g)[32])
{
    const T Va = s[3], CL = s[7], Th = s[10];
    T sg, cg, sx, cx, sp, cp;
    sincos_t(s[4], sg, cg);
    sincos_t(s[5], sx, cx);
    sincos_t(s[6], sp, cp);
    const T ea0 = cx * cg, ea1 = sx * cg, ea2 = -sg;     // along the air-relative velocity
    const T eg0 = cx * sg, eg1 = sx * sg, eg2 = cg;      // d e_a / d gam = -e_g
    // e_x = (-sx, cx, 0) = (1/cg) d e_a / d chi,  e_h = (cx, sx, 0) = -d e_x / d chi

    T W[3] = {T(0), T(0), T(0)};
    T A[3] = {T(0), T(0), T(0)}, B[3] = {T(0), T(0), T(0)}, C[3] = {T(0), T(0), T(0)}, H[3] = {T(0), T(0), T(0)};
    if constexpr (WIND == WIND_SHEAR) {
        // src/problem.cpp:521-524 through the ENU->NED map: Wx = shear * z_NED, dWx/dz = shear
        W[0] = shear * s[2];
        A[2] = ea0 * shear; B[2] = eg0 * shear; C[2] = -sx * shear; H[2] = cx * shear;
    } else if constexpr (WIND == WIND_TABLE) {
        // NED <- ENU, src/problem.cpp:970-981
        W[0] = we[1]; W[1] = we[0]; W[2] = -we[2];
        const T J00 = we[7], J01 = we[6], J02 = -we[8];
        const T J10 = we[4], J11 = we[3], J12 = -we[5];
        const T J20 = -we[10], J21 = -we[9], J22 = we[11];
        A[0] = ea0 * J00 + ea1 * J10 + ea2 * J20; A[1] = ea0 * J01 + ea1 * J11 + ea2 * J21; A[2] = ea0 * J02 + ea1 * J12 + ea2 * J22;
        B[0] = eg0 * J00 + eg1 * J10 + eg2 * J20; B[1] = eg0 * J01 + eg1 * J11 + eg2 * J21; B[2] = eg0 * J02 + eg1 * J12 + eg2 * J22;
        C[0] = -sx * J00 + cx * J10; C[1] = -sx * J01 + cx * J11; C[2] = -sx * J02 + cx * J12;
        H[0] = cx * J00 + sx * J10;  H[1] = cx * J01 + sx * J11;  H[2] = cx * J02 + sx * J12;
    }
    const T v0 = W[0] + Va * ea0, v1 = W[1] + Va * ea1, v2 = W[2] + Va * ea2;
    const T vA = dotw<WIND>(v0, v1, v2, A), vB = dotw<WIND>(v0, v1, v2, B);
    const T vC = dotw<WIND>(v0, v1, v2, C), vH = dotw<WIND>(v0, v1, v2, H);

    const T g9 = T(kGrav);
    const T iVa = T(1) / Va, icg = T(1) / cg;
    const T qV = qk * Va;            // rho S Va / (2 m)
    const T q = qV * Va;             // rho S Va^2 / (2 m)
    const T CD = Cd0 + CL * CL * kind;
    const T N5 = vB - g9 * cg + q * CL * cp;        // Va * gamdot
    const T N6 = q * CL * sp - vC;                  // Va cg * chidot

    f[0] = v0; f[1] = v1; f[2] = v2;
    f[3] = Th * inv_m - vA - g9 * sg - q * CD;
    f[4] = N5 * iVa;
    f[5] = N6 * iVa * icg;
    f[6] = s[8];
    f[7] = s[9];

    const T dtVa = dt * Va;
    // rows 1-3  (src/problem.cpp:1084-1115)
    g[0] = -v0; g[1] = -dt * ea0; g[2] = dtVa * eg0; g[3] = dtVa * ea1;
    g[4] = -v1; g[5] = -dt * ea1; g[6] = dtVa * eg1; g[7] = -dtVa * ea0;
    g[8] = -v2; g[9] = dt * sg;   g[10] = dtVa * cg;
    // row 4  (:1125-1130)
    g[11] = -f[3];
    g[12] = dt * (dotw<WIND>(ea0, ea1, ea2, A) + T(2) * qV * CD) - T(1);
    g[13] = -dt * (vB - g9 * cg + Va * dotw<WIND>(eg0, eg1, eg2, A));
    g[14] = dt * cg * (vC + Va * dotw<WIND>(-sx, cx, T(0), A));
    g[15] = dt * T(2) * q * CL * kind;
    g[16] = -dt * inv_m;
    // row 5  (:1140-1145)
    g[17] = -f[4];
    g[18] = dt * N5 * iVa * iVa - dt * (dotw<WIND>(ea0, ea1, ea2, B) + T(2) * qV * CL * cp) * iVa;
    g[19] = -dt * (vA + g9 * sg - Va * dotw<WIND>(eg0, eg1, eg2, B)) * iVa - T(1);
    g[20] = -dt * (sg * vC + Va * cg * dotw<WIND>(-sx, cx, T(0), B)) * iVa;
    g[21] = dt * qV * CL * sp;
    g[22] = -dt * qV * cp;
    // row 6  (:1155-1160)
    g[23] = -f[5];
    g[24] = -dt * (T(2) * qV * CL * sp - dotw<WIND>(ea0, ea1, ea2, C)) * iVa * icg + dt * N6 * iVa * iVa * icg;
    g[25] = -dt * sg * N6 * iVa * icg * icg - dt * dotw<WIND>(eg0, eg1, eg2, C) * icg;
    g[26] = -dt * (vH - Va * cg * dotw<WIND>(-sx, cx, T(0), C)) * iVa * icg - T(1);
    g[27] = -dt * qV * CL * cp * icg;
    g[28] = -dt * qV * sp * icg;
    // rows 7-8  (:1170-1184)
    g[29] = -s[8]; g[30] = -s[9]; g[31] = -dt;
}

// Stream the tile's cnt slabs (SLAB*cnt contiguous elements at gslab) out of the LDS rows.
// Wave instruction i covers elements [64*VEC*i, 64*VEC*(i+1)); the (node, element) a lane meets
// repeats every 13 instructions = 8*VEC nodes, so the 13 LDS offsets are formed once.
template <typename T, int VEC>
__device__ __forceinline__ void store_slabs(const T *lds, T *gslab, int cnt, int lane)
{
    typedef typename Vec<T, VEC>::type vec;
    constexpr int PN = SLAB / VEC;       // vectors per node
    constexpr int NPP = 8 * VEC;         // nodes per period
    constexpr int NPER = TILE / NPP;     // periods per tile
    int off[13][VEC];
#pragma unroll
    for (int t = 0; t < 13; t++) {
        const int pp = TILE * t + lane;
        const int nd = pp / PN;
        const int e = (pp - nd * PN) * VEC;
#pragma unroll
        for (int v = 0; v < VEC; v++) off[t][v] = nd * RS + kSlab.c[e + v];
    }
    const int total = PN * cnt;
#pragma unroll
    for (int j = 0; j < NPER; j++) {
        if (j * NPP >= cnt) break;       // wave-uniform
        const T *grp = lds + j * NPP * RS;
#pragma unroll
        for (int t = 0; t < 13; t++) {
            const int p = TILE * (13 * j + t) + lane;
            if (p < total) {
                if constexpr (VEC == 1) {
                    gslab[p] = grp[off[t][0]];
                } else {
                    vec val;
#pragma unroll
                    for (int v = 0; v < VEC; v++) val[v] = grp[off[t][v]];
                    *reinterpret_cast<vec *>(gslab + (long)p * VEC) = val;
                }
            }
        }
    }
}

template <typename T, int WIND, int VEC>
__device__ __forceinline__ void dynamics_tile(const FgArgs &a, T *lds, const T *xrow, T *Frow, T *Grow,
                                              const T *wrow, const TrajDev &tr, int tile, int lane)
{
    const int N = a.N;
    const int k0 = tile * TILE;
    const int cnt = min(TILE, N - k0);
    // window = x[11*k0 .. 11*(k0+cnt)+9): one element before node k0 (keeps the start 16-byte
    // aligned) up to the 8 states of node k0+cnt
    stage_window<T, VEC>(lds, xrow + NI * k0, NI * cnt + 9, lane);
    const T dt = xrow[0];
    __syncthreads();
    const int ll = lane < cnt ? lane : 0;      // idle lanes redo node k0; their results are never stored
    T s[NI], sn[8], we[12];
#pragma unroll
    for (int m = 0; m < NI; m++) s[m] = lds[1 + NI * ll + m];
#pragma unroll
    for (int r = 0; r < 8; r++) sn[r] = lds[1 + NI * (ll + 1) + r];
    if constexpr (WIND == WIND_TABLE) {
#pragma unroll
        for (int f = 0; f < 12; f++) we[f] = wrow[(long)f * (N + 1) + k0 + ll];
    } else {
#pragma unroll
        for (int f = 0; f < 12; f++) we[f] = T(0);
    }
    __syncthreads();                           // the rows below overwrite the window

    const AcCoef &ac = a.ac[tr.ac];
    T f[8], g[32];
    node_eval<T, WIND>(s, dt, T(tr.shear), we, T(ac.inv_m), T(ac.qk), T(ac.Cd0), T(ac.kind), f, g);

    T *row = lds + lane * RS;
#pragma unroll
    for (int i = 0; i < 32; i++) row[i] = g[i];
    row[SL_ZERO] = T(0); row[SL_ONE] = T(1); row[SL_MONE] = T(-1);
#pragma unroll
    for (int r = 0; r < 8; r++) row[SL_F + r] = sn[r] - f[r] * dt - s[r];    // src/problem.cpp:1012-1019
    __syncthreads();

    if (a.needF) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int q = TILE * i + lane;     // defect index within the tile, 8 per node
            if (q < 8 * cnt) Frow[1 + 8 * k0 + q] = lds[(q >> 3) * RS + SL_F + (q & 7)];
        }
    }
    if (a.needG) store_slabs<T, VEC>(lds, Grow + a.c0 + (long)SLAB * k0, cnt, lane);
}

template <typename T, int MISSION, int VEC>
__device__ __forceinline__ void epilogue(const FgArgs &a, T *lds, const T *xrow, T *Frow, T *Grow,
                                         const TrajDev &tr, int lane)
{
    const int N = a.N;
    const T dt = xrow[0];
    const T kT = T(a.kT), kp = T(a.kp);
    T accT = T(0), accP = T(0);
    for (int k0 = 0; k0 <= N; k0 += TILE) {
        const int cnt = min(TILE, N + 1 - k0);
        stage_window<T, VEC>(lds, xrow + NI * k0, NI * cnt + 1, lane);
        __syncthreads();
        const bool act = lane < cnt;
        const int ll = act ? lane : 0;
        const T xs = lds[1 + NI * ll], ys = lds[2 + NI * ll], Th = lds[11 + NI * ll];
        __syncthreads();
        if (act) accT += Th * Th;
        if constexpr (MISSION == MISSION_S10) {
            // src/problemS10.cpp:247-262 (value), :346-375 (gradient)
            const T dx = xs - T(tr.xg), dy = ys - T(tr.yg);
            const T r = sqrt_t(dx * dx + dy * dy);
            const T d = r - T(tr.rg);
            if (act) accP += d * d;
            if (a.needG) {
                lds[3 * lane + 0] = kp * d * dx / r;
                lds[3 * lane + 1] = kp * d * dy / r;
                lds[3 * lane + 2] = kT * Th;
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const int q = TILE * i + lane;
                    if (q < 3 * cnt) Grow[1 + 3 * k0 + q] = lds[q];
                }
                __syncthreads();
            }
        } else {
            // src/problemG7.cpp:364-368: thrust entries; node N's sits after the (xN, yN) pair
            if (a.needG && act) {
                const int k = k0 + lane;
                Grow[k < N ? 3 + k : N + 5] = kT * Th;
            }
        }
    }
    accT = wave_sum(accT);
    accP = wave_sum(accP);

    const long gb = a.c0 + (long)SLAB * N;     // first boundary-row entry
    if constexpr (MISSION == MISSION_S10) {
        if (a.needF) {
            if (lane == 0) Frow[0] = T(0.5) * kT * accT + T(0.5) * kp * accP + T(a.kdt) * dt;   // :264
            if (lane < 11) {                   // src/problemS10.cpp:292-303
                T d = xrow[NI * N + 1 + lane] - xrow[1 + lane];
                if (lane == 5) d = d - T(kTwoPi);
                Frow[8 * N + 1 + lane] = d;
            }
        }
        if (a.needG) {
            if (lane == 0) Grow[0] = T(a.kdt);
            // rows [dt, node 0, node N] = [0, -1, +1]; the dt entry is undefined in the
            // reference (src/problemS10.cpp:397), defined as 0 here
            if (lane < 33) {
                const int c = lane % 3;
                Grow[gb + lane] = c == 0 ? T(0) : (c == 1 ? T(-1) : T(1));
            }
        }
    } else {
        const T x0 = xrow[1], y0 = xrow[2], xf = xrow[NI * N + 1], yf = xrow[NI * N + 2];
        const T dxf = xf - x0, dyf = yf - y0;
        const T dist = sqrt_t(dxf * dxf + dyf * dyf);
        const T cchi = T(tr.cchi), schi = T(tr.schi);
        if (a.needF) {
            if (lane == 0) Frow[0] = kT * T(0.5) * accT + T(a.kv) * T(N) * dt / dist;          // :249
            if (lane < 12) {                   // src/problemG7.cpp:274-294
                T v;
                if (lane == 0) v = dxf - dist * cchi;
                else if (lane == 1) v = dyf - dist * schi;
                else if (lane == 11) {
                    const T ex = T(tr.xg) - x0, ey = T(tr.yg) - y0;
                    v = dist - sqrt_t(ex * ex + ey * ey);
                } else v = xrow[NI * N + 1 + lane] - xrow[1 + lane];
                Frow[8 * N + 1 + lane] = v;
            }
        }
        if (a.needG) {
            // cost row ends; written with kp where the value uses kv (src/problemG7.cpp:345-381)
            if (lane < 5) {
                const T c3 = kp * T(N) * dt / (dist * dist * dist);
                T v; long idx;
                if (lane == 0)      { v = kp * T(N) / dist;  idx = 0; }
                else if (lane == 1) { v = kp * T(N) * dt * dxf / (dist * dist * dist);  idx = 1; }
                else if (lane == 2) { v = kp * T(N) * dt * dyf / (dist * dist * dist);  idx = 2; }
                else if (lane == 3) { v = -(kp * T(N) * dt * dxf / (dist * dist * dist)); idx = N + 3; }
                else                { v = -(kp * T(N) * dt * dyf / (dist * dist * dist)); idx = N + 4; }
                (void)c3;
                Grow[idx] = v;
            }
            // boundary rows: 5 + 5 + 9*3 + 5 = 42 entries  (src/problemG7.cpp:407-511)
            if (lane < 42) {
                const T ux = dxf / dist, uy = dyf / dist;
                T v;
                if (lane < 10) {
                    const int c = lane % 5;          // [dt, x0, y0, xN, yN]
                    const T trig = lane < 5 ? cchi : schi;
                    const int diag = lane < 5 ? 1 : 2;
                    if (c == 0) v = T(0);
                    else {
                        const bool isx = (c == 1 || c == 3);
                        const T g0 = ((isx ? 1 : 2) == diag ? T(-1) : T(0)) + (isx ? ux : uy) * trig;
                        v = c <= 2 ? g0 : -g0;
                    }
                } else if (lane < 37) {
                    const int c = (lane - 10) % 3;
                    v = c == 0 ? T(0) : (c == 1 ? T(-1) : T(1));
                } else {
                    const int c = lane - 37;         // [dt, x0, y0, xN, yN] of dist - dmax
                    v = c == 0 ? T(0) : (c == 1 ? -ux : (c == 2 ? -uy : (c == 3 ? ux : uy)));
                }
                Grow[gb + lane] = v;
            }
        }
    }
}

template <typename T, int MISSION, int WIND, int VEC>
__global__ __launch_bounds__(TILE) void fg_kernel(const FgArgs a)
{
    __shared__ __attribute__((aligned(16))) T lds[TILE * RS];
    const int lane = threadIdx.x;
    // blocks id and id+8 run on the same XCD: keep all waves of a trajectory on one XCD so its x row
    // is fetched into a single L2 (affinity only; nothing depends on it)
    const unsigned id = blockIdx.x;
    const int per = a.tiles + 1;
    const unsigned q = id >> 3;
    const int role = q % per;
    const int b = 8 * (int)(q / per) + (int)(id & 7);
    if (b >= a.B) return;
    const T *xrow = static_cast<const T *>(a.X) + (long)b * a.ldx;
    T *Frow = static_cast<T *>(a.F) + (long)b * a.ldf;
    T *Grow = static_cast<T *>(a.G) + (long)b * a.ldg;
    const TrajDev tr = a.traj[b];
    if (role < a.tiles) {
        const T *wrow = WIND == WIND_TABLE ? static_cast<const T *>(a.wind) + (long)b * 12 * (a.N + 1) : nullptr;
        dynamics_tile<T, WIND, VEC>(a, lds, xrow, Frow, Grow, wrow, tr, role, lane);
    } else {
        epilogue<T, MISSION, VEC>(a, lds, xrow, Frow, Grow, tr, lane);
    }
}

template <typename T>
__global__ void objectives_kernel(const T *F, long ldf, T *obj, int B)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < B) obj[t] = F[(long)t * ldf];
}

template <typename T, int MISSION, int WIND>
hipError_t launch_vec(const FgArgs &a, int vec, dim3 grid, hipStream_t s)
{
    constexpr int VMAX = 16 / sizeof(T);
    if (vec == VMAX) hipLaunchKernelGGL((fg_kernel<T, MISSION, WIND, VMAX>), grid, dim3(TILE), 0, s, a);
    else             hipLaunchKernelGGL((fg_kernel<T, MISSION, WIND, 1>), grid, dim3(TILE), 0, s, a);
    return hipGetLastError();
}

template <typename T, int MISSION>
hipError_t launch_wind(const FgArgs &a, int wind, int vec, dim3 grid, hipStream_t s)
{
    switch (wind) {
    case WIND_NONE:  return launch_vec<T, MISSION, WIND_NONE>(a, vec, grid, s);
    case WIND_SHEAR: return launch_vec<T, MISSION, WIND_SHEAR>(a, vec, grid, s);
    case WIND_TABLE: return launch_vec<T, MISSION, WIND_TABLE>(a, vec, grid, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace

hipError_t launch_fg(const FgArgs &a, int mission, int wind, int dtype, int vec, hipStream_t s)
{
    if (a.B <= 0) return hipSuccess;
    if (a.N < 1 || a.tiles != (a.N + TILE - 1) / TILE) return hipErrorInvalidValue;
    const long groups = (a.B + 7) / 8;
    const long blocks = groups * 8 * (a.tiles + 1);
    if (blocks > 0x7fffffffL) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks);
    if (dtype == 0) {
        return mission == MISSION_S10 ? launch_wind<double, MISSION_S10>(a, wind, vec, grid, s)
                                      : launch_wind<double, MISSION_G7>(a, wind, vec, grid, s);
    }
    return mission == MISSION_S10 ? launch_wind<float, MISSION_S10>(a, wind, vec, grid, s)
                                  : launch_wind<float, MISSION_G7>(a, wind, vec, grid, s);
}

hipError_t launch_objectives(const void *F, long ldf, void *obj, int B, int dtype, hipStream_t s)
{
    if (B <= 0) return hipSuccess;
    const dim3 grid((B + 255) / 256), block(256);
    if (dtype == 0)
        hipLaunchKernelGGL(objectives_kernel<double>, grid, block, 0, s, static_cast<const double *>(F), ldf,
                           static_cast<double *>(obj), B);
    else
        hipLaunchKernelGGL(objectives_kernel<float>, grid, block, 0, s, static_cast<const float *>(F), ldf,
                           static_cast<float *>(obj), B);
    return hipGetLastError();
}

int fg_lds_bytes(int dtype) { return TILE * RS * (dtype == 0 ? 8 : 4); }

}  // namespace tolfg
