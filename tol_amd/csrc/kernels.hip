// kernels.hip -- hand-written gfx950 (CDNA4) kernels for tol's SNOPT user function.
//
// One evaluation produces F and G of a batch of trajectories.  Work decomposition:
//
//   * a TILE is up to 64 consecutive collocation nodes of one trajectory, one wavefront, lane = node
//     (tile_body); the packed fp32 kernels (NP = 2) put two nodes on a lane -- a pair of floats per
//     register pair, v_pk_*_f32 arithmetic -- and a tile is then up to 128 nodes.  Per node it produces the defects F[8k+1..8k+8], the node's contiguous Jacobian
//     slab (104 entries, or 46 in the compact pattern), its objective-gradient entries and its
//     objective terms (ref: problem::dynamicConstraints src/problem.cpp:929-1021,
//     problem::dynamicsGradients src/problem.cpp:1035-1208, wind models 0/1/3 src/problem.cpp:480-695,
//     problemS10::cost/costGradient src/problemS10.cpp:227-386, problemG7::... src/problemG7.cpp:225-384).
//   * finalize_body: one wavefront per trajectory, lanes = output entries, adds the tiles' objective
//     terms in one fixed order (deterministic, no floating-point atomics), handles the last node's terms
//     and writes the boundary rows and their gradients (ref: src/problemS10.cpp:273-305,395-415;
//     src/problemG7.cpp:258-296,393-513).
//   * batched path: fg_kernel, one 64-lane workgroup per tile, ONE launch per evaluation: the tile wave
//     that arrives last at its trajectory's counter runs finalize_body (Publish / sum_partials: payload
//     through write-through stores, polled, nobody waits).  One tile per workgroup on purpose:
//     workgroups that walk several tiles, or several waves that start together, stay in phase, and
//     measured 5-10 % slower.  Tiles are dealt to the 8 XCDs in contiguous eighths.  When the outputs
//     exceed the Infinity Cache the launch requests more LDS than a tile uses to cap the resident
//     waves per CU (the kernel is bound by the HBM write path, which serves fewer concurrent store
//     streams better) and the slab stream is non-temporal; when they fit, plain stores, no cap, and the
//     waves take issue priorities from their SIMD slots so that they do not run in step (FgArgs::stagger).
//     Launches within the cache with 11-20 tile waves per CU are bound by how many of them a CU holds: the fp64
//     reference-pattern kernels send a tile's rows through LDS in two 32-node passes (FgArgs::sub_nodes: 9.2 instead of
//     14.8 KB per wave) and are held to 128 VGPRs (min_waves_per_simd), so that 16 fit instead of 10.
//   * callback path (a few short trajectories): fg_single_kernel, whole trajectory per workgroup,
//     one launch, optional completion word for the spinning host; the callback's single trajectory of 100+ nodes
//     with the Jacobian wanted goes through fg_kernel as 5-8 tile workgroups on different CUs instead (every tile
//     wave fetches the finalizer's x values at its start: run_tile).
//   * set-up kernels: x0_kernel (initial guesses, one thread per node, bitwise the serial walk), bounds_kernel;
//     store_shape_kernel is a measurement aid (the launch's shape with only the slab stores in it).
//   * a mixed batch (MISSION_MIXED) reads each trajectory's mission from its record; a tile is one
//     trajectory, so the branch is wave-uniform.
//   * the SNOPT-facing layouts are node-major (x[11k+1+m], G slab c0+104k), so a lane-per-node
//     access is 88 B / 832 B strided.  The x window is therefore loaded with contiguous 16-byte
//     loads and transposed through LDS, and the slabs (83 % of all bytes) are written with
//     contiguous 16-byte stores, 1 KiB per wave instruction.  Of the 104 slab elements
//     only 32 are computed per node; the 58 structural zeros and the +-1 constants are injected from
//     a compile-time table while streaming out (SlabStream: per-lane LDS offsets from the kStream table,
//     loaded with the x window), so the LDS exchange is 35 elements per node (14.8 KB per wave at ts = 200
//     in fp64).  F (64 B per node) and the objective-gradient entries (24 B per
//     node) go straight from registers: every lane's piece is contiguous with its neighbour's, so
//     whole lines are completed inside L2 by consecutive instructions of the same wave.
//   * air-frame coefficients and per-trajectory constants are wave-uniform: they arrive through
//     the kernarg segment / scalar loads and live in SGPRs (cheaper than an LDS copy).
//   * no MFMA: ~1 flop/byte, HBM-bound.
//
// The maths is the vector form stated in oracle/tolfg_oracle.c; W and grad W are held constant in
// the Jacobian exactly as the reference's tabulated entries do.
#include "kernels.h"
#include "slab_table.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

namespace tolfg {

namespace {

constexpr int TILE = kTileNodes;    // nodes per dynamics tile = wavefront width (kernels.h)
#ifndef TOLFG_MIN_WAVES_PER_SIMD
#define TOLFG_MIN_WAVES_PER_SIMD 2  // register budget; the fp64 tile uses ~100 VGPRs, so this never binds
#endif
constexpr int NI = 11;             // variables per node   (problems/*/snopt.param:3)
constexpr double kGrav = 9.81;     // include/problem.h:72
constexpr double kTwoPi = 6.283185307179586476925286766559;

// LDS row of one node (elements): 32 computed Jacobian values and the 3 constants (slab_table.h).
constexpr int RS = 35;             // odd stride: conflict-free ds_write_b64 across lanes
// a wave's LDS region in the one-workgroup-per-trajectory kernel: a spare row + 64 rows, whole 16-byte vectors
constexpr int WAVE_LDS = ((TILE + 1) * RS + 3) & ~3;
// an empty objective-partial slot of the polling fused path: a NaN no arithmetic produces, both halves
// equal so that hipMemsetD32 can write it
constexpr unsigned long long kEmptySlot = 0xFFFBADADFFFBADADull;

__device__ constexpr SlabTableFull kSlabFull = make_slab_table();
__device__ constexpr SlabTableCompact kSlabCompact = make_compact_table();

constexpr int gcd_c(int a, int b) { return b == 0 ? a : gcd_c(b, a % b); }

// The kernels that can send a tile's rows through LDS in two passes (FgArgs::sub_nodes) and are held to 128 VGPRs so that 16 of
// their waves fit a CU: fp64, reference pattern, the wind models the reference itself uses (none, shear).
template <typename T, int WIND, int PAT> constexpr bool two_pass_family()
{
    return sizeof(T) == 8 && PAT == PATTERN_REFERENCE && (WIND == WIND_NONE || WIND == WIND_SHEAR);
}

#ifdef TOLFG_STAMPS
#define TOLFG_STAMP(a, slot)                                                                         \
    do {                                                                                             \
        unsigned long long t_;                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        if (threadIdx.x == 0) (a).stamps[(size_t)blockIdx.x * 10 + (slot)] = t_;                     \
    } while (0)
#define TOLFG_REALTIME(a, slot)                                                                      \
    do {                                                                                             \
        unsigned long long t_;                                                                       \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
        if (threadIdx.x == 0) (a).stamps[(size_t)blockIdx.x * 10 + (slot)] = t_;                     \
    } while (0)
#define TOLFG_WHERE(a, slot)                                                                         \
    do {                                                                                             \
        unsigned hw_, xcc_;                                                                          \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw_), "=s"(xcc_)); \
        if (threadIdx.x == 0) (a).stamps[(size_t)blockIdx.x * 10 + (slot)] = ((unsigned long long)xcc_ << 32) | hw_;     \
    } while (0)
#define TOLFG_VARIANT(a) ((a).variant)
#else
#define TOLFG_WHERE(a, slot) do {} while (0)
#define TOLFG_STAMP(a, slot) do {} while (0)
#define TOLFG_REALTIME(a, slot) do {} while (0)
#ifdef TOLFG_ABLATE                 // tools/fgbench.cpp -DTOLFG_ABLATE: the ablation switches without the stamps
#define TOLFG_VARIANT(a) ((a).variant)
#else
#define TOLFG_VARIANT(a) 0
#endif
#endif

template <typename T, int VEC> struct Vec { typedef T type __attribute__((ext_vector_type(VEC))); };
template <typename T> struct Vec<T, 1> { typedef T type; };

// What one lane computes with.  NP = 1: one node per lane, the lane value is the element type.  NP = 2 (fp32
// only): two nodes per lane, the lane value is a pair of floats in a 64-bit register pair, and the compiler
// turns the node arithmetic into v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (two nodes per instruction:
// 128-node tiles, half the waves, half the vector issue per node).  Component c of lane l is node l + 64*c.
typedef float f2 __attribute__((ext_vector_type(2)));
template <typename T, int NP> struct LaneOf { typedef T type; };
template <> struct LaneOf<float, 2> { typedef f2 type; };
template <typename L> struct LaneTraits { typedef L elem; enum { NP = 1 }; };
template <> struct LaneTraits<f2> { typedef float elem; enum { NP = 2 }; };
__device__ __forceinline__ double comp(double v, int) { return v; }
__device__ __forceinline__ float comp(float v, int) { return v; }
__device__ __forceinline__ float comp(f2 v, int c) { return v[c]; }
__device__ __forceinline__ void set_comp(double &l, int, double v) { l = v; }
__device__ __forceinline__ void set_comp(float &l, int, float v) { l = v; }
__device__ __forceinline__ void set_comp(f2 &l, int c, float v) { l[c] = v; }

// sin and cos of a double, both at once.  The angles on this path (flight-path angle, course, bank) are a few
// radians at most, so the common case is a two-constant Cody-Waite reduction by pi/2 (exact products through
// FMA) and the classic degree-13 / degree-14 minimax kernels on [-pi/4, pi/4]; |a| >= 2^17 (or NaN) takes the
// library routine with its full-range reduction.  Worst observed difference from libm: 1 ulp; about a third
// of the library routine's instructions, which shortens every tile wave's dependent chain (three per node).
__device__ __forceinline__ void sincos_t(double a, double &s, double &c)
{
#ifdef TOLFG_LIB_SINCOS      // A/B switch of tools/fgbench.cpp
    sincos(a, &s, &c);
    return;
#endif
    if (!(fabs(a) < 131072.0)) {
        sincos(a, &s, &c);
        return;
    }
    const double k = rint(a * 0.63661977236758134308);             // 2/pi
    double r = fma(-k, 1.57079632679489655800e+00, a);              // pi/2, high part
    r = fma(-k, 6.12323399573676603587e-17, r);                     // pi/2, low part
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double sn = fma(z * r, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double cs = w + (((1.0 - w) - hz) + z * z * pc);
    const int q = (int)k & 3;
    const double s0 = (q & 1) ? cs : sn, c0 = (q & 1) ? sn : cs;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}
__device__ __forceinline__ void sincos_t(float a, float &s, float &c) { sincosf(a, &s, &c); }
// Two floats at once: three-constant Cody-Waite reduction by pi/2 and the degree-7 / degree-8 minimax kernels
// on [-pi/4, pi/4] as packed arithmetic (about 1 ulp); the quadrant logic is per component.  Beyond 2^13 (or
// NaN) the library routine, component by component.
__device__ __forceinline__ void sincos_t(f2 a, f2 &s, f2 &c)
{
    if (!(fabsf(a.x) < 8192.0f) || !(fabsf(a.y) < 8192.0f)) {
        float s0, c0, s1, c1;
        sincosf(a.x, &s0, &c0);
        sincosf(a.y, &s1, &c1);
        s = f2{s0, s1}; c = f2{c0, c1};
        return;
    }
    const f2 t = a * 0.636619772367581343f;                         // 2/pi
    const f2 k = f2{__builtin_rintf(t.x), __builtin_rintf(t.y)};
    f2 r = __builtin_elementwise_fma(k, f2(-1.5703125f), a);         // pi/2 in three pieces, the first two exact products
    r = __builtin_elementwise_fma(k, f2(-4.837512969970703125e-4f), r);
    r = __builtin_elementwise_fma(k, f2(-7.54978995489188216e-8f), r);
    const f2 z = r * r;
    f2 ps = __builtin_elementwise_fma(z, f2(-1.9515295891e-4f), f2(8.3321608736e-3f));
    ps = __builtin_elementwise_fma(z, ps, f2(-1.6666654611e-1f));
    const f2 sn = __builtin_elementwise_fma(z * r, ps, r);
    f2 pc = __builtin_elementwise_fma(z, f2(2.443315711809948e-5f), f2(-1.388731625493765e-3f));
    pc = __builtin_elementwise_fma(z, pc, f2(4.166664568298827e-2f));
    const f2 cs = __builtin_elementwise_fma(z * z, pc, __builtin_elementwise_fma(z, f2(-0.5f), f2(1.0f)));
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int q = (int)k[i];
        const bool odd = q & 1;
        const unsigned s0 = __builtin_bit_cast(unsigned, odd ? cs[i] : sn[i]);
        const unsigned c0 = __builtin_bit_cast(unsigned, odd ? sn[i] : cs[i]);
        s[i] = __builtin_bit_cast(float, s0 ^ (((unsigned)q << 30) & 0x80000000u));
        c[i] = __builtin_bit_cast(float, c0 ^ (((unsigned)(q + 1) << 30) & 0x80000000u));
    }
}
__device__ __forceinline__ double sqrt_t(double a) { return sqrt(a); }
__device__ __forceinline__ float sqrt_t(float a) { return sqrtf(a); }
__device__ __forceinline__ f2 sqrt_t(f2 a) { return f2{__builtin_amdgcn_sqrtf(a.x), __builtin_amdgcn_sqrtf(a.y)}; }   // 1 ulp
// reciprocal and quotient: IEEE division for the one-node-per-lane forms (results as in round 2); the packed form
// takes v_rcp_f32 (1 ulp) and one Newton step in packed arithmetic
__device__ __forceinline__ double rcp_t(double a) { return 1.0 / a; }
__device__ __forceinline__ float rcp_t(float a) { return 1.0f / a; }
__device__ __forceinline__ f2 rcp_t(f2 a)
{
    const f2 r = f2{__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)};
    return __builtin_elementwise_fma(__builtin_elementwise_fma(-a, r, f2(1.0f)), r, r);
}
__device__ __forceinline__ double div_t(double a, double b) { return a / b; }
__device__ __forceinline__ float div_t(float a, float b) { return a / b; }
__device__ __forceinline__ f2 div_t(f2 a, f2 b) { return a * rcp_t(b); }

// Output is written once and never read back by the GPU: with TOLFG_NT_STORES the streaming stores
// carry the non-temporal hint so that they do not displace the x rows from L2 / Infinity Cache.
template <bool NT, typename V>
__device__ __forceinline__ void stream_store(V *p, V v)
{
#if defined(TOLFG_STORE_FLAVOR)
    // cache-policy experiments (tools/fgprobe.cpp): 16-byte stores with explicit sc0/sc1/nt bits
    if constexpr (sizeof(V) == 16) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 d = __builtin_bit_cast(u4, v);
        asm volatile("global_store_dwordx4 %0, %1, off " TOLFG_STORE_FLAVOR : : "v"(p), "v"(d) : "memory");
    } else {
        *p = v;
    }
#else
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
#endif
}

// Defects and objective-gradient entries are 16/8-byte pieces that consecutive instructions of the
// same wave complete into whole lines inside L2; TOLFG_NT_SMALL makes them non-temporal too (A/B).
template <typename V>
__device__ __forceinline__ void small_store(V *p, V v)
{
#ifdef TOLFG_NT_SMALL
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
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
    else return e0 * V[0] + e1 * V[1] + e2 * V[2];     // WIND_TABLE, WIND_GRID
}

// Per-node evaluation in two steps so that the defects can leave (and their registers die) before
// the Jacobian values are formed: rates() fills f[0..7] and keeps the shared sub-expressions,
// jacobian() writes the 32 computed Jacobian values straight into the node's LDS row.
// s = x y z Va gam chi phi CL dphi dCL T;  we = the node's 12 ENU wind values (WIND_TABLE only).
// T is the lane type (LaneOf): double, float, or a pair of floats = two nodes.
template <typename T, int WIND> struct NodeCtx {
    T Va, CL, dt, sg, cg, sx, cx, sp, cp, ea0, ea1, ea2, eg0, eg1, eg2;
    T A[3], B[3], C[3], H[3];
    T v0, v1, v2, vA, vB, vC, vH, iVa, icg, qV, q, CD, N5, N6, inv_m, kind, dphi, dCL;

    typedef typename LaneTraits<T>::elem E;      // element type: T itself, or float when T is a pair of floats
    static constexpr int NP = LaneTraits<T>::NP;

    // Trilinear interpolation of the gridded v component and its gradient at one ENU point
    // (ref: src/problem.cpp:551-692).  Returns v, d v / d(east, north, up).
    __device__ __forceinline__ static void grid_wind(const GridDev &gr, E pn, E pe, E pd, E &v, E &dve, E &dvn, E &dvu)
    {
        const E xs = pe + E(gr.e0), ys = pn + E(gr.n0), zs = -pd + E(gr.u0);
        const E dx = E(gr.dx), dy = E(gr.dy), dz = E(gr.dz);
        // lower corner: the first grid coordinate within one spacing below the point, edge cell outside
        const int xi = min(max((int)floor((xs - E(gr.x0)) / dx), 0), gr.nx - 2);
        const int yi = min(max((int)floor((ys - E(gr.y0)) / dy), 0), gr.ny - 2);
        const int zi = min(max((int)floor((zs - E(gr.z0)) / dz), 0), gr.nz - 2);
        const E *g = static_cast<const E *>(gr.v) + ((long)xi * gr.ny + yi) * gr.nz + zi;
        const long sx_ = (long)gr.ny * gr.nz, sy_ = gr.nz;
        const E v0 = g[0], v1 = g[sx_], v2 = g[sy_], v3 = g[sx_ + sy_];
        const E v4 = g[1], v5 = g[sx_ + 1], v6 = g[sy_ + 1], v7 = g[sx_ + sy_ + 1];
        const E ze = (xs - (E(gr.x0) + E(xi) * dx)) / dx, et = (ys - (E(gr.y0) + E(yi) * dy)) / dy;
        const E mu = (zs - (E(gr.z0) + E(zi) * dz)) / dz;
        const E a = E(1) - ze, b = E(1) - et, c = E(1) - mu;
        v = a * b * c * v0 + ze * b * c * v1 + a * et * c * v2 + ze * et * c * v3 +
            a * b * mu * v4 + ze * b * mu * v5 + a * et * mu * v6 + ze * et * mu * v7;
        dve = ((v1 - v0) * b * c + (v3 - v2) * et * c + (v5 - v4) * b * mu + (v7 - v6) * et * mu) / dx;
        dvn = ((v2 - v0) * a * c + (v3 - v1) * ze * c + (v6 - v4) * a * mu + (v7 - v5) * ze * mu) / dy;
        dvu = ((v4 - v0) * a * b + (v5 - v1) * ze * b + (v6 - v2) * a * et + (v7 - v3) * ze * et) / dz;
    }

    __device__ __forceinline__ void rates(const T (&s)[NI], T dt_, T shear, const T (&we)[12], T inv_m_, T qk, T Cd0,
                                          T kind_, T (&f)[8], const GridDev &gr)
    {
        Va = s[3]; CL = s[7]; dt = dt_; inv_m = inv_m_; kind = kind_; dphi = s[8]; dCL = s[9];
        const T Th = s[10];
        sincos_t(s[4], sg, cg);
        sincos_t(s[5], sx, cx);
        sincos_t(s[6], sp, cp);
        ea0 = cx * cg; ea1 = sx * cg; ea2 = -sg;     // along the air-relative velocity
        eg0 = cx * sg; eg1 = sx * sg; eg2 = cg;      // d e_a / d gam = -e_g
        // e_x = (-sx, cx, 0) = (1/cg) d e_a / d chi,  e_h = (cx, sx, 0) = -d e_x / d chi
        T W[3] = {T(0), T(0), T(0)};
#pragma unroll
        for (int i = 0; i < 3; i++) { A[i] = T(0); B[i] = T(0); C[i] = T(0); H[i] = T(0); }
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
        } else if constexpr (WIND == WIND_GRID) {
            // only Wx (= ENU v) is non-zero: dWx/dx_NED = dv/dnorth, dWx/dy_NED = dv/deast, dWx/dz_NED = -dv/dup
            T v, dve, dvn, dvu;
#pragma unroll
            for (int c = 0; c < NP; c++) {
                E v_, dve_, dvn_, dvu_;
                grid_wind(gr, comp(s[0], c), comp(s[1], c), comp(s[2], c), v_, dve_, dvn_, dvu_);
                set_comp(v, c, v_); set_comp(dve, c, dve_); set_comp(dvn, c, dvn_); set_comp(dvu, c, dvu_);
            }
            W[0] = v;
            const T J00 = dvn, J01 = dve, J02 = -dvu;
            A[0] = ea0 * J00; A[1] = ea0 * J01; A[2] = ea0 * J02;
            B[0] = eg0 * J00; B[1] = eg0 * J01; B[2] = eg0 * J02;
            C[0] = -sx * J00; C[1] = -sx * J01; C[2] = -sx * J02;
            H[0] = cx * J00;  H[1] = cx * J01;  H[2] = cx * J02;
        }
        v0 = W[0] + Va * ea0; v1 = W[1] + Va * ea1; v2 = W[2] + Va * ea2;
        vA = dotw<WIND>(v0, v1, v2, A); vB = dotw<WIND>(v0, v1, v2, B);
        vC = dotw<WIND>(v0, v1, v2, C); vH = dotw<WIND>(v0, v1, v2, H);
        const T g9 = T(kGrav);
        iVa = rcp_t(Va); icg = rcp_t(cg);
        qV = qk * Va;               // rho S Va / (2 m)
        q = qV * Va;                // rho S Va^2 / (2 m)
        CD = Cd0 + CL * CL * kind;
        N5 = vB - g9 * cg + q * CL * cp;        // Va * gamdot
        N6 = q * CL * sp - vC;                  // Va cg * chidot
        f[0] = v0; f[1] = v1; f[2] = v2;
        f[3] = Th * inv_m - vA - g9 * sg - q * CD;
        f[4] = N5 * iVa;
        f[5] = N6 * iVa * icg;
        f[6] = s[8];
        f[7] = s[9];
    }

    // g: the node's LDS row (E *), or the pair of rows of the lane's two nodes (RowPair)
    template <typename Row> __device__ __forceinline__ void jacobian(const T (&f)[8], Row g) const
    {
        const T g9 = T(kGrav);
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
        g[29] = -dphi; g[30] = -dCL; g[31] = -dt;
    }
};

// ---- the slab stream.  A tile's slabs are SLABN*cnt contiguous elements of G (gslab); the wave writes them as
// whole, 16-byte aligned vectors of GV elements, 1 KiB per wave instruction, wherever the region sits relative to
// a 16-byte boundary: with the region starting SHIFT elements past a boundary the stream is cut SHIFT elements
// earlier, i.e. vector q covers stream elements [q*GV - SHIFT, q*GV - SHIFT + GV); the few elements before the
// first and after the last whole vector leave as scalars.  Round R of the wave handles vectors 64*R + lane.
// Which LDS slot feeds stream element r = SLABN*node + e is slab_table.h's code(e); the (node, element) pairs a
// lane meets repeat every P rounds (NPP nodes later), so the byte offsets of a lane's P*GV elements -- relative
// to the LDS row of the period's first node, rows laid out one RS-element row per node after one spare row
// (SHIFT > 0 reaches into the node before) -- are a compile-time table per (element size, pattern, SHIFT),
// 16 bits each, which every wave loads with a few coalesced 16-byte loads.  (Round 2 formed these offsets with
// integer arithmetic and byte loads per wave: ~10 VALU instructions per element, 40 % of the fp32 kernel's
// dynamic VALU work.)
template <int ES, int PAT> struct StreamGeom {
    static constexpr int SLABN = PAT == PATTERN_COMPACT ? SLAB_COMPACT : SLAB_FULL;
    static constexpr int GV = 16 / ES;                                     // elements per 16-byte vector
    static constexpr int P = SLABN / gcd_c(SLABN, TILE * GV);              // rounds per period
    static constexpr int NPP = TILE * GV * P / SLABN;                      // nodes per period
    static constexpr int D = (P * GV + 1) / 2;                             // dwords per lane (two offsets each)
    static constexpr int CH = (D + 3) / 4;                                 // 16-byte chunks per lane
    static_assert((SLABN * 4) % GV == 0, "a tile starts at a node number that is a multiple of 4");
};

template <int ES, int PAT> struct StreamTable {
    typedef StreamGeom<ES, PAT> Gm;
    unsigned w[Gm::GV][Gm::CH][TILE][4];      // [SHIFT][chunk][lane][dword]: coalesced 16-byte loads
};

template <int PAT> __device__ __forceinline__ int slab_code_dev(int e)
{
    if constexpr (PAT == PATTERN_COMPACT) return kSlabCompact.c[e];
    else return kSlabFull.c[e];
}

constexpr int slab_code(int pat, int e)
{
    return pat == PATTERN_COMPACT ? make_compact_table().c[e] : make_slab_table().c[e];
}

template <int ES, int PAT> constexpr StreamTable<ES, PAT> make_stream_table()
{
    typedef StreamGeom<ES, PAT> Gm;
    StreamTable<ES, PAT> t{};
    unsigned char code[Gm::SLABN] = {};
    for (int e = 0; e < Gm::SLABN; e++) code[e] = (unsigned char)slab_code(PAT, e);
    for (int sh = 0; sh < Gm::GV; sh++)
        for (int lane = 0; lane < TILE; lane++)
            for (int i = 0; i < Gm::P * Gm::GV; i++) {
                const int r = (TILE * (i / Gm::GV) + lane) * Gm::GV - sh + (i % Gm::GV);   // >= -sh
                const int nd = (r + Gm::SLABN) / Gm::SLABN;                // row index incl. the spare row: floor(r/SLABN) + 1
                const int e = r + Gm::SLABN - nd * Gm::SLABN;
                const unsigned off = (unsigned)((nd * RS + code[e]) * ES);
                const int d = i / 2;
                t.w[sh][d / 4][lane][d % 4] |= (i & 1) ? (off << 16) : off;
            }
    return t;
}

template <int ES, int PAT> __device__ constexpr StreamTable<ES, PAT> kStream = make_stream_table<ES, PAT>();

// TN = nodes a tile may hold (64, or 128 with two nodes per lane): bounds the unrolled rounds.
// load() asks for the lane's offsets -- early, together with the x window, so that they are back long before the
// stream starts and no wait for them ends up behind the wave's own defect stores (one in-order vmcnt);
// run() streams the rows out.
template <typename T, int PAT, bool NT, int TN> struct SlabStream {
    static constexpr int ES = (int)sizeof(T);
    typedef StreamGeom<ES, PAT> Gm;
    static constexpr int GV = Gm::GV, SLABN = Gm::SLABN, P = Gm::P;
    typedef typename Vec<T, GV>::type vec;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    static constexpr int RMAX = (TN * SLABN + GV - 1) / (TILE * GV) + 1;   // rounds a full tile needs at most
    static constexpr int PU = P < RMAX ? P : RMAX;                         // table rounds actually used
    static constexpr int CHU = ((PU * GV + 1) / 2 + 3) / 4;
#ifndef TOLFG_STREAM_AHEAD
#define TOLFG_STREAM_AHEAD 1
#endif
    static constexpr int AHEAD = TOLFG_STREAM_AHEAD < RMAX ? TOLFG_STREAM_AHEAD : 1;     // rounds of LDS reads in flight ahead of the store
    u4 tb[CHU];
    T *gslab;
    int shift;             // the region starts `shift` elements past a 16-byte boundary: wave-uniform

    __device__ __forceinline__ void load(T *gslab_, int lane)
    {
        gslab = gslab_;
        shift = __builtin_amdgcn_readfirstlane((int)((reinterpret_cast<unsigned long long>(gslab_) / ES) % GV));
        const u4 *tp = reinterpret_cast<const u4 *>(&kStream<ES, PAT>.w[0][0][0][0]) + (long)shift * (Gm::CH * TILE) + lane;
#pragma unroll
        for (int c = 0; c < CHU; c++) tb[c] = tp[c * TILE];
    }

    // the vector of round R (LDS gather through the lane's offsets); rounds past the tile read rows nobody wrote,
    // or past the allocation (LDS then returns zeros): harmless, such a vector is never stored
    __device__ __forceinline__ vec gather(const char *lb, int R) const
    {
        const int j = R / P, t = R % P;
        const char *grp = lb + j * Gm::NPP * RS * ES;
        vec val;
#pragma unroll
        for (int v = 0; v < GV; v++) {
            const int i = t * GV + v;
            const unsigned wv = tb[(i / 2) / 4][(i / 2) % 4];
            const unsigned off = (i & 1) ? (wv >> 16) : (wv & 0xffffu);
            val[v] = *reinterpret_cast<const T *>(grp + off);
        }
        return val;
    }

    __device__ __forceinline__ void run(const T *lds, int cnt, int lane) const
    {
        const int E = SLABN * cnt;                         // elements of the region
        const int qhi = (E + shift) / GV;                  // whole vectors are q in [shift ? 1 : 0, qhi)
        const char *lb = reinterpret_cast<const char *>(lds);
        vec *gp = reinterpret_cast<vec *>(gslab - shift) + lane;
        // AHEAD rounds ahead: the LDS reads of rounds R + 1 .. R + AHEAD are in flight while round R's vector is stored (a
        // wave alone on its SIMD -- small batches, the SNOPT callback -- otherwise pays the LDS latency once per round)
        vec q[AHEAD];
#pragma unroll
        for (int d = 0; d < AHEAD; d++) q[d] = gather(lb, d);
#pragma unroll
        for (int R = 0; R < RMAX; R++) {
            if (TILE * R >= qhi) break;                    // wave-uniform
            const vec cur = q[R % AHEAD];
            if (R + AHEAD < RMAX) q[R % AHEAD] = gather(lb, R + AHEAD);
            const int qv = TILE * R + lane;
            const bool whole = R > 0 && TILE * (R + 1) <= qhi;     // wave-uniform: no lane is cut off
            if (whole || (qv < qhi && (R > 0 || shift == 0 || lane > 0))) stream_store<NT>(gp + TILE * R, cur);
        }
        // the elements before the first and after the last whole vector
        const int ntail = E + shift - qhi * GV;            // in [0, GV)
        if (shift > 0 && lane < GV - shift) {
            gslab[lane] = lds[RS + slab_code_dev<PAT>(lane)];
        } else if (lane >= GV && lane < GV + ntail) {
            const int e = qhi * GV - shift + (lane - GV) - SLABN * (cnt - 1);
            gslab[(long)SLABN * (cnt - 1) + e] = lds[cnt * RS + slab_code_dev<PAT>(e)];
        }
    }
};

// The 8 defects of one node are 8 contiguous elements at F[1+8k]; element 1+8k sits one element
// past a 16-byte boundary, so the aligned middle goes out as vectors and the ends as scalars.
template <typename T, int VEC>
__device__ __forceinline__ void store_defects(T *p, const T (&d)[8])
{
    if constexpr (VEC == 2) {
        typedef typename Vec<T, 2>::type v2;
        small_store(p, d[0]);
        small_store(reinterpret_cast<v2 *>(p + 1), v2{d[1], d[2]});
        small_store(reinterpret_cast<v2 *>(p + 3), v2{d[3], d[4]});
        small_store(reinterpret_cast<v2 *>(p + 5), v2{d[5], d[6]});
        small_store(p + 7, d[7]);
    } else if constexpr (VEC == 4) {
        typedef typename Vec<T, 2>::type v2;
        typedef typename Vec<T, 4>::type v4;
        p[0] = d[0];
        *reinterpret_cast<v2 *>(p + 1) = v2{d[1], d[2]};
        *reinterpret_cast<v4 *>(p + 3) = v4{d[3], d[4], d[5], d[6]};
        p[7] = d[7];
    } else {
#pragma unroll
        for (int r = 0; r < 8; r++) p[r] = d[r];
    }
}

// Fused path: how a tile wave hands its objective terms to whichever wave finalizes the trajectory.
// Nobody waits: the payload leaves as write-through (sc1) stores into this tile's own slot, then one
// agent-scope atomic add on the trajectory's arrival counter, whose returned value is looked at only
// when the wave has issued all its stores.  The wave whose add returned tiles-1 finalizes: it polls
// the slots (sc1 loads) until none is empty -- every tile stored its payload before it arrived, so
// that is at most one store round trip -- and empties them again for the next launch.
struct Publish {
    double *slot;          // this tile's [2] partial slot, or nullptr (not fused)
    unsigned *counter;     // the trajectory's arrival counter
    unsigned old;          // value the add returned (lane 0)
    __device__ __forceinline__ void arrive(int lane, bool needF, double sumT, double sumP)
    {
        if (!slot || lane != 0) return;
        if (needF) {
            __hip_atomic_store(slot + 0, sumT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(slot + 1, sumP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// One tile: everything a wavefront does for `cnt` consecutive nodes of trajectory b.  lds is the
// wave's own TILE*RS-element region; sumT / sumP return the tile's objective terms (wave-uniform).

// Where a tile lies: trajectory, first node, tiling of that trajectory, number of the trajectory's first tile.
struct TileAt { int b, k0, tiles, nt, first; };

__device__ __forceinline__ long body_tiles(const FgArgs &a) { return (long)(a.B - a.tail_count) * a.tiles; }
__device__ __forceinline__ long total_tiles(const FgArgs &a) { return body_tiles(a) + (long)a.tail_count * a.tail_tiles; }

__device__ __forceinline__ TileAt tile_at(const FgArgs &a, int item)
{
    const int tb = a.B - a.tail_count, body = tb * a.tiles;
    if (item < body) {                                       // wave-uniform
        const int b = item / a.tiles;
        return {b, (item - b * a.tiles) * a.nt, a.tiles, a.nt, b * a.tiles};
    }
    const int r = item - body, q = r / a.tail_tiles;
    return {tb + q, (r - q * a.tail_tiles) * a.tail_nt, a.tail_tiles, a.tail_nt, body + q * a.tail_tiles};
}

__device__ __forceinline__ long first_tile_of(const FgArgs &a, int b)
{
    const int tb = a.B - a.tail_count;
    return b < tb ? (long)b * a.tiles : body_tiles(a) + (long)(b - tb) * a.tail_tiles;
}

// The LDS rows of a lane's two nodes, written through one subscript (NodeCtx::jacobian, NP = 2)
struct RowPair {
    float *r0, *r1;
    struct Ref {
        float *p0, *p1;
        __device__ __forceinline__ void operator=(f2 v) const { *p0 = v.x; *p1 = v.y; }
    };
    __device__ __forceinline__ Ref operator[](int i) const { return Ref{r0 + i, r1 + i}; }
};

// One tile: everything a wavefront does for `cnt` consecutive nodes of trajectory b.  lds is the wave's own region
// (a spare row and a.nt rows of RS elements); sumT / sumP return the tile's objective terms (wave-uniform).
// NP nodes per lane (LaneOf): node `lane` and, with NP = 2, node `lane + 64` of the tile.
template <typename T, int MISSION, int WIND, int VEC, int PAT, bool NT, int NP = 1>
__device__ __forceinline__ void tile_body(const FgArgs &a, T *lds, int item, int lane, T &sumT_out, T &sumP_out, Publish &pub)
{
    typedef typename Vec<T, VEC>::type vec;
    typedef typename LaneOf<T, NP>::type L;
    constexpr int TN = TILE * NP;                                             // nodes a tile may hold
    constexpr int SLABN = StreamGeom<(int)sizeof(T), PAT>::SLABN;
    constexpr int NW = ((NI * TN + 9 + VEC - 1) / VEC + TILE - 1) / TILE;     // window vectors per lane
    const int N = a.N;
    const TileAt at = tile_at(a, item);
    const int b = at.b, k0 = at.k0;
    const int cnt = min(at.nt, N - k0);
    const T *xrow = static_cast<const T *>(a.X) + (long)b * a.ldx;
    T *Frow = static_cast<T *>(a.F) + (long)b * a.ldf;
    T *Grow = static_cast<T *>(a.G) + (long)b * a.ldg;

    const TrajDev tr = a.traj[b];
    // the trajectory's mission: a template constant, or (mixed batch) wave-uniform from its record
    const int ms = MISSION == MISSION_MIXED ? __builtin_amdgcn_readfirstlane(tr.mission) : MISSION;
    // The slab stream's per-lane offsets (SlabStream) are asked for FIRST, the x window right behind them: both are then
    // in flight together, and the one wait before the window's LDS writes covers both (loads return in order).  Asked
    // for after the window, they were issued only once the window had arrived: a second memory latency on every wave's
    // critical path.
    SlabStream<T, PAT, NT, TN> stream;
#if defined(TOLFG_STAMPS) || defined(TOLFG_ABLATE)
    const bool late_table = TOLFG_VARIANT(a) & 8192;       // A/B: round-3's first order (offsets asked for after the window)
#else
    constexpr bool late_table = false;
#endif
    if (a.needG && !late_table) stream.load(Grow + a.c0[ms] + (long)SLABN * k0, lane);
    const T dt = xrow[0];

    // ---- x window = x[11*k0 .. 11*(k0+cnt)+9): one element before node k0 (keeps the start 16-byte
    // aligned) up to the 8 states of node k0+cnt, rounded up to whole vectors (stays inside the row);
    // contiguous 16-byte loads -> LDS -> this lane's node(s) (transpose)
    TOLFG_REALTIME(a, 7);
    TOLFG_WHERE(a, 9);
    TOLFG_STAMP(a, 0);
    {
        const int nvec = (NI * cnt + 9 + VEC - 1) / VEC;
        const T *xwin = xrow + NI * k0;
        vec win[NW];
#pragma unroll
        for (int j = 0; j < NW; j++) {
            const int i = lane + TILE * j;
#if defined(TOLFG_STAMPS) || defined(TOLFG_ABLATE)
            win[j] = vec{};
            if (TOLFG_VARIANT(a) & 4096) continue;     // ablation: no x window
#endif
            if (i < nvec) win[j] = *reinterpret_cast<const vec *>(xwin + (long)i * VEC);
        }
#pragma unroll
        for (int j = 0; j < NW; j++) {
            const int i = lane + TILE * j;
            if (i < nvec) *reinterpret_cast<vec *>(lds + i * VEC) = win[j];
        }
    }
    if (a.needG && late_table) stream.load(Grow + a.c0[ms] + (long)SLABN * k0, lane);
    __syncthreads();
    TOLFG_STAMP(a, 1);
    bool act[NP];
    int ll[NP];                                // idle components redo node k0; nothing of theirs is stored
#pragma unroll
    for (int c = 0; c < NP; c++) { act[c] = lane + TILE * c < cnt; ll[c] = act[c] ? lane + TILE * c : 0; }
    L s[NI], sn[8], we[12];
#pragma unroll
    for (int c = 0; c < NP; c++) {
#pragma unroll
        for (int m = 0; m < NI; m++) set_comp(s[m], c, lds[1 + NI * ll[c] + m]);
#pragma unroll
        for (int r = 0; r < 8; r++) set_comp(sn[r], c, lds[1 + NI * (ll[c] + 1) + r]);
        if constexpr (WIND == WIND_TABLE) {
            const T *wrow = static_cast<const T *>(a.wind) + (long)b * 12 * (N + 1);
#pragma unroll
            for (int f = 0; f < 12; f++) set_comp(we[f], c, wrow[(long)f * (N + 1) + k0 + ll[c]]);
        } else {
#pragma unroll
            for (int f = 0; f < 12; f++) set_comp(we[f], c, T(0));
        }
    }
    __syncthreads();                           // the rows below overwrite the window
    TOLFG_STAMP(a, 2);

    const AcCoef &ac = a.ac[tr.ac];
    L f[8];
    NodeCtx<L, WIND> nc;
#if defined(TOLFG_STAMPS) || defined(TOLFG_ABLATE)
    if (TOLFG_VARIANT(a) & 256) {              // ablation: no arithmetic, outputs are garbage
#pragma unroll
        for (int r = 0; r < 8; r++) f[r] = s[r];
    } else
#endif
    nc.rates(s, L(dt), L(T(tr.shear)), we, L(T(ac.inv_m)), L(T(ac.qk)), L(T(ac.Cd0)), L(T(ac.kind)), f, a.grid);

    // ---- defects leave first (src/problem.cpp:1012-1019); sn dies here
    if (a.needF && !(TOLFG_VARIANT(a) & 512)) {
        L d8[8];
#pragma unroll
        for (int r = 0; r < 8; r++) d8[r] = sn[r] - f[r] * L(dt) - s[r];
#pragma unroll
        for (int c = 0; c < NP; c++)
            if (act[c]) {
                T e8[8];
#pragma unroll
                for (int r = 0; r < 8; r++) e8[r] = comp(d8[r], c);
                store_defects<T, VEC>(Frow + 1 + 8 * (k0 + lane + TILE * c), e8);
            }
    }

    // ---- objective terms of this tile's nodes (node N is the finalizing wave's)
    const T kT = T(a.kT[ms]), kp = T(a.kp[ms]);
    const L th2 = s[10] * s[10];
    T sumT = T(0), sumP = T(0);
#pragma unroll
    for (int c = 0; c < NP; c++) sumT += act[c] ? comp(th2, c) : T(0);
    if (ms == MISSION_S10) {
        // src/problemS10.cpp:247-262 (value), :346-375 (gradient)
        const L dx = s[0] - L(T(tr.xg)), dy = s[1] - L(T(tr.yg));
        const L r = sqrt_t(dx * dx + dy * dy);
        const L d = r - L(T(tr.rg));
        const L d2 = d * d, gx = div_t(L(kp) * d * dx, r), gy = div_t(L(kp) * d * dy, r), gt = L(kT) * s[10];
#pragma unroll
        for (int c = 0; c < NP; c++) {
            if (act[c]) sumP += comp(d2, c);
            if (a.needG && act[c] && !(TOLFG_VARIANT(a) & 1024)) {
                T *gc = Grow + 1 + 3 * (k0 + lane + TILE * c);
                small_store(gc + 0, comp(gx, c));
                small_store(gc + 1, comp(gy, c));
                small_store(gc + 2, comp(gt, c));
            }
        }
    } else {
        // src/problemG7.cpp:364-368: one thrust entry per node, after (dt, x0, y0)
        const L gt = L(kT) * s[10];
#pragma unroll
        for (int c = 0; c < NP; c++)
            if (a.needG && act[c]) Grow[3 + k0 + lane + TILE * c] = comp(gt, c);
    }
    if (a.needF) {
        sumT = wave_sum(sumT);
        sumP = wave_sum(sumP);
    }
    sumT_out = sumT;
    sumP_out = sumP;
    pub.arrive(lane, a.needF != 0, (double)sumT, (double)sumP);

    if (a.needG) {
        // idle lanes (lane >= cnt) leave no row: the workgroup's LDS holds a.nt + 1 rows, not TN + 1; the second
        // component of a lane whose second node lies beyond the tile writes the spare row, which nobody reads
        if constexpr (NP == 1 && two_pass_family<T, WIND, PAT>()) {
            // (only the kernel family that is also held to 128 VGPRs, min_waves_per_simd: fp32 rows are half the size and never the
            // limit, the other fp64 variants could not use the residency anyway, and the pass loop costs registers -- it kept the
            // fp32 kernels at 141 VGPRs instead of 88)
            // The rows go through LDS a.sub_nodes nodes at a time (0 = the whole tile at once): a tile of 52 fp64 nodes
            // holds 14.8 KB of rows, which caps a CU at 10 resident tile waves; with 32-node passes the same LDS space
            // (9.2 KB) serves the tile in two passes and 16 waves fit.  A pass is a sub-tile to the stream: 32 nodes of
            // slabs are a whole number of 16-byte vectors for every element size and pattern, so the second pass starts
            // at the same offset from a 16-byte boundary as the first and SlabStream::run streams it unchanged.  The
            // lanes outside the pass idle through jacobian() (the launches this is for are not VALU-bound).
            const int sub = a.sub_nodes > 0 ? a.sub_nodes : TN;
            T *const gtile = stream.gslab;
            for (int n0 = 0; n0 < cnt; n0 += sub) {          // wave-uniform: one or two passes
                if (n0 > 0) __syncthreads();                 // the pass before has read its rows
                const int c = min(sub, cnt - n0);
                if (lane >= n0 && lane < n0 + c) {
                    T *row = lds + (lane - n0 + 1) * RS;     // rows follow one spare row (SlabStream)
#if defined(TOLFG_STAMPS) || defined(TOLFG_ABLATE)
                    if (TOLFG_VARIANT(a) & 256) {
#pragma unroll
                        for (int i = 0; i < 32; i++) row[i] = s[i % NI];
                    } else
#endif
                    nc.jacobian(f, row);
                    row[SL_ZERO] = T(0); row[SL_ONE] = T(1); row[SL_MONE] = T(-1);
                }
                __syncthreads();
                TOLFG_STAMP(a, 3);
                __builtin_amdgcn_sched_barrier(0);
                stream.gslab = gtile + (long)SLABN * n0;
                if (!(TOLFG_VARIANT(a) & 2048)) stream.run(lds, c, lane);
            }
        } else {
            if (act[0]) {
                T *row = lds + (lane + 1) * RS;          // rows follow one spare row (SlabStream)
                if constexpr (NP == 1) {
#if defined(TOLFG_STAMPS) || defined(TOLFG_ABLATE)
                    if (TOLFG_VARIANT(a) & 256) {
#pragma unroll
                        for (int i = 0; i < 32; i++) row[i] = s[i % NI];
                    } else
#endif
                    nc.jacobian(f, row);
                    row[SL_ZERO] = T(0); row[SL_ONE] = T(1); row[SL_MONE] = T(-1);
                } else {
                    T *row1 = act[NP - 1] ? row + TILE * RS : lds;
                    nc.jacobian(f, RowPair{row, row1});
                    row[SL_ZERO] = T(0); row[SL_ONE] = T(1); row[SL_MONE] = T(-1);
                    row1[SL_ZERO] = T(0); row1[SL_ONE] = T(1); row1[SL_MONE] = T(-1);
                }
            }
            __syncthreads();
            TOLFG_STAMP(a, 3);
            __builtin_amdgcn_sched_barrier(0);
            if (!(TOLFG_VARIANT(a) & 2048)) stream.run(lds, cnt, lane);
        }
        TOLFG_STAMP(a, 4);
    }
    TOLFG_STAMP(a, 5);
#ifdef TOLFG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TOLFG_STAMP(a, 6);
    TOLFG_REALTIME(a, 8);
#endif
}

// One wavefront per trajectory, lanes = output entries: objective value, the last node's
// objective-gradient entries, boundary rows and their gradients.  O(1) per trajectory except the
// in-order sum over the tiles' objective partials (deterministic, no atomics).
// sumT / sumP: the objective terms of nodes 0..N-1, already added up in tile order.
template <typename T, int MISSION, int PAT>
__device__ __forceinline__ void finalize_body(const FgArgs &a, int b, int lane, T sumT, T sumP, const T *edge = nullptr)
{
    // edge (optional): [dt | node 0's 11 values | node N's 11 values], fetched ahead of time by the caller
    // (the callback kernel reads x from host memory: one PCIe round trip less on its critical path)
    constexpr int SLABN = PAT == PATTERN_COMPACT ? SLAB_COMPACT : SLAB_FULL;
    const int N = a.N;
    const T *x = static_cast<const T *>(a.X) + (long)b * a.ldx;
    T *F = static_cast<T *>(a.F) + (long)b * a.ldf;
    T *G = static_cast<T *>(a.G) + (long)b * a.ldg;
    const TrajDev tr = a.traj[b];
    const int ms = MISSION == MISSION_MIXED ? __builtin_amdgcn_readfirstlane(tr.mission) : MISSION;
    auto X0 = [&](int m) { return edge ? edge[1 + m] : x[1 + m]; };               // node 0, m = 0..10
    auto XN = [&](int m) { return edge ? edge[12 + m] : x[NI * N + 1 + m]; };     // node N
    const T dt = edge ? edge[0] : x[0];
    const T kT = T(a.kT[ms]), kp = T(a.kp[ms]);
    const T TN = XN(10);
    const long gb = a.c0[ms] + (long)SLABN * N;     // first boundary-row entry

    if (a.needF) sumT += TN * TN;

    if (ms == MISSION_S10) {
        const T dx = XN(0) - T(tr.xg), dy = XN(1) - T(tr.yg);
        const T r = sqrt_t(dx * dx + dy * dy);
        const T d = r - T(tr.rg);
        if (a.needF) {
            sumP += d * d;
            if (lane == 0) {                                                                  // src/problemS10.cpp:264
                const T obj = T(0.5) * kT * sumT + T(0.5) * kp * sumP + T(a.kdt[ms]) * dt;
                F[0] = obj;
                if (a.obj) static_cast<T *>(a.obj)[b] = obj;
            }
            if (lane < 11) {                                                                  // :292-303
                T v = XN(lane) - X0(lane);
                if (lane == 5) v = v - T(kTwoPi);
                F[8 * N + 1 + lane] = v;
            }
        }
        if (a.needG) {
            if (lane == 0) G[0] = T(a.kdt[ms]);
            if (lane < 3) G[1 + 3 * N + lane] = lane == 0 ? kp * d * dx / r : (lane == 1 ? kp * d * dy / r : kT * TN);
            // rows [dt, node 0, node N] = [0, -1, +1]; the dt entry is undefined in the
            // reference (src/problemS10.cpp:397), defined as 0 here and dropped by the compact pattern
            if constexpr (PAT == PATTERN_COMPACT) {
                if (lane < 22) G[gb + lane] = (lane & 1) ? T(1) : T(-1);
            } else if (lane < 33) {
                const int c = lane % 3;
                G[gb + lane] = c == 0 ? T(0) : (c == 1 ? T(-1) : T(1));
            }
        }
    } else {
        const T x0 = X0(0), y0 = X0(1), xf = XN(0), yf = XN(1);
        const T dxf = xf - x0, dyf = yf - y0;
        const T dist = sqrt_t(dxf * dxf + dyf * dyf);
        const T cchi = T(tr.cchi), schi = T(tr.schi);
        if (a.needF) {
            if (lane == 0) {                                                                  // src/problemG7.cpp:249
                const T obj = kT * T(0.5) * sumT + T(a.kv[ms]) * T(N) * dt / dist;
                F[0] = obj;
                if (a.obj) static_cast<T *>(a.obj)[b] = obj;
            }
            if (lane < 12) {                                                                  // :274-294
                T v;
                if (lane == 0) v = dxf - dist * cchi;
                else if (lane == 1) v = dyf - dist * schi;
                else if (lane == 11) {
                    const T ex = T(tr.xg) - x0, ey = T(tr.yg) - y0;
                    v = dist - sqrt_t(ex * ex + ey * ey);
                } else v = XN(lane) - X0(lane);
                F[8 * N + 1 + lane] = v;
            }
        }
        if (a.needG) {
            // objective row; written with kp where the value uses kv (src/problemG7.cpp:345-381)
            if (lane < 6) {
                const T d3 = dist * dist * dist;
                T v; long idx;
                if (lane == 0)      { v = kp * T(N) / dist;                 idx = 0; }
                else if (lane == 1) { v = kp * T(N) * dt * dxf / d3;        idx = 1; }
                else if (lane == 2) { v = kp * T(N) * dt * dyf / d3;        idx = 2; }
                else if (lane == 3) { v = -(kp * T(N) * dt * dxf / d3);     idx = N + 3; }
                else if (lane == 4) { v = -(kp * T(N) * dt * dyf / d3);     idx = N + 4; }
                else                { v = kT * TN;                          idx = N + 5; }
                G[idx] = v;
            }
            // boundary rows: 5 + 5 + 9*3 + 5 = 42 entries  (src/problemG7.cpp:407-511); the compact
            // pattern drops each row's dt entry (always 0): 4 + 4 + 9*2 + 4 = 30
            if (lane < 42) {
                const T ux = dxf / dist, uy = dyf / dist;
                T v;
                int c, pos;                          // c: 0 = dt column; pos: index in the compact layout
                if (lane < 10) {
                    c = lane % 5;                    // [dt, x0, y0, xN, yN]
                    pos = 4 * (lane / 5) + c - 1;
                    const T trig = lane < 5 ? cchi : schi;
                    const int diag = lane < 5 ? 1 : 2;
                    if (c == 0) v = T(0);
                    else {
                        const bool isx = (c == 1 || c == 3);
                        const T g0 = ((isx ? 1 : 2) == diag ? T(-1) : T(0)) + (isx ? ux : uy) * trig;
                        v = c <= 2 ? g0 : -g0;
                    }
                } else if (lane < 37) {
                    c = (lane - 10) % 3;
                    pos = 8 + 2 * ((lane - 10) / 3) + c - 1;
                    v = c == 0 ? T(0) : (c == 1 ? T(-1) : T(1));
                } else {
                    c = lane - 37;                   // [dt, x0, y0, xN, yN] of dist - dmax
                    pos = 26 + c - 1;
                    v = c == 0 ? T(0) : (c == 1 ? -ux : (c == 2 ? -uy : (c == 3 ? ux : uy)));
                }
                if constexpr (PAT == PATTERN_COMPACT) {
                    if (c != 0) G[gb + pos] = v;
                } else {
                    G[gb + lane] = v;
                }
            }
        }
    }
}

// Callback completion (a.done): each leaving wave (or workgroup, through `leader`) makes its stores
// visible at system scope (FENCE; the single-workgroup kernel has done so before its barrier), then
// adds to the departure counter; the last one resets the counter and writes the completion word the
// host spins on.  `s_waitcnt vmcnt(0)` instead of the release fence is NOT enough: the host then saw
// the word before the last stores (F[0] still 0, tests/test_compact_pattern.py on MI355X).
template <bool FENCE = true>
__device__ __forceinline__ void signal_done(const FgArgs &a, int participants, bool leader)
{
    if constexpr (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (leader) {
        unsigned *dep = a.counter + a.B;
        bool last = true;
        if (participants > 1) {
            // acquire-release: the last arrival's completion word is ordered after every other participant's stores by
            // the memory model, not just by what the hardware happens to do
            last = __hip_atomic_fetch_add(dep, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(participants - 1);
            if (last) __hip_atomic_store(dep, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (last) __hip_atomic_store(a.done, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Sum of a trajectory's tile partials in one fixed order (lane l adds tiles l, l+64, ...; then the
// butterfly), so the objective does not depend on which wave finalizes or on arrival order.
// POLL false: plain loads (an earlier launch or this workgroup's LDS wrote them); true: waves of this
// launch on other XCDs wrote them -- sc1 loads that poll until the slot is no longer empty, then
// empty it again for the next launch.
template <typename T, bool POLL>
__device__ __forceinline__ void sum_partials(const double *part, int tiles, int lane, T &sumT, T &sumP, unsigned *status = nullptr)
{
    T st = T(0), sp = T(0);
    bool lost = false;
    for (int t = lane; t < tiles; t += TILE) {
        if constexpr (POLL) {
            unsigned long long *q = reinterpret_cast<unsigned long long *>(const_cast<double *>(part)) + 2 * t;
            unsigned long long b0, b1;
            // every tile wave stores its payload before it arrives, so the slots fill within a store
            // round trip; the spin bound only guards against an input whose objective terms ARE the marker
            // (a NaN with that payload in x): the launch then ends, F[0] is that NaN, and the batch's status
            // word tells the host that a partial was lost (tolfg_batch_status, *Status = -2 in the callback)
            for (int spin = 0; spin < (1 << 16); spin++) {
                b0 = __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b1 = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (b0 != kEmptySlot && b1 != kEmptySlot) break;
                __builtin_amdgcn_s_sleep(8);
            }
            lost |= b0 == kEmptySlot || b1 == kEmptySlot;
            __hip_atomic_store(q + 0, kEmptySlot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(q + 1, kEmptySlot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st += T(__builtin_bit_cast(double, b0));
            sp += T(__builtin_bit_cast(double, b1));
        } else {
            st += T(part[2 * t + 0]);
            sp += T(part[2 * t + 1]);
        }
    }
    if constexpr (POLL) {
        if (lost && status) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    sumT = wave_sum(st);
    sumP = wave_sum(sp);
}

// One tile and, on the fused path, the finalization of its trajectory when this wave arrived last.
template <typename T, int MISSION, int WIND, int VEC, int PAT, bool NT, int NP>
__device__ __forceinline__ void run_tile(const FgArgs &a, T *lds, int item, int lane)
{
    const TileAt at = tile_at(a, item);
    const int b = at.b;
    T sumT, sumP;
    Publish pub{nullptr, nullptr, 0u};
    if (a.fused) {
        pub.slot = a.partial + 2 * (long)item;
        pub.counter = a.counter + b;
    }
    // The SNOPT callback (a.done) reads x where the caller keeps it, in host memory: every tile wave asks for the 23 values
    // finalize_body reads (dt, node 0, node N) right away, so that whichever wave finalizes has them without a second
    // PCIe round trip at the end (fg_single_kernel's wave 0 does the same).
    T pre = T(0);
    if (a.done && lane < 23) {
        const T *x = static_cast<const T *>(a.X) + (long)b * a.ldx;
        pre = x[lane <= 11 ? lane : NI * a.N + lane - 11];
    }
    tile_body<T, MISSION, WIND, VEC, PAT, NT, NP>(a, lds, item, lane, sumT, sumP, pub);
    if (a.fused) {
        const unsigned old = __builtin_amdgcn_readfirstlane(pub.old);
        if (old == (unsigned)(at.tiles - 1)) {         // wave-uniform: every tile of b has arrived
            if (lane == 0) __hip_atomic_store(pub.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            T st = T(0), sp = T(0);
            if (a.needF) sum_partials<T, true>(a.partial + 2 * (long)at.first, at.tiles, lane, st, sp, a.status);
            const T *edge = nullptr;
            if (a.done) {                               // the wave's LDS rows have been streamed: the space is free
                if (lane < 23) lds[lane] = pre;
                __syncthreads();
                edge = lds;
            }
            finalize_body<T, MISSION, PAT>(a, b, lane, st, sp, edge);
        }
    } else if (a.needF && lane == 0) {
        a.partial[2 * (long)item + 0] = (double)sumT;
        a.partial[2 * (long)item + 1] = (double)sumP;
    }
}

// fg_kernel: one 64-lane workgroup per tile.  Workgroup ids go round-robin over the 8 XCDs; with
// a.xcd_chunk > 0 workgroup id works on tile (id % 8) * xcd_chunk + id / 8, so that every XCD walks
// its own contiguous eighth of the batch's memory (measured +6 % on the write stream,
// profiles/r02_write_shapes.md); otherwise consecutive workgroups walk the memory in order.
// Fused form (a.fused): the wave that arrives last at its trajectory's counter also finalizes it, so
// an evaluation is one launch; otherwise finalize_kernel follows.
// (A persistent form with per-XCD tile queues was measured 12-25 % slower: profiles/r02_persistent.md keeps the patch.)
// Register budget.  The fp64 reference-pattern kernels with the wind models the reference itself uses (none, shear) are held to
// 128 VGPRs = 4 waves per SIMD = 16 per CU: with their rows going through LDS in two passes (FgArgs::sub_nodes) that many fit
// the LDS too, and the launches in the cache are bound by exactly that residency (they compile to 133-135 registers otherwise, to
// 128 without spilling when asked).  The table / grid wind and compact-pattern variants would spill (56-190 bytes of scratch) and
// keep the compiler's own allocation; the fp32 kernels are far below the limit anyway.
template <typename T, int WIND, int PAT> constexpr int min_waves_per_simd()
{
    return two_pass_family<T, WIND, PAT>() ? 4 : TOLFG_MIN_WAVES_PER_SIMD;
}

template <typename T, int MISSION, int WIND, int VEC, int PAT, bool NT, int NP>
__global__ __launch_bounds__(TILE, (min_waves_per_simd<T, WIND, PAT>())) void fg_kernel(const FgArgs a)
{
    // dynamic LDS: (nt + 1) * RS elements are used; the launch may request more to cap the waves per CU
    // (fewer concurrent store streams suit the HBM write path better)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T *lds = reinterpret_cast<T *>(lds_raw);
    const int lane = threadIdx.x;
    int item = blockIdx.x;
    if (item < 8 * a.xcd_chunk) {                  // the rest (the launch's tail) in id order
        int x = item & 7;
        if (TOLFG_VARIANT(a) & (1 << 16)) x ^= 1;              // probes (stamped build): which XCD walks which eighth
        if (TOLFG_VARIANT(a) & (1 << 17)) x = (x + 4) & 7;
        if (TOLFG_VARIANT(a) & (1 << 18)) x = (x + 2) & 7;
        item = x * a.xcd_chunk + (item >> 3);
    }
    if (item >= total_tiles(a)) return;            // the grid may be rounded up to 8 * xcd_chunk
    if (a.stagger) {
        // A launch whose waves are all resident from the start (outputs within the cache, no cap) would run them in
        // step: every wave loads, then every wave computes -- four per SIMD, interleaved, so that none is done before
        // all are -- and only then does the first store leave.  Issue priority by the wave's slot on its SIMD lets one
        // wave per SIMD run ahead of the next, so the write path starts early and the phases overlap.
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        switch (hw & 3u) {                         // WAVE_ID, the slot on the SIMD
        case 0: __builtin_amdgcn_s_setprio(3); break;
        case 1: __builtin_amdgcn_s_setprio(2); break;
        case 2: __builtin_amdgcn_s_setprio(1); break;
        default: break;
        }
    }
    run_tile<T, MISSION, WIND, VEC, PAT, NT, NP>(a, lds, item, lane);
    if (a.done) signal_done(a, (int)total_tiles(a), lane == 0);
}

// Measurement aid (FgArgs::store_shape, tolfg_batch_set_store_shape): fg_kernel's launch -- the same grid, tile order,
// LDS request (resident-wave cap) and store flavour -- with nothing in it but the slab stream's stores: every wave
// writes the whole 16-byte vectors of its tile's slab region with a constant.  No loads, no arithmetic, no LDS
// traffic: what the write path gives THIS stream shape on THIS box, measured in the same process as the evaluation
// (bench.py: roofline.box_stream_shape_GBs).  G holds garbage afterwards.
template <typename T, int PAT, bool NT>
__global__ __launch_bounds__(TILE) void store_shape_kernel(const FgArgs a, int mission)
{
    typedef StreamGeom<(int)sizeof(T), PAT> Gm;
    typedef typename Vec<T, Gm::GV>::type vec;
    const int lane = threadIdx.x;
    int item = blockIdx.x;
    if (item < 8 * a.xcd_chunk) item = (item & 7) * a.xcd_chunk + (item >> 3);
    if (item >= total_tiles(a)) return;
    const TileAt at = tile_at(a, item);
    const int cnt = min(at.nt, a.N - at.k0);
    const int ms = mission == MISSION_MIXED ? __builtin_amdgcn_readfirstlane(a.traj[at.b].mission) : mission;
    T *g = static_cast<T *>(a.G) + (long)at.b * a.ldg + a.c0[ms] + (long)Gm::SLABN * at.k0;
    const int shift = (int)((reinterpret_cast<unsigned long long>(g) / sizeof(T)) % Gm::GV);
    const int qhi = (Gm::SLABN * cnt + shift) / Gm::GV;
    vec *gp = reinterpret_cast<vec *>(g - shift);
    vec v;
#pragma unroll
    for (int i = 0; i < Gm::GV; i++) v[i] = T(1 + lane);
    for (int q = lane + (shift ? TILE : 0); q < qhi; q += TILE) stream_store<NT>(gp + q, v);
    if (shift && lane > 0 && lane < qhi) stream_store<NT>(gp + lane, v);
}

template <typename T, int MISSION, int PAT>
__global__ __launch_bounds__(TILE) void finalize_kernel(const FgArgs a)
{
    const int b = blockIdx.x;
    T sumT = T(0), sumP = T(0);
    if (a.needF) sum_partials<T, false>(a.partial + 2 * first_tile_of(a, b), b < a.B - a.tail_count ? a.tiles : a.tail_tiles, threadIdx.x, sumT, sumP);
    finalize_body<T, MISSION, PAT>(a, b, threadIdx.x, sumT, sumP);
}

// Whole trajectories in one workgroup (ts <= 512): wave w evaluates tile w, the tiles' objective
// terms meet in LDS, wave 0 finalizes.  One launch instead of two and no workspace round trip --
// what the latency-bound SNOPT callback wants; the batched path keeps fg_kernel + finalize_kernel
// because waves that start together stay in phase and stream worse.
template <typename T, int MISSION, int WIND, int VEC, int PAT>
__global__ __launch_bounds__(8 * TILE) void fg_single_kernel(const FgArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T *lds = reinterpret_cast<T *>(lds_raw);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x / TILE), lane = threadIdx.x % TILE;   // w is wave-uniform: keep it scalar
    const int b = blockIdx.x;
    double *red = reinterpret_cast<double *>(lds + (long)a.tiles * WAVE_LDS);     // [tiles][2]
    T *edge = reinterpret_cast<T *>(red + 2 * a.tiles);                            // [24] dt, node 0, node N
    // wave 0 finalizes: it asks for the values finalize_body reads right away, so that they are back
    // (x may live in host memory) long before they are needed
    T pre = T(0);
    if (w == 0 && lane < 23) {
        const T *x = static_cast<const T *>(a.X) + (long)b * a.ldx;
        pre = x[lane <= 11 ? lane : NI * a.N + lane - 11];
    }
    T sumT, sumP;
    Publish pub{nullptr, nullptr, 0u};
    tile_body<T, MISSION, WIND, VEC, PAT, false, 1>(a, lds + (long)w * WAVE_LDS, b * a.tiles + w, lane, sumT, sumP, pub);
    if (lane == 0) { red[2 * w] = (double)sumT; red[2 * w + 1] = (double)sumP; }
    if (w == 0 && lane < 23) edge[lane] = pre;
    __syncthreads();
    if (w == 0) {
        T st = T(0), sp = T(0);
        if (a.needF) sum_partials<T, false>(red, a.tiles, lane, st, sp);
        finalize_body<T, MISSION, PAT>(a, b, lane, st, sp, edge);
    }
    if (a.done) {
        // every wave: its stores are visible to the host (system-scope release: acknowledged AND written
        // back -- an acknowledgement alone was measured not to be enough) before the leader reports
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __syncthreads();
        signal_done<false>(a, a.B, threadIdx.x == 0);
    }
}

template <typename T>
__global__ void objectives_kernel(const T *F, long ldf, T *obj, int B)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < B) obj[t] = F[(long)t * ldf];
}

// sum of B contiguous values in double, one workgroup (the per-device partial sum of a Monte-Carlo mean; B is a
// few thousand at most, so a single 256-lane workgroup walking the vector is latency-, not bandwidth-bound)
template <typename T>
__global__ __launch_bounds__(256) void sum_kernel(const T *v, int B, double *out)
{
    __shared__ double part[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < B; i += 256) s += (double)v[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = part[0] + part[1] + part[2] + part[3];
}

template <typename T, int MISSION, int WIND, int PAT, int NP>
hipError_t launch_vec(const FgArgs &a, int vec, dim3 grid, hipStream_t s, hipEvent_t t0, hipEvent_t t1)
{
    constexpr int VMAX = 16 / sizeof(T);
    hipError_t e;
    if (a.single) {
        // one workgroup per trajectory, one launch (the callback path)
        if constexpr (MISSION == MISSION_MIXED || NP != 1) {
            return hipErrorInvalidValue;           // a mixed batch always takes the tile-per-workgroup path
        } else {
            const unsigned ldsz = (unsigned)(a.tiles * WAVE_LDS * sizeof(T) + 16 * a.tiles + 24 * sizeof(T));
            const dim3 g1(a.B), b1(TILE * a.tiles);
            if (t0 && (e = hipEventRecord(t0, s)) != hipSuccess) return e;
            if (vec == VMAX) {
                auto kf = fg_single_kernel<T, MISSION, WIND, VMAX, PAT>;
                if (ldsz > 64 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, ldsz)) != hipSuccess) return e;
                hipLaunchKernelGGL(kf, g1, b1, ldsz, s, a);
            } else {
                auto kf = fg_single_kernel<T, MISSION, WIND, 1, PAT>;
                if (ldsz > 64 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, ldsz)) != hipSuccess) return e;
                hipLaunchKernelGGL(kf, g1, b1, ldsz, s, a);
            }
            if ((e = hipGetLastError()) != hipSuccess) return e;
            return t1 ? hipEventRecord(t1, s) : hipSuccess;
        }
    }
    // Timing events ride on the dispatches themselves (hipExtLaunchKernelGGL: the kernel's own start / end
    // timestamps, no extra commands on the stream): t0 = start of fg_kernel, t1 = end of the evaluation's
    // last kernel (fg_kernel when fused, else finalize_kernel).
    const unsigned lds = (unsigned)fg_lds_request(sizeof(T) == 8 ? 0 : 1, a.waves_per_cu, a.nt, two_pass_family<T, WIND, PAT>() ? a.sub_nodes : 0);
    hipEvent_t fg_end = a.fused ? t1 : nullptr;
    auto go = [&](auto kernel, dim3 g, unsigned ldsz, hipEvent_t st, hipEvent_t en) {
        if (st || en) hipExtLaunchKernelGGL(kernel, g, dim3(TILE), ldsz, s, st, en, 0, a);
        else hipLaunchKernelGGL(kernel, g, dim3(TILE), ldsz, s, a);
    };
    if (a.store_shape) {              // measurement aid: the launch's shape with only the slab stores in it
        if (a.nt_stores) hipLaunchKernelGGL((store_shape_kernel<T, PAT, true>), grid, dim3(TILE), lds, s, a, (int)MISSION);
        else             hipLaunchKernelGGL((store_shape_kernel<T, PAT, false>), grid, dim3(TILE), lds, s, a, (int)MISSION);
        return hipGetLastError();
    }
    if (vec == VMAX) {
        if (a.nt_stores) go(fg_kernel<T, MISSION, WIND, VMAX, PAT, true, NP>, grid, lds, t0, fg_end);
        else             go(fg_kernel<T, MISSION, WIND, VMAX, PAT, false, NP>, grid, lds, t0, fg_end);
    } else if constexpr (NP == 1) {   // rows of X or F off 16-byte boundaries: scalar window loads and defect stores (plain slab stores)
        go(fg_kernel<T, MISSION, WIND, 1, PAT, false, 1>, grid, lds, t0, fg_end);
    } else {
        return hipErrorInvalidValue;  // two nodes per lane: aligned rows only (the host plans 64-node tiles otherwise)
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (!a.fused) {
        go(finalize_kernel<T, MISSION, PAT>, dim3(a.B), 0u, nullptr, t1);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipSuccess;
}

template <typename T, int MISSION, int PAT, int NP>
hipError_t launch_wind(const FgArgs &a, int wind, int vec, dim3 grid, hipStream_t s, hipEvent_t t0, hipEvent_t t1)
{
    switch (wind) {
    case WIND_NONE:  return launch_vec<T, MISSION, WIND_NONE, PAT, NP>(a, vec, grid, s, t0, t1);
    case WIND_SHEAR: return launch_vec<T, MISSION, WIND_SHEAR, PAT, NP>(a, vec, grid, s, t0, t1);
    case WIND_TABLE: return launch_vec<T, MISSION, WIND_TABLE, PAT, NP>(a, vec, grid, s, t0, t1);
    case WIND_GRID:  return launch_vec<T, MISSION, WIND_GRID, PAT, NP>(a, vec, grid, s, t0, t1);
    }
    return hipErrorInvalidValue;
}

template <typename T, int PAT, int NP>
hipError_t launch_mission(const FgArgs &a, int mission, int wind, int vec, dim3 grid, hipStream_t s, hipEvent_t t0,
                          hipEvent_t t1)
{
    switch (mission) {
    case MISSION_S10:   return launch_wind<T, MISSION_S10, PAT, NP>(a, wind, vec, grid, s, t0, t1);
    case MISSION_G7:    return launch_wind<T, MISSION_G7, PAT, NP>(a, wind, vec, grid, s, t0, t1);
    case MISSION_MIXED: return launch_wind<T, MISSION_MIXED, PAT, NP>(a, wind, vec, grid, s, t0, t1);
    }
    return hipErrorInvalidValue;
}


// ---- set-up kernels: initial guess and bounds generated on the device.
// ref: problemS10::InitialCond src/problemS10.cpp:19-219, problemG7::InitialCond src/problemG7.cpp:19-217 (restated in
// setup.cpp::initial_guess, whose operations these kernels repeat one for one).
__device__ __forceinline__ double atan2_t(double y, double x) { return atan2(y, x); }

// Everything InitialCond computes at ONE node, in two parts.  x0_shape: what depends on the mission and the node's time
// alone -- the flown curve's sine and cosine, air-relative speed, the course before chi_d is added and before it is made
// continuous, flight-path angle, the load-factor magnitude, bank, and the specific force along the velocity: the same
// for every trajectory of a mission, so the node-parallel form computes it ONCE per (mission, node) into a table
// (x0_table_kernel).  x0_finish: what the trajectory adds -- start offset, chi_d, the aircraft (lift coefficient, drag,
// thrust).  x0_node = finish(shape) is what the serial form evaluates per node.
struct X0Shape { double s, c, Va, chi0, gam, nmag, phi, dot; };
struct X0Node { double p[3], Va, chi_raw, gam, phi, CL, thrust; };

// FAST: atan2(+-0, x > 0) is +-0 by definition, and both missions' guesses have such arguments at every node (no vertical
// motion: az = 0; G7 flies a straight line: ay = 0) -- the node-parallel form skips the library call there; the serial
// reference form always calls it, and the two are compared bit for bit.
template <bool FAST> __device__ __forceinline__ double x0_atan2(double y, double x)
{
    if (FAST && y == 0.0 && x > 0.0) return y;
    return atan2_t(y, x);
}

template <bool FAST>
__device__ __forceinline__ X0Shape x0_shape(bool loiter, double t)
{
#pragma clang fp contract(off)      // every operation rounded on its own, as the host's build of InitialCond does: the serial and the node-parallel form then agree by construction
    constexpr double kPi = 3.14159265358979323846;
    const double tfinal = loiter ? 20.0 : 10.0;
    const double ax = loiter ? 100.0 : 40.0, ay = loiter ? 100.0 : 0.0, az = 0.0;
    const double w = 2.0 * kPi / tfinal;
    X0Shape o;
    o.s = sin(w * t); o.c = cos(w * t);     // (sincos() shares the reduction but differs from these in the last bit at some nodes)
    const double s = o.s, c = o.c;
    double v[3], acc[3];
    if (loiter) {
        v[0] = w * ax * c;              v[1] = w * ay * s;             v[2] = -w * az * s;
        acc[0] = -w * w * ax * s;       acc[1] = w * w * ay * c;       acc[2] = -w * w * az * c;
    } else {
        v[0] = ax / tfinal;             v[1] = ay * w * s;             v[2] = -az * w * s;
        acc[0] = 0.0;                   acc[1] = ay * w * w * c;       acc[2] = -az * w * w * c;
    }
    o.Va = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    o.chi0 = x0_atan2<FAST>(v[1], v[0]);
    const double h2 = v[0] * v[0] + v[1] * v[1];
    o.gam = (FAST && v[2] == 0.0 && h2 > 0.0) ? -v[2] : atan2_t(-v[2], sqrt(h2));       // atan2(+-0, sqrt(h2) > 0) = +-0
    const double u[3] = {v[0] / o.Va, v[1] / o.Va, v[2] / o.Va};
    const double sf[3] = {acc[0], acc[1], acc[2] - kGrav};
    const double n0 = -sf[0] * (u[0] * u[0] - 1.0) - u[0] * u[1] * sf[1] - u[0] * u[2] * sf[2];
    const double n1 = -sf[1] * (u[1] * u[1] - 1.0) - u[0] * u[1] * sf[0] - u[1] * u[2] * sf[2];
    const double n2 = -sf[2] * (u[2] * u[2] - 1.0) - u[0] * u[2] * sf[0] - u[1] * u[2] * sf[1];
    o.nmag = sqrt(n0 * n0 + n1 * n1 + n2 * n2);
    const double l0 = -n0 / o.nmag, l1 = -n1 / o.nmag, l2 = -n2 / o.nmag;
    o.phi = atan2_t(l0 * u[1] - l1 * u[0], l2);
    o.dot = u[0] * sf[0] + u[1] * sf[1] + u[2] * sf[2];
    return o;
}

__device__ __forceinline__ X0Node x0_finish(bool loiter, double t, const X0Shape &sh, const TrajDev &tr, const AcCoef &ac, double cd, double sd)
{
#pragma clang fp contract(off)
    constexpr double kRho = 1.2682, kPi = 3.14159265358979323846;
    const double tfinal = loiter ? 20.0 : 10.0;
    const double ax = loiter ? 100.0 : 40.0, ay = loiter ? 100.0 : 0.0, az = 0.0;
    const double s = sh.s, c = sh.c;
    X0Node o;
    if (loiter) {
        o.p[0] = ax * s - ax + tr.xi;   o.p[1] = -ay * c + tr.yi;      o.p[2] = az * c - az + tr.zi;
    } else {
        const double px = ax / tfinal * t + tr.xi, py = -ay * c + ay + tr.yi;
        o.p[0] = cd * px - sd * py;
        o.p[1] = sd * px + cd * py;
        o.p[2] = az * c - az + tr.zi;
    }
    o.Va = sh.Va;
    o.chi_raw = sh.chi0 + (loiter ? 0.0 : tr.chi_d);
    o.gam = sh.gam;
    o.phi = sh.phi;
    o.CL = 2.0 * (ac.mm * sh.nmag) / (kRho * o.Va * o.Va * ac.SS);
    const double drag = 0.5 * kRho * o.Va * o.Va * ac.SS * (ac.Cd0 + o.CL * o.CL / (kPi * ac.AR * ac.ee));
    o.thrust = ac.mm * sh.dot + drag;
    return o;
}

template <bool FAST>
__device__ __forceinline__ X0Node x0_node(bool loiter, double t, const TrajDev &tr, const AcCoef &ac, double cd, double sd)
{
    return x0_finish(loiter, t, x0_shape<FAST>(loiter, t), tr, ac, cd, sd);
}

// The one step of InitialCond that looks at the node before: the course is kept continuous from node to node.
__device__ __forceinline__ double x0_unwrap(double chi, double chi_prev)
{
#pragma clang fp contract(off)
    constexpr double kPi = 3.14159265358979323846;
    double jump = chi - chi_prev;
    while (jump < -kPi || jump > kPi) {
        if (jump < -kPi) chi = chi + 2.0 * kPi * ceil((-kPi - jump) / (2.0 * kPi));
        if (jump > kPi) chi = chi + 2.0 * kPi * floor((kPi - jump) / (2.0 * kPi));
        jump = chi - chi_prev;
    }
    return chi;
}

// Serial form: one thread per trajectory walks its nodes in order, exactly like the host code.  Kept as the reference the
// node-parallel kernel is checked against bit for bit (TOLFG_X0_SERIAL=1, tests/test_device_setup.py); 353 us for 8192
// trajectories of 201 nodes, longer than the evaluation it feeds.
template <typename T, int MISSION>
__global__ void x0_serial_kernel(const FgArgs a)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const int N = a.N;
    const TrajDev tr = a.traj[b];
    const bool loiter = (MISSION == MISSION_MIXED ? tr.mission : MISSION) == MISSION_S10;
    const AcCoef ac = a.ac[tr.ac];
    T *x = static_cast<T *>(const_cast<void *>(a.X)) + (long)b * a.ldx;
    const double tfinal = loiter ? 20.0 : 10.0;
    const double dt = tfinal / N;
    const double cd = tr.cchi, sd = tr.schi;       // cos / sin of chi_d as the host formed them (setup.cpp::initial_guess uses the same)
    double t = 0.0, chi_prev = 0.0, phi_prev = 0.0, CL_prev = 0.0, dphi_last = 0.0, dCL_last = 0.0;
    for (int k = 0; k <= N; ++k, t = t + dt) {
        const X0Node o = x0_node<false>(loiter, t, tr, ac, cd, sd);
        const double chi = k > 0 ? x0_unwrap(o.chi_raw, chi_prev) : o.chi_raw;
        T *nd = x + 11 * k + 1;
        dphi_last = k ? (o.phi - phi_prev) / dt : 0.0;
        dCL_last = k ? (o.CL - CL_prev) / dt : 0.0;
        nd[0] = T(o.p[0]); nd[1] = T(o.p[1]); nd[2] = T(o.p[2]);
        nd[3] = T(o.Va); nd[4] = T(o.gam); nd[5] = T(chi); nd[6] = T(o.phi); nd[7] = T(o.CL);
        nd[8] = T(dphi_last); nd[9] = T(dCL_last); nd[10] = T(o.thrust);
        chi_prev = chi; phi_prev = o.phi; CL_prev = o.CL;
    }
    x[0] = T(dt);
    if (loiter) {   // src/problemS10.cpp:210-211
        x[9] = T(dphi_last);
        x[10] = T(dCL_last);
    }
}

// Node-parallel form: one workgroup per trajectory, one thread per node, X0_NODES nodes per pass.  What a node's row
// holds beyond the trajectory's own start offset, chi_d and aircraft is the same for every trajectory of a mission:
// x0_table_kernel computes it once per (mission, node) -- with the library's sin, cos, atan2, the square roots and
// divisions of x0_shape -- and x0_kernel reads it back (X0_FIELDS doubles per node, from L2), which leaves three
// divisions per node and makes the launch a writer of rows.  Only three things in InitialCond look at the node before,
// and each has a parallel form whose result is the serial one bit for bit:
//   * the node's time is the running sum t = t + dt: the host adds it up once per mission (the same IEEE additions)
//     and the kernels read t_k from that table (field 0);
//   * the course is made continuous against the PREVIOUS node's continuous course: every node first takes a guess --
//     its own course plus 2 pi times the number of wraps before it (a scan over the nodes of the jumps of the raw
//     course) -- and then runs the reference's own step (x0_unwrap) against its neighbour's guess; where every node
//     reproduces its guess, the guesses ARE the serial result (induction from node 0), otherwise -- never seen -- one
//     thread redoes the pass serially;
//   * the control rates are differences to the previous node's bank (table: both nodes' banks are mission data) and
//     lift coefficient (a neighbour read in LDS).
// The rows leave through an LDS image of the pass (node-major, as in memory) with coalesced stores.
constexpr int X0_NODES = 256;
// table fields: [mission][field][node], field 0 = t (from the host)
enum { X0F_T = 0, X0F_S, X0F_C, X0F_VA, X0F_CHI0, X0F_GAM, X0F_NMAG, X0F_PHI, X0F_DOT, X0F_DPHI };
static_assert(X0F_DPHI + 1 == X0_FIELDS, "x0 table fields");

template <int MISSION>
__global__ __launch_bounds__(X0_NODES) void x0_table_kernel(double *tab, int N)
{
    const int m = MISSION == MISSION_MIXED ? (int)blockIdx.y : (MISSION == MISSION_S10 ? 0 : 1);
    const bool loiter = (MISSION == MISSION_MIXED ? m == 0 : MISSION == MISSION_S10);
    const int k = blockIdx.x * X0_NODES + threadIdx.x;
    if (k > N) return;
    const long ld = N + 1;
    double *row = tab + (long)m * X0_FIELDS * ld;
    const double dt = (loiter ? 20.0 : 10.0) / N;
    const X0Shape sh = x0_shape<true>(loiter, row[X0F_T * ld + k]);
    double dphi = 0.0;
    if (k > 0) dphi = (sh.phi - x0_shape<true>(loiter, row[X0F_T * ld + k - 1]).phi) / dt;    // the neighbour's bank again: a one-off table
    row[X0F_S * ld + k] = sh.s;        row[X0F_C * ld + k] = sh.c;       row[X0F_VA * ld + k] = sh.Va;
    row[X0F_CHI0 * ld + k] = sh.chi0;  row[X0F_GAM * ld + k] = sh.gam;   row[X0F_NMAG * ld + k] = sh.nmag;
    row[X0F_PHI * ld + k] = sh.phi;    row[X0F_DOT * ld + k] = sh.dot;   row[X0F_DPHI * ld + k] = dphi;
}

template <typename T, int MISSION>
__global__ __launch_bounds__(X0_NODES) void x0_kernel(const FgArgs a, const double *tab)
{
    constexpr double kPi = 3.14159265358979323846;
    __shared__ double raw_s[X0_NODES + 1], chi_s[X0_NODES + 1], CL_s[X0_NODES + 1];   // [0] = the node before the pass
    __shared__ int wraps_s[X0_NODES / TILE];
    __shared__ int redo_s;
    __shared__ T img[X0_NODES * NI];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid % TILE, wv = tid / TILE;
    const int N = a.N;
    const TrajDev tr = a.traj[b];
    const bool loiter = (MISSION == MISSION_MIXED ? tr.mission : MISSION) == MISSION_S10;
    const AcCoef ac = a.ac[tr.ac];
    T *x = static_cast<T *>(const_cast<void *>(a.X)) + (long)b * a.ldx;
    const double dt = (loiter ? 20.0 : 10.0) / N;
    const double cd = tr.cchi, sd = tr.schi;
    const long ld = N + 1;
    const double *row = tab + (loiter ? 0 : (long)X0_FIELDS * ld);
    int wraps_before = 0;                       // 2 pi multiples the node before the pass carries (every thread keeps its copy)
    if (tid == 0) { raw_s[0] = 0.0; chi_s[0] = 0.0; CL_s[0] = 0.0; redo_s = 0; x[0] = T(dt); }
    double dphi = 0.0, dCL = 0.0;
    for (int k0 = 0; k0 <= N; k0 += X0_NODES) {
        const int k = k0 + tid;
        const int cnt = min(X0_NODES, N + 1 - k0);
        const bool act = tid < cnt;
        X0Node o{};
        if (act) {
            X0Shape sh;
            sh.s = row[X0F_S * ld + k];        sh.c = row[X0F_C * ld + k];       sh.Va = row[X0F_VA * ld + k];
            sh.chi0 = row[X0F_CHI0 * ld + k];  sh.gam = row[X0F_GAM * ld + k];   sh.nmag = row[X0F_NMAG * ld + k];
            sh.phi = row[X0F_PHI * ld + k];    sh.dot = row[X0F_DOT * ld + k];
            dphi = row[X0F_DPHI * ld + k];
            o = x0_finish(loiter, row[X0F_T * ld + k], sh, tr, ac, cd, sd);
            raw_s[tid + 1] = o.chi_raw; CL_s[tid + 1] = o.CL;
        }
        __syncthreads();
        // wraps up to and including this node: +1 where the raw course falls by more than pi, -1 where it rises by more
        int d = 0;
        if (act && k > 0) {
            const double J = o.chi_raw - raw_s[tid];
            d = J < -kPi ? 1 : (J > kPi ? -1 : 0);
        }
        int m = d;
#pragma unroll
        for (int sh = 1; sh < TILE; sh <<= 1) {
            const int up = __shfl_up(m, sh, TILE);
            if (lane >= sh) m += up;
        }
        if (lane == TILE - 1) wraps_s[wv] = m;
        __syncthreads();
        for (int q = 0; q < wv; q++) m += wraps_s[q];
        int pass_wraps = 0;
        for (int q = 0; q < X0_NODES / TILE; q++) pass_wraps += wraps_s[q];
        m += wraps_before;
        double chi = o.chi_raw;
        if (m != 0) {
#pragma clang fp contract(off)
            chi = chi + 2.0 * kPi * (double)m;
        }
        if (act) chi_s[tid + 1] = chi;
        __syncthreads();
        // the reference's own step against the neighbour's guess
        if (act && k > 0 && x0_unwrap(o.chi_raw, chi_s[tid]) != chi) redo_s = 1;
        __syncthreads();
        if (redo_s) {                           // not seen in practice: the pass again, serially, by one thread
            if (tid == 0) {
                for (int j = 0; j < cnt; j++) chi_s[j + 1] = (k0 + j) > 0 ? x0_unwrap(raw_s[j + 1], chi_s[j]) : raw_s[j + 1];
            }
            __syncthreads();
            if (act) chi = chi_s[tid + 1];
        }
        if (act) {
            dCL = k ? (o.CL - CL_s[tid]) / dt : 0.0;
            T *nd = img + NI * tid;
            nd[0] = T(o.p[0]); nd[1] = T(o.p[1]); nd[2] = T(o.p[2]);
            nd[3] = T(o.Va); nd[4] = T(o.gam); nd[5] = T(chi); nd[6] = T(o.phi); nd[7] = T(o.CL);
            nd[8] = T(dphi); nd[9] = T(dCL); nd[10] = T(o.thrust);
        }
        __syncthreads();
        // the pass's NI * cnt elements start at x[1 + NI * k0]: one element (two, three in fp32) up to the next 16-byte
        // boundary, then whole vectors, then the rest
        {
            constexpr int GV = 16 / (int)sizeof(T);
            typedef typename Vec<T, GV>::type vec;
            T *dst = x + 1 + (long)NI * k0;
            const int E = NI * cnt;
            const int head = min(E, (int)((GV - (reinterpret_cast<unsigned long long>(dst) / sizeof(T)) % GV) % GV));
            const int nv = (E - head) / GV;
            if (tid < head) dst[tid] = img[tid];
            for (int i = tid; i < nv; i += X0_NODES) {
                vec v;
#pragma unroll
                for (int c = 0; c < GV; c++) v[c] = img[head + GV * i + c];
                *reinterpret_cast<vec *>(dst + head + GV * i) = v;
            }
            const int done_ = head + GV * nv;
            if (tid < E - done_) dst[done_ + tid] = img[done_ + tid];
        }
        // the last node of this pass is "the node before" of the next
        const double craw = raw_s[cnt], cchi = chi_s[cnt], cCL = CL_s[cnt];
        // a redone pass may have left the guess's wrap count behind: take it from the continuous course itself
        wraps_before = redo_s ? (int)rint((cchi - craw) / (2.0 * kPi)) : wraps_before + pass_wraps;
        __syncthreads();
        if (tid == 0) { raw_s[0] = craw; chi_s[0] = cchi; CL_s[0] = cCL; redo_s = 0; }
        if (loiter && k == N) {                 // src/problemS10.cpp:210-211: node 0's rates are overwritten by node N's
            // (the stores of node 0's row left before a barrier of this workgroup, so these land after them)
            x[9] = T(dphi);
            x[10] = T(dCL);
        }
        __syncthreads();
    }
}

// bounds: one thread per element of the x row / F row
template <typename T>
__global__ void bounds_kernel(const BoundsArgs a)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const TrajDev tr = a.traj[b];
    const int ms = a.mission == MISSION_MIXED ? tr.mission : a.mission;
    const int n = 11 * (a.N + 1) + 1, neF = 8 * a.N + 1 + (ms == MISSION_S10 ? 11 : 12);
    if (i < n) {
        double lo, up;
        if (i == 0) { lo = a.dtmin[ms]; up = a.dtmax[ms]; }
        else {
            const int k = (i - 1) / 11, m = (i - 1) % 11;
            if (k > 0) { lo = a.ac[ms][tr.ac].lo[m]; up = a.ac[ms][tr.ac].up[m]; }
            else {
                // node 0 is pinned to the start with the constants of src/problem.cpp:80-134
                constexpr double kPi = 3.14159265358979323846;
                const bool loiter = ms == MISSION_S10;
                switch (m) {
                case 0: lo = up = tr.xi; break;
                case 1: lo = up = tr.yi; break;
                case 2: lo = up = tr.zi; break;
                case 3: lo = 4.0; up = 50.0; break;
                case 4: lo = 0.0; up = 0.0; break;
                case 5: up = loiter ? 1.7453292519943296e+18 : 1e20 * kPi / 180.0; lo = -up; break;
                case 6: up = loiter ? 1.5707963267948966 : 90.0 * kPi / 180.0; lo = -up; break;
                case 7: lo = -0.5; up = 3.0; break;
                case 8: lo = -3.4906585039886591; up = 3.4906585039886591; break;
                case 9: lo = -200.0; up = 200.0; break;
                default: lo = 0.0; up = 1e20; break;
                }
            }
        }
        static_cast<T *>(a.xlow)[(long)b * a.ldx + i] = T(lo);
        static_cast<T *>(a.xupp)[(long)b * a.ldx + i] = T(up);
    }
    if (i < neF) {
        double lo = 0.0, up = 0.0;
        if (i == 0) { lo = -1e20; up = 1e20; }
        else if (ms == MISSION_G7 && i == neF - 1) lo = -1e20;     // dist <= dmax
        static_cast<T *>(a.Flow)[(long)b * a.ldf + i] = T(lo);
        static_cast<T *>(a.Fupp)[(long)b * a.ldf + i] = T(up);
    }
}

}  // namespace

// The evaluation kernels are instantiated in three translation units so that the library builds in parallel
// (csrc/Makefile: kernels.o with -DTOLFG_TU=0 holds fp64 and everything else, kernels_f32.o with -DTOLFG_TU=1 the
// fp32 evaluation kernels with one node per lane, kernels_f32p.o with -DTOLFG_TU=2 the packed fp32 kernels with two
// nodes per lane); a build without TOLFG_TU (tools/fgbench.cpp) holds all of them.
hipError_t launch_fg_f64(const FgArgs &a, int mission, int wind, int vec, dim3 grid, hipStream_t s, hipEvent_t t0, hipEvent_t t1);
hipError_t launch_fg_f32(const FgArgs &a, int mission, int wind, int vec, dim3 grid, hipStream_t s, hipEvent_t t0, hipEvent_t t1);
hipError_t launch_fg_f32p(const FgArgs &a, int mission, int wind, int vec, dim3 grid, hipStream_t s, hipEvent_t t0, hipEvent_t t1);

#if !defined(TOLFG_TU) || TOLFG_TU == 1
hipError_t launch_fg_f32(const FgArgs &a, int mission, int wind, int vec, dim3 grid, hipStream_t s, hipEvent_t t0, hipEvent_t t1)
{
    return a.pattern == PATTERN_COMPACT ? launch_mission<float, PATTERN_COMPACT, 1>(a, mission, wind, vec, grid, s, t0, t1)
                                        : launch_mission<float, PATTERN_REFERENCE, 1>(a, mission, wind, vec, grid, s, t0, t1);
}
#endif

#if !defined(TOLFG_TU) || TOLFG_TU == 2
hipError_t launch_fg_f32p(const FgArgs &a, int mission, int wind, int vec, dim3 grid, hipStream_t s, hipEvent_t t0, hipEvent_t t1)
{
    return a.pattern == PATTERN_COMPACT ? launch_mission<float, PATTERN_COMPACT, 2>(a, mission, wind, vec, grid, s, t0, t1)
                                        : launch_mission<float, PATTERN_REFERENCE, 2>(a, mission, wind, vec, grid, s, t0, t1);
}
#endif

#if !defined(TOLFG_TU) || TOLFG_TU == 0
hipError_t launch_fg_f64(const FgArgs &a, int mission, int wind, int vec, dim3 grid, hipStream_t s, hipEvent_t t0, hipEvent_t t1)
{
    return a.pattern == PATTERN_COMPACT ? launch_mission<double, PATTERN_COMPACT, 1>(a, mission, wind, vec, grid, s, t0, t1)
                                        : launch_mission<double, PATTERN_REFERENCE, 1>(a, mission, wind, vec, grid, s, t0, t1);
}

hipError_t launch_fg(const FgArgs &a, int mission, int wind, int dtype, int vec, hipStream_t s, hipEvent_t t0,
                     hipEvent_t t1)
{
    if (a.B <= 0) return hipSuccess;
    // the tiling must be one plan_tiles() can produce: nt a multiple of 4 in [4, 64], tiles = ceil(N/nt); fp32 tiles of
    // more than 64 (up to 128) nodes select the packed kernels, two nodes per lane
    const bool packed = dtype == 1 && a.nt > TILE;
    if (a.N < 1 || a.nt < 4 || a.nt > (dtype == 1 ? 2 * TILE : TILE) || (a.nt & 3) || a.tiles != (a.N + a.nt - 1) / a.nt || !a.partial)
        return hipErrorInvalidValue;
    if (packed && (a.single || vec != 4)) return hipErrorInvalidValue;
    // rows through LDS in passes: 32 nodes of slabs are whole 16-byte vectors for every element size and pattern
    if (a.sub_nodes != 0 && (a.sub_nodes != 32 || a.single || dtype != 0 || a.pattern != PATTERN_REFERENCE || (wind != WIND_NONE && wind != WIND_SHEAR)))
        return hipErrorInvalidValue;
    if ((a.fused || a.done) && !a.counter) return hipErrorInvalidValue;
    if (a.done && !a.fused && !a.single) return hipErrorInvalidValue;     // finalize_kernel would still be running
    if (a.tail_count < 0 || a.tail_count > a.B) return hipErrorInvalidValue;
    if (a.tail_count > 0 && (a.single || a.tail_nt < 4 || a.tail_nt > a.nt || (a.tail_nt & 3) ||
                              a.tail_tiles != (a.N + a.tail_nt - 1) / a.tail_nt))
        return hipErrorInvalidValue;
    const long W = (long)(a.B - a.tail_count) * a.tiles + (long)a.tail_count * a.tail_tiles;
    if (W > 0x7ffffff0L) return hipErrorInvalidValue;
    if (a.xcd_chunk < 0 || a.xcd_chunk > (int)((W + 7) / 8)) return hipErrorInvalidValue;
    const long covered = 8L * a.xcd_chunk;
    const dim3 grid((unsigned)(covered > W ? covered : W));
    if (dtype == 0) return launch_fg_f64(a, mission, wind, vec, grid, s, t0, t1);
    return packed ? launch_fg_f32p(a, mission, wind, vec, grid, s, t0, t1) : launch_fg_f32(a, mission, wind, vec, grid, s, t0, t1);
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

hipError_t launch_sum(const void *v, int B, int dtype, double *out, hipStream_t s)
{
    if (dtype == 0) hipLaunchKernelGGL(sum_kernel<double>, dim3(1), dim3(256), 0, s, static_cast<const double *>(v), B, out);
    else            hipLaunchKernelGGL(sum_kernel<float>, dim3(1), dim3(256), 0, s, static_cast<const float *>(v), B, out);
    return hipGetLastError();
}

hipError_t launch_x0_table(double *tab, int N, int mission, hipStream_t s)
{
    const dim3 block(X0_NODES), grid((N + 1 + X0_NODES - 1) / X0_NODES, mission == MISSION_MIXED ? 2 : 1);
    if (mission == MISSION_S10)     hipLaunchKernelGGL(x0_table_kernel<MISSION_S10>, grid, block, 0, s, tab, N);
    else if (mission == MISSION_G7) hipLaunchKernelGGL(x0_table_kernel<MISSION_G7>, grid, block, 0, s, tab, N);
    else                            hipLaunchKernelGGL(x0_table_kernel<MISSION_MIXED>, grid, block, 0, s, tab, N);
    return hipGetLastError();
}

hipError_t launch_x0(const FgArgs &a, int mission, int dtype, const double *tab, hipStream_t s)
{
    if (a.B <= 0) return hipSuccess;
    if (!tab) {       // the serial reference form (measurements, bitwise A/B)
        const dim3 grid((a.B + 63) / 64), block(64);
        if (dtype == 0) {
            if (mission == MISSION_S10)     hipLaunchKernelGGL((x0_serial_kernel<double, MISSION_S10>), grid, block, 0, s, a);
            else if (mission == MISSION_G7) hipLaunchKernelGGL((x0_serial_kernel<double, MISSION_G7>), grid, block, 0, s, a);
            else                            hipLaunchKernelGGL((x0_serial_kernel<double, MISSION_MIXED>), grid, block, 0, s, a);
        } else {
            if (mission == MISSION_S10)     hipLaunchKernelGGL((x0_serial_kernel<float, MISSION_S10>), grid, block, 0, s, a);
            else if (mission == MISSION_G7) hipLaunchKernelGGL((x0_serial_kernel<float, MISSION_G7>), grid, block, 0, s, a);
            else                            hipLaunchKernelGGL((x0_serial_kernel<float, MISSION_MIXED>), grid, block, 0, s, a);
        }
        return hipGetLastError();
    }
    const dim3 grid(a.B), block(X0_NODES);
    if (dtype == 0) {
        if (mission == MISSION_S10)     hipLaunchKernelGGL((x0_kernel<double, MISSION_S10>), grid, block, 0, s, a, tab);
        else if (mission == MISSION_G7) hipLaunchKernelGGL((x0_kernel<double, MISSION_G7>), grid, block, 0, s, a, tab);
        else                            hipLaunchKernelGGL((x0_kernel<double, MISSION_MIXED>), grid, block, 0, s, a, tab);
    } else {
        if (mission == MISSION_S10)     hipLaunchKernelGGL((x0_kernel<float, MISSION_S10>), grid, block, 0, s, a, tab);
        else if (mission == MISSION_G7) hipLaunchKernelGGL((x0_kernel<float, MISSION_G7>), grid, block, 0, s, a, tab);
        else                            hipLaunchKernelGGL((x0_kernel<float, MISSION_MIXED>), grid, block, 0, s, a, tab);
    }
    return hipGetLastError();
}

hipError_t launch_bounds(const BoundsArgs &a, int dtype, hipStream_t s)
{
    if (a.B <= 0) return hipSuccess;
    const int n = 11 * (a.N + 1) + 1;
    const dim3 grid((n + 255) / 256, a.B), block(256);
    if (dtype == 0) hipLaunchKernelGGL(bounds_kernel<double>, grid, block, 0, s, a);
    else            hipLaunchKernelGGL(bounds_kernel<float>, grid, block, 0, s, a);
    return hipGetLastError();
}

int fg_lds_bytes(int dtype, int nt, int sub_nodes)
{
    // a spare row and the rows of one pass, RS elements each; the x window (11*nt + 9 elements, rounded up to whole
    // 16-byte vectors) is staged in the same space first
    const int tile = nt > 0 && nt <= 2 * TILE ? nt : TILE;
    const int rows = sub_nodes > 0 && sub_nodes < tile ? sub_nodes : tile;
    const int elems = (rows + 1) * RS > NI * tile + 12 ? (rows + 1) * RS : NI * tile + 12;
    return ((elems * (dtype == 0 ? 8 : 4)) + 15) & ~15;
}

int fg_lds_request(int dtype, int waves_per_cu, int nt, int sub_nodes)
{
    // LDS per workgroup such that at most `waves_per_cu` one-wave workgroups fit the CU's 160 KiB
    const int need = fg_lds_bytes(dtype, nt, sub_nodes);
    if (waves_per_cu <= 0) return need;
    int cap = (160 * 1024 / waves_per_cu) & ~15;
    if (cap > 64 * 1024) cap = 64 * 1024;
    return cap > need ? cap : need;
}
#endif   // TOLFG_TU != 1

}  // namespace tolfg
