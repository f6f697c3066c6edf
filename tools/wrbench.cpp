// wrbench.cpp -- write-pattern microbenchmark (diagnostic, NOT part of the product).
//
// Question it answers: which shapes of concurrent 16-B/lane store streams does the MI355X memory
// system sustain beyond the 256 MiB Infinity Cache?  The fg kernel writes ~53 KB sequentially per
// wavefront from ~1800 concurrently resident waves; a plain fill writes 1 KiB per wave in launch
// order.  Patterns:
//   mode 0: every wave writes S KiB sequentially (S store instructions of 1 KiB) to segment = block id
//   mode 1: groups of GRP consecutive waves share a GRP*S KiB region and interleave KiB chunks
//   mode 2: like 0 but each store instruction is followed by `delay` x s_sleep (slow producer)
//   mode 3/4/5: XCD-contiguous / mode 0 / mode 1 with non-temporal stores;  mode 6: 1 KiB read + S KiB nt stores
//   mode 7: workgroups of GRP waves write one contiguous S KiB region together, KiB chunks interleaved over the waves
//   mode 8: like 7 but wave w writes the w-th contiguous S/GRP KiB piece of the region
//   mode 9: mode 4 after `delay` x 64 s_sleep(8) of idling (long-lived waves, short streams)
//   mode 11: long-lived waves, interleaved globally: wave j of a super-group of G (= `delay`, 0 -> all) waves writes GRP-KiB
//            chunks j, j+G, j+2G, ... of the super-group's G*S KiB region (compact in-flight address window, long streams)
//   mode 13: mode 4 with the segments dealt in a scattered order (segment = id * 7919 mod nblocks)
//   mode 14: mode 4 (S KiB per wave, in order) but every store instruction is spread over 64/GRP sub-regions of the wave's
//            region: lanes [g*GRP, (g+1)*GRP) write GRP*16 contiguous bytes of sub-region g (one instruction touches
//            64/GRP places S/(64/GRP) KiB apart instead of one contiguous KiB)
//   mode 15: mode 4's shape with the store's cache-policy bits chosen by GRP: 0 none, 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1,
//            5 nt sc0, 6 nt sc1, 7 nt sc0 sc1
//   mode 10: mode 7 where every wave first reads 1 KiB and idles `delay` x 64 s_sleep(8) (load -> compute -> store)
// LDS bytes per block (dynamic) limit the occupancy like the real kernel's 22 KB does.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/wrbench tools/wrbench.cpp
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double vec2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void wrg(vec2 *out, int S, int grp, int delay, long nblocks)
{
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long id = blockIdx.x;
    vec2 v = {1.0 + lane, 2.0};
    if (MODE == 10) {
        const vec2 *in = reinterpret_cast<const vec2 *>(out) + (nblocks * (long)S + id * grp + w) * 64;
        vec2 r = in[lane];
        for (int d = 0; d < delay * 64; d++) __builtin_amdgcn_s_sleep(8);
        v.x = r.x + r.y;
    }
    vec2 *base = out + id * S * 64;
    if (MODE == 8) {
        const int per = (S + grp - 1) / grp;
        for (int i = w * per; i < min(S, (w + 1) * per); i++) __builtin_nontemporal_store(v, &base[(long)i * 64 + lane]);
    } else {
        for (int i = w; i < S; i += grp) __builtin_nontemporal_store(v, &base[(long)i * 64 + lane]);
    }
}

template <int MODE>
__global__ __launch_bounds__(64) void wr(vec2 *out, int S, int grp, int delay, long nblocks)
{
    extern __shared__ char smem[];
    const int lane = threadIdx.x;
    const long id = blockIdx.x;
    vec2 v = {1.0 + lane, 2.0};
    if (MODE == 1) {
        const long g = id / grp, w = id % grp;
        vec2 *base = out + (g * grp * S) * 64;
        for (int i = 0; i < S; i++) base[((long)i * grp + w) * 64 + lane] = v;
    } else if (MODE == 3) {       // XCD-contiguous: workgroups id and id+8 share an XCD -> give each XCD one region
        const long seg = (id % 8) * (nblocks / 8) + id / 8;
        vec2 *base = out + seg * S * 64;
        for (int i = 0; i < S; i++) __builtin_nontemporal_store(v, &base[(long)i * 64 + lane]);
    } else if (MODE == 4) {       // mode 0 with non-temporal stores
        vec2 *base = out + id * S * 64;
        for (int i = 0; i < S; i++) __builtin_nontemporal_store(v, &base[(long)i * 64 + lane]);
    } else if (MODE == 6) {       // short-stream tile emulation: read 1 KiB, `delay` x 64 dependent fp64 FMAs, S KiB nt stores
        const vec2 *in = reinterpret_cast<const vec2 *>(out) + (nblocks * (long)S + id) * 64;
        vec2 r = in[lane];
        double acc = r.x;
        for (int d = 0; d < delay * 64; d++) acc = acc * 1.0000001 + r.y;
        v.x = acc;
        vec2 *base = out + id * S * 64;
        for (int i = 0; i < S; i++) __builtin_nontemporal_store(v, &base[(long)i * 64 + lane]);
    } else if (MODE == 9) {       // idle first, then S KiB nt stores
        for (int d = 0; d < delay * 64; d++) __builtin_amdgcn_s_sleep(8);
        vec2 *base = out + id * S * 64;
        for (int i = 0; i < S; i++) __builtin_nontemporal_store(v, &base[(long)i * 64 + lane]);
    } else if (MODE == 11) {
        const long G = delay > 0 ? delay : nblocks, g = id / G, j = id % G;
        const long Gn = min(G, nblocks - g * G);             // the last super-group may be short
        const int C = grp, chunks = S / C;
        vec2 *base = out + g * G * S * 64;
        for (int i = 0; i < chunks; i++) {
            vec2 *c = base + (j + Gn * i) * C * 64;
            for (int q = 0; q < C; q++) __builtin_nontemporal_store(v, &c[(long)q * 64 + lane]);
        }
    } else if (MODE == 14) {
        const int LG = grp, subs = 64 / LG;                  // S * 64 vec2 per region, S * 64 / subs per sub-region
        const long sub_len = (long)S * 64 / subs;
        vec2 *base = out + id * S * 64 + (lane / LG) * sub_len + lane % LG;
        for (int i = 0; i < S; i++) __builtin_nontemporal_store(v, &base[(long)i * LG]);
    } else if (MODE == 15) {
        vec2 *base = out + id * S * 64;
#define WR_POLICY(bits) for (int i = 0; i < S; i++) asm volatile("global_store_dwordx4 %0, %1, off " bits :: "v"(&base[(long)i * 64 + lane]), "v"(v) : "memory")
        switch (grp) {
        case 0: WR_POLICY(""); break;
        case 1: WR_POLICY("nt"); break;
        case 2: WR_POLICY("sc0"); break;
        case 3: WR_POLICY("sc1"); break;
        case 4: WR_POLICY("sc0 sc1"); break;
        case 5: WR_POLICY("sc0 nt"); break;
        case 6: WR_POLICY("sc1 nt"); break;
        default: WR_POLICY("sc0 sc1 nt"); break;
        }
    } else if (MODE == 13) {
        const long seg = (id * 7919L) % nblocks;
        vec2 *base = out + seg * S * 64;
        for (int i = 0; i < S; i++) __builtin_nontemporal_store(v, &base[(long)i * 64 + lane]);
    } else if (MODE == 5) {       // mode 1 (interleaved groups) with non-temporal stores
        const long g = id / grp, w = id % grp;
        vec2 *base = out + (g * grp * S) * 64;
        for (int i = 0; i < S; i++) __builtin_nontemporal_store(v, &base[((long)i * grp + w) * 64 + lane]);
    } else {
        vec2 *base = out + id * S * 64;
        for (int i = 0; i < S; i++) {
            base[(long)i * 64 + lane] = v;
            if (MODE == 2)
                for (int d = 0; d < delay; d++) __builtin_amdgcn_s_sleep(8);
        }
    }
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int S = argc > 2 ? atoi(argv[2]) : 52;
    const int lds = argc > 3 ? atoi(argv[3]) : 0;
    const int grp = argc > 4 ? atoi(argv[4]) : 8;
    const int delay = argc > 5 ? atoi(argv[5]) : 0;
    const long total = 800L << 20;
    const long nblocks = total / (1024L * S);
    vec2 *d;
    CK(hipMalloc(&d, total + (total / (S > 0 ? S : 1) + (4 << 20)) * (mode == 10 ? grp : 1)));   // modes 6, 10 read 1 KiB per wave from behind the output
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(wr<0>, dim3(nblocks), dim3(64), lds, 0, d, S, grp, delay, nblocks);
        if (mode == 1) hipLaunchKernelGGL(wr<1>, dim3(nblocks / grp * grp), dim3(64), lds, 0, d, S, grp, delay, nblocks);
        if (mode == 2) hipLaunchKernelGGL(wr<2>, dim3(nblocks), dim3(64), lds, 0, d, S, grp, delay, nblocks);
        if (mode == 3) hipLaunchKernelGGL(wr<3>, dim3(nblocks / 8 * 8), dim3(64), lds, 0, d, S, grp, delay, nblocks / 8 * 8);
        if (mode == 4) hipLaunchKernelGGL(wr<4>, dim3(nblocks), dim3(64), lds, 0, d, S, grp, delay, nblocks);
        if (mode == 6) hipLaunchKernelGGL(wr<6>, dim3(nblocks), dim3(64), lds, 0, d, S, grp, delay, nblocks);
        if (mode == 5) hipLaunchKernelGGL(wr<5>, dim3(nblocks / grp * grp), dim3(64), lds, 0, d, S, grp, delay, nblocks);
        if (mode == 9) hipLaunchKernelGGL(wr<9>, dim3(nblocks), dim3(64), lds, 0, d, S, grp, delay, nblocks);
        if (mode == 7) hipLaunchKernelGGL(wrg<7>, dim3(nblocks), dim3(64 * grp), lds, 0, d, S, grp, delay, nblocks);
        if (mode == 8) hipLaunchKernelGGL(wrg<8>, dim3(nblocks), dim3(64 * grp), lds, 0, d, S, grp, delay, nblocks);
        if (mode == 11) hipLaunchKernelGGL(wr<11>, dim3(nblocks), dim3(64), lds, 0, d, S, grp, delay, nblocks);
        if (mode == 14) hipLaunchKernelGGL(wr<14>, dim3(nblocks), dim3(64), lds, 0, d, S, grp, delay, nblocks);
        if (mode == 15) hipLaunchKernelGGL(wr<15>, dim3(nblocks), dim3(64), lds, 0, d, S, grp, delay, nblocks);
        if (mode == 13) hipLaunchKernelGGL(wr<13>, dim3(nblocks), dim3(64), lds, 0, d, S, grp, delay, nblocks);
        if (mode == 10) hipLaunchKernelGGL(wrg<10>, dim3(nblocks), dim3(64 * grp), lds, 0, d, S, grp, delay, nblocks);
    };
    for (int i = 0; i < 3; i++) launch();
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)nblocks * S * 1024.0;
    printf("mode %d S %3d KiB/wave lds %6d grp %3d delay %3d: %8.1f us  %7.1f GB/s\n", mode, S, lds, grp, delay,
           1e3 * ms / reps, bytes / (1e6 * ms / reps));
    return 0;
}
