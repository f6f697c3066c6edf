// fgprobe.cpp -- diagnostic harness for the fg kernel (NOT part of the product).
//
// Builds tol_amd/csrc/kernels.hip with -DTOLFG_STAMPS into a standalone program that
//   * times the fused F+G launch on a synthetic S10/ts=200 batch with HIP events, and
//   * reports where a dynamics-tile wavefront spends its cycles (s_memtime stamps, medians).
// Stamp builds are for phase SHARES only; their run time is not a benchmark figure
// (cdna_hip_programming.md section 7, "In-kernel stamps").
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTOLFG_STAMPS -o gpurun_out/fgprobe tools/fgprobe.cpp
#include "../tol_amd/csrc/kernels.hip"
#include "../tol_amd/csrc/plan.cpp"

#include <algorithm>
#include <map>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 4096;
    const int N = argc > 2 ? atoi(argv[2]) : 200;
    const int reps = argc > 3 ? atoi(argv[3]) : 30;
    const int variant = argc > 4 ? atoi(argv[4]) : 0;
    const int pattern = argc > 9 ? atoi(argv[9]) : 0;        // 1 = compact pattern (46-entry slabs)
    const int n = 11 * (N + 1) + 1, neF = 8 * N + 12, neG = pattern ? (3 * N + 4) + 46 * N + 22 : 107 * N + 37;
    const long ldx = (n + 1) & ~1L, ldf = (neF + 1) & ~1L, ldg = (neG + 1) & ~1L;

    std::vector<double> X((size_t)B * ldx);
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1, 1);
    for (int b = 0; b < B; b++) {
        double *x = &X[(size_t)b * ldx];
        x[0] = 0.1;
        for (int k = 0; k <= N; k++) {
            double *s = x + 11 * k + 1;
            s[0] = 100 * U(rng); s[1] = 100 * U(rng); s[2] = -50 + 20 * U(rng); s[3] = 15 + 3 * U(rng);
            s[4] = 0.2 * U(rng); s[5] = 3.1 * U(rng); s[6] = 0.3 * U(rng); s[7] = 0.8 + 0.2 * U(rng);
            s[8] = 0.1 * U(rng); s[9] = 0.1 * U(rng); s[10] = 10 + 5 * U(rng);
        }
    }
    std::vector<tolfg::TrajDev> tr(B);
    for (int b = 0; b < B; b++) tr[b] = tolfg::TrajDev{0.24, 0.0, 400.0, 100.0, 0.0, 1.0, 0, 0};

    double *dX, *dF, *dG; tolfg::TrajDev *dT; unsigned long long *dS;
    int tiles, nt;
    tolfg::plan_tiles(N, 0, 0, &tiles, &nt);
    const int ipb = 1;
    const long W = (long)B * tiles;
    const int xcd = argc > 6 ? atoi(argv[6]) : 1, fused = argc > 7 ? atoi(argv[7]) : 1, nts = argc > 8 ? atoi(argv[8]) : 1;
    const long blocks = xcd ? 8 * ((W + 7) / 8) : W;
    double *dP; unsigned *dC;
    CK(hipMalloc(&dP, sizeof(double) * 2 * W));
    CK(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(dP), tolfg::kEmptySlotWord, 4 * (size_t)W));
    CK(hipMalloc(&dC, sizeof(unsigned) * (B + 1)));
    CK(hipMemset(dC, 0, sizeof(unsigned) * (B + 1)));
    CK(hipMalloc(&dX, sizeof(double) * X.size()));
    CK(hipMalloc(&dF, sizeof(double) * B * ldf));
    CK(hipMalloc(&dG, sizeof(double) * B * ldg));
    CK(hipMalloc(&dT, sizeof(tolfg::TrajDev) * B));
    CK(hipMalloc(&dS, sizeof(unsigned long long) * 10 * blocks));
    CK(hipMemcpy(dX, X.data(), sizeof(double) * X.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dT, tr.data(), sizeof(tolfg::TrajDev) * B, hipMemcpyHostToDevice));
    CK(hipMemset(dS, 0, sizeof(unsigned long long) * 10 * blocks));

    tolfg::FgArgs a{};
    a.X = dX; a.ldx = ldx; a.F = dF; a.ldf = ldf; a.G = dG; a.ldg = ldg; a.wind = nullptr; a.traj = dT;
    a.B = B; a.N = N; a.tiles = tiles; a.nt = nt; a.partial = dP; a.obj = nullptr; a.c0[0] = 3 * N + 4; a.c0[1] = N + 6; a.needF = 1; a.needG = 1;
    a.kT[0] = 0; a.kp[0] = 8; a.kv[0] = 0; a.kdt[0] = 1;
    a.pattern = pattern;

    a.ac[0] = tolfg::AcCoef{1.0 / 6.1228, 1.2682 * 0.6316 / (2 * 6.1228), 0.03, 1.0 / (16.4457 * M_PI * 0.9693)};
    a.stamps = dS; a.variant = variant;
    a.waves_per_cu = argc > 5 ? atoi(argv[5]) : 0;
    a.single = 0;
    a.counter = dC; a.fused = fused; a.xcd_chunk = xcd ? (int)((W + 7) / 8) : 0; a.nt_stores = nts;

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (variant & 32768) a.needF = 0;
    for (int i = 0; i < 5; i++) CK(tolfg::launch_fg(a, tolfg::MISSION_S10, tolfg::WIND_SHEAR, 0, 2, nullptr));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) CK(tolfg::launch_fg(a, tolfg::MISSION_S10, tolfg::WIND_SHEAR, 0, 2, nullptr));
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 8.0 * B * ((double)n + neF + neG);
    printf("B=%d N=%d variant=%d waves/CU cap %d: %.2f us/launch  %.1f GB/s algorithmic (stamped build, not a benchmark)\n", B, N,
           variant, a.waves_per_cu, 1e3 * ms / reps, bytes / (1e6 * ms / reps));

    std::vector<unsigned long long> S((size_t)10 * blocks);
    CK(hipMemcpy(S.data(), dS, sizeof(unsigned long long) * S.size(), hipMemcpyDeviceToHost));
    // stamps of the LAST tile a workgroup walked
    const char *names[6] = {"wait window + regs->LDS + barrier", "LDS->regs, issue next window", "node_eval, F + cost stores, rows->LDS",
                            "G slab stores (issue)", "loop exit", "drain vmcnt(0)"};
    std::vector<std::vector<double>> d(7);
    std::vector<double> clk;
    for (long id = 0; id < blocks; id++) {
        const unsigned long long *s = &S[(size_t)id * 10];
        if (!s[0] || !s[6]) continue;
        for (int i = 0; i < 6; i++) d[i].push_back((double)(s[i + 1] - s[i]));
        d[6].push_back((double)(s[6] - s[0]));
        if (s[8] > s[7]) clk.push_back((double)(s[6] - s[0]) / (double)(s[8] - s[7]) * 100.0);
    }
    auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    const double tot = med(d[6]);
    printf("in-kernel clock (d s_memtime / d s_memrealtime x 100 MHz), median over workgroups: %.0f MHz\n", med(clk));
    printf("tiles=%d nt=%d ipb=%d: %zu workgroups, median span of the last tile %.0f cycles (s_memtime ticks)\n", tiles, nt, ipb, d[6].size(), tot);
    for (int i = 0; i < 6; i++) {
        const double m = med(d[i]);
        printf("  %-34s %9.0f cycles  %5.1f %%\n", names[i], m, 100.0 * m / tot);
    }
    // timeline of the last launch from the s_memrealtime stamps (100 MHz): how long the machine takes to fill,
    // how long it stays full, how long the tail is
    std::vector<std::pair<unsigned long long, int>> ev;
    unsigned long long t0 = ~0ull, t1 = 0;
    std::vector<double> life;
    for (long id = 0; id < blocks; id++) {
        const unsigned long long *q = &S[(size_t)id * 10];
        if (!q[7] || !q[8] || q[8] < q[7]) continue;
        ev.push_back({q[7], +1}); ev.push_back({q[8], -1});
        t0 = std::min(t0, q[7]); t1 = std::max(t1, q[8]);
        life.push_back((double)(q[8] - q[7]) * 0.01);
    }
    std::sort(ev.begin(), ev.end());
    int cur = 0, peak = 0;
    for (auto &e : ev) { cur += e.second; peak = std::max(peak, cur); }
    unsigned long long first_full = 0, last_full = 0;
    cur = 0;
    for (auto &e : ev) {
        cur += e.second;
        if (cur >= 0.9 * peak) { if (!first_full) first_full = e.first; last_full = e.first; }
    }
    double area = 0; cur = 0; unsigned long long prev = t0;
    for (auto &e : ev) { area += (double)cur * (double)(e.first - prev); prev = e.first; cur += e.second; }
    {   // concurrency over time (20 samples) and per-XCD shares; slot-idle time per CU from the HW ids
        const int NS = 20;
        std::vector<int> conc(NS, 0);
        std::vector<std::vector<int>> cx(8, std::vector<int>(NS, 0));
        std::map<unsigned, std::vector<std::pair<unsigned long long, unsigned long long>>> percu;
        for (long id = 0; id < blocks; id++) {
            const unsigned long long *q = &S[(size_t)id * 10];
            if (!q[7] || !q[8] || q[8] < q[7]) continue;
            const unsigned hw = (unsigned)(q[9] & 0xffffffffu), xcc = (unsigned)(q[9] >> 32) & 0xf;
            const unsigned cu = (xcc << 16) | (hw & 0xff00);        // xcc | se, sh, cu
            percu[cu].push_back({q[7], q[8]});
            for (int k = 0; k < NS; k++) {
                const unsigned long long t = t0 + (t1 - t0) * (2 * k + 1) / (2 * NS);
                if (q[7] <= t && t < q[8]) { conc[k]++; cx[xcc & 7][k]++; }
            }
        }
        printf("resident tile waves at 20 evenly spaced moments:");
        for (int k = 0; k < NS; k++) printf(" %d", conc[k]);
        printf("\n  of which on XCC 0 / 3 / 7:");
        for (int k = 0; k < NS; k += 3) printf(" %d/%d/%d", cx[0][k], cx[3][k], cx[7][k]);
        // per CU: mean resident waves and the idle time between one wave's end and the next start in steady state
        double sum_res = 0; int ncu = 0; std::vector<double> gaps;
        for (auto &kv : percu) {
            auto &v = kv.second;
            double busy = 0;
            for (auto &w : v) busy += (double)(w.second - w.first);
            sum_res += busy / (double)(t1 - t0); ncu++;
        }
        printf("\n  distinct CUs seen %d, mean resident waves per CU %.2f\n", ncu, ncu ? sum_res / ncu : 0.0);
        // per XCD: tile waves run, when its last wave ended (us after the launch's first wave started), median wave life
        printf("  per XCC: waves / last end us / median life us:");
        for (int x = 0; x < 8; x++) {
            unsigned long long last = 0; long cnt = 0; std::vector<double> lf;
            for (long id = 0; id < blocks; id++) {
                const unsigned long long *q = &S[(size_t)id * 10];
                if (!q[7] || !q[8] || q[8] < q[7] || (int)((q[9] >> 32) & 7) != x) continue;
                cnt++; last = std::max(last, q[8]); lf.push_back((double)(q[8] - q[7]) * 0.01);
            }
            printf("  [%d] %ld / %.1f / %.2f", x, cnt, last ? (last - t0) * 0.01 : 0.0, med(lf));
        }
        printf("\n");
    }
    printf("timeline (last launch): span first wave start -> last wave end %.2f us; peak %d resident tile waves (%.1f per CU); "
           "fill to 90%% of peak %.2f us; at >= 90%% for %.2f us; tail after the last 90%% moment %.2f us; mean occupancy %.0f waves = %.0f %% of peak; "
           "median wave life %.2f us\n",
           (t1 - t0) * 0.01, peak, peak / 256.0, (first_full - t0) * 0.01, (last_full - first_full) * 0.01, (t1 - last_full) * 0.01,
           area / (double)(t1 - t0), 100.0 * area / (double)(t1 - t0) / peak, med(life));
    return 0;
}
