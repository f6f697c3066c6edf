// fgbench.cpp -- native A/B harness for the batched evaluation (diagnostic, NOT part of the product).
//
// Compiles tol_amd/csrc/kernels.hip into a standalone program (no stamps: these ARE timings) and runs
// a list of launch configurations back to back on one synthetic S10 or G7 batch, so that tile size,
// waves-per-CU cap and the fused/unfused finalize can be compared inside ONE gpurun call:
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/fgbench tools/fgbench.cpp      
//   tools/bin/fgbench [reps=N] [xbuf=K] [nt=0|1] [xcd=0|1] [pat=0|1] B,N,max_nt,cap,fused[,mission[,dtype]] ...
//
// Every configuration is first checked against the reference configuration of its shape
// (max_nt 64, unfused): defects and G must agree bitwise, the objective to 1e-13 relative.
// Timed region: `reps` evaluations between two HIP events on the launch stream (whole evaluation,
// finalize included), inputs rotated over `xbuf` buffers so that X comes from HBM.
#include "../tol_amd/csrc/kernels.hip"
#include "../tol_amd/csrc/plan.cpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

namespace {

struct Shape {
    int B, N, mission, dtype;
    long n, neF, neG, ldx, ldf, ldg;
    int c0;
};

static long g_goff = 0;             // goff=: G starts this many elements into its allocation (alignment experiments)
static long g_ldg_pad = 0;          // ldgpad=: extra elements between the rows of G (row-stride experiments)
static int g_stagger = 0;           // stagger=: FgArgs::stagger
static int g_variant = 0;           // variant=: ablation switches of a -DTOLFG_ABLATE build (256 no arithmetic, 512 no defect stores,
                                    // 1024 no objective-gradient stores, 2048 no slab stream, 4096 no x window: results are then wrong; 8192 = the stream's
                                    // offset table asked for after the x window instead of before it: results unchanged)
Shape make_shape(int B, int N, int mission, int dtype)
{
    Shape s{};
    s.B = B; s.N = N; s.mission = mission; s.dtype = dtype;
    s.n = 11L * (N + 1) + 1;
    s.neF = 8L * N + 1 + (mission == tolfg::MISSION_S10 ? 11 : 12);      // mixed: rows sized for the larger mission
    s.neG = mission == tolfg::MISSION_G7 ? 105L * N + 48 : 107L * N + 37;
    s.c0 = mission == tolfg::MISSION_S10 ? 3 * N + 4 : N + 6;
    const long v = dtype == 0 ? 2 : 4;
    auto up = [&](long m) { return (m + v - 1) / v * v; };
    s.ldx = up(s.n); s.ldf = up(s.neF); s.ldg = up(s.neG) + g_ldg_pad;
    return s;
}

struct Buffers {
    Shape sh{};
    std::vector<void *> dX;
    void *dF = nullptr, *dG = nullptr, *dF2 = nullptr, *dG2 = nullptr;
    tolfg::TrajDev *dT = nullptr;
    double *dP = nullptr;
    unsigned *dC = nullptr;
    long capW = 0, poll_ready = -1;
    int nt = 1, xcd = 0, pat = 0, xcdpct = 100, tail_count = 0, tail_nt = 16;
    size_t es() const { return sh.dtype == 0 ? 8 : 4; }
    void release()
    {
        for (void *p : dX) (void)hipFree(p);
        dX.clear();
        for (void *p : {dF, dG, dF2, dG2, (void *)dT, (void *)dP, (void *)dC}) if (p) (void)hipFree(p);
        dF = dG = dF2 = dG2 = nullptr; dT = nullptr; dP = nullptr; dC = nullptr; capW = 0; poll_ready = -1;
    }
};

void fill(Buffers &bf, const Shape &sh, int xbuf)
{
    bf.release();
    bf.sh = sh;
    std::vector<double> X((size_t)sh.B * sh.ldx);
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1, 1);
    for (int b = 0; b < sh.B; b++) {
        double *x = &X[(size_t)b * sh.ldx];
        x[0] = 0.1;
        for (int k = 0; k <= sh.N; k++) {
            double *s = x + 11 * k + 1;
            s[0] = 100 * U(rng); s[1] = 100 * U(rng); s[2] = -50 + 20 * U(rng); s[3] = 15 + 3 * U(rng);
            s[4] = 0.2 * U(rng); s[5] = 3.1 * U(rng); s[6] = 0.3 * U(rng); s[7] = 0.8 + 0.2 * U(rng);
            s[8] = 0.1 * U(rng); s[9] = 0.1 * U(rng); s[10] = 10 + 5 * U(rng);
        }
    }
    std::vector<float> Xf;
    if (sh.dtype == 1) Xf.assign(X.begin(), X.end());
    const void *src = sh.dtype == 0 ? (const void *)X.data() : (const void *)Xf.data();
    const size_t xb = bf.es() * X.size();
    for (int j = 0; j < xbuf; j++) {
        void *p;
        CK(hipMalloc(&p, xb));
        // buffer j = the same rows rotated by 7*j rows (every row stays a valid input)
        const size_t rot = ((size_t)7 * j % sh.B) * sh.ldx * bf.es();
        CK(hipMemcpy(p, (const char *)src + rot, xb - rot, hipMemcpyHostToDevice));
        if (rot) CK(hipMemcpy((char *)p + (xb - rot), src, rot, hipMemcpyHostToDevice));
        bf.dX.push_back(p);
    }
    std::vector<tolfg::TrajDev> tr(sh.B);
    for (int b = 0; b < sh.B; b++) {
        tr[b] = tolfg::TrajDev{};
        tr[b].shear = 0.1 + 0.3 * ((b * 37) % 100) / 100.0;
        tr[b].xg = 0.0; tr[b].yg = 400.0; tr[b].rg = 100.0; tr[b].cchi = 0.0; tr[b].schi = 1.0; tr[b].ac = 0;
        tr[b].mission = sh.mission == tolfg::MISSION_MIXED ? (b & 1) : sh.mission;
    }
    CK(hipMalloc(&bf.dF, bf.es() * sh.B * sh.ldf));
    CK(hipMalloc(&bf.dG, bf.es() * (sh.B * sh.ldg + 64)));
    CK(hipMalloc(&bf.dF2, bf.es() * sh.B * sh.ldf));
    CK(hipMalloc(&bf.dG2, bf.es() * (sh.B * sh.ldg + 64)));
    CK(hipMemset(bf.dF2, 0xff, bf.es() * sh.B * sh.ldf));      // rows of a mixed batch leave their tails untouched
    CK(hipMemset(bf.dG2, 0xff, bf.es() * sh.B * sh.ldg));
    CK(hipMalloc(&bf.dT, sizeof(tolfg::TrajDev) * sh.B));
    CK(hipMalloc(&bf.dC, sizeof(unsigned) * (sh.B + 1)));
    CK(hipMemset(bf.dC, 0, sizeof(unsigned) * (sh.B + 1)));
    CK(hipMemcpy(bf.dT, tr.data(), sizeof(tolfg::TrajDev) * sh.B, hipMemcpyHostToDevice));
}

tolfg::FgArgs make_args(Buffers &bf, int max_nt, int cap, int fused, int xi, void *F, void *G)
{
    const Shape &sh = bf.sh;
    tolfg::FgArgs a{};
    a.X = bf.dX[xi % bf.dX.size()]; a.ldx = sh.ldx; a.F = F; a.ldf = sh.ldf; a.G = static_cast<char *>(G) + g_goff * (sh.dtype == 0 ? 8 : 4); a.ldg = sh.ldg;
    a.wind = nullptr; a.traj = bf.dT; a.B = sh.B; a.N = sh.N; a.c0[0] = 3 * sh.N + 4; a.c0[1] = sh.N + 6;
    tolfg::plan_tiles(sh.N, sh.dtype, max_nt, &a.tiles, &a.nt);
    if (bf.tail_count > 0) {                 // finer tiles for the trajectories reached last
        a.tail_count = bf.tail_count < sh.B ? bf.tail_count : sh.B;
        tolfg::plan_tiles(sh.N, sh.dtype, bf.tail_nt < a.nt ? bf.tail_nt : a.nt, &a.tail_tiles, &a.tail_nt);
    }
    const long body = (long)(sh.B - a.tail_count) * a.tiles;
    const long W = body + (long)a.tail_count * a.tail_tiles;
    if (W > bf.capW) {
        if (bf.dP) CK(hipFree(bf.dP));
        CK(hipMalloc(&bf.dP, sizeof(double) * 2 * W));
        bf.capW = W;
    }
    a.partial = bf.dP; a.counter = bf.dC; a.fused = fused; a.single = 0; a.obj = nullptr;
    a.nt_stores = bf.nt; a.xcd_chunk = bf.xcd ? (int)(((a.tail_count ? body : W) * bf.xcdpct / 100) / 8) : 0;
    if (fused && bf.poll_ready != W) {   // slots start empty; the unfused path leaves values behind
        CK(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(bf.dP), tolfg::kEmptySlotWord, 4 * (size_t)W));
        CK(hipDeviceSynchronize());
        bf.poll_ready = W;
    } else if (!fused) {
        bf.poll_ready = -1;
    }
    a.needF = 1; a.needG = 1; a.pattern = bf.pat; a.waves_per_cu = cap; a.stagger = g_stagger;
#ifdef TOLFG_ABLATE
    a.variant = g_variant;
#endif
    if (bf.pat == tolfg::PATTERN_COMPACT) { a.c0[0] = 3 * sh.N + 4; a.c0[1] = sh.N + 6; }
    a.kT[0] = 0.3; a.kp[0] = 8; a.kv[0] = 0; a.kdt[0] = 1;
    a.kT[1] = 100; a.kp[1] = 0.7; a.kv[1] = 0.4; a.kdt[1] = 0;
    a.ac[0] = tolfg::AcCoef{1.0 / 6.1228, 1.2682 * 0.6316 / (2 * 6.1228), 0.03, 1.0 / (16.4457 * M_PI * 0.9693), 6.1228, 0.6316, 16.4457, 0.9693};
    return a;
}

long compare(const Buffers &bf, bool approx = false)
{
    const Shape &sh = bf.sh;
    const size_t es = bf.es();
    std::vector<char> F1(es * sh.B * sh.ldf), F2(F1.size()), G1(es * sh.B * sh.ldg), G2(G1.size());
    CK(hipMemcpy(F1.data(), bf.dF, F1.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(F2.data(), bf.dF2, F2.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(G1.data(), bf.dG, G1.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(G2.data(), bf.dG2, G2.size(), hipMemcpyDeviceToHost));
    long bad = 0;
    for (int b = 0; b < sh.B; b++) {
        const char *f1 = &F1[es * b * sh.ldf], *f2 = &F2[es * b * sh.ldf];
        double o1, o2;
        if (es == 8) { o1 = *(const double *)f1; o2 = *(const double *)f2; }
        else { o1 = *(const float *)f1; o2 = *(const float *)f2; }
        if (!(std::fabs(o1 - o2) <= (es == 8 ? 1e-13 : 2e-6) * (1 + std::fabs(o1)))) {
            static int shown_obj = 0;
            if (shown_obj++ < 4) fprintf(stderr, "  objective mismatch b=%d: %.17g vs %.17g\n", b, o1, o2);
            bad++;
        }
        if (approx) {      // packed fp32 kernels: own sin/cos and reciprocals, so a tolerance instead of bit equality
            static int shown = 0;
            auto close = [&](const char *what, const char *p, const char *q, long n) {
                for (long i = 0; i < n; i++) {
                    const float u = ((const float *)p)[i], v = ((const float *)q)[i];
                    if (std::memcmp(&u, &v, 4) != 0 && !(std::fabs(u - v) <= 2e-5f * (1 + std::fabs(v)))) {
                        if (shown++ < 12) fprintf(stderr, "  mismatch %s[b=%d][%ld]: %.9g vs reference configuration %.9g\n", what, b, i, u, v);
                        return false;
                    }
                }
                return true;
            };
            if (!close("F", f1 + es, f2 + es, sh.neF - 1)) bad++;
            if (!close("G", &G1[es * b * sh.ldg], &G2[es * b * sh.ldg], sh.neG)) bad++;
            continue;
        }
        if (std::memcmp(f1 + es, f2 + es, es * (sh.neF - 1)) != 0) bad++;
        if (std::memcmp(&G1[es * b * sh.ldg], &G2[es * b * sh.ldg], es * sh.neG) != 0) bad++;
    }
    return bad;
}

}  // namespace

int main(int argc, char **argv)
{
    int reps = 40, xbuf = 4;
    Buffers bf;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("| B | N | mission | dtype | max_nt -> tiles x nt | cap | fused | us/eval back to back, per-dispatch events | kernel us (dispatch events) | us/eval back to back, uninstrumented | GB/s alg (uninstrumented) | %% of 8 TB/s | check |\n|---|---|---|---|---|---|---|---|---|---|---|---|---|\n");
    for (int i = 1; i < argc; i++) {
        if (!strncmp(argv[i], "reps=", 5)) { reps = atoi(argv[i] + 5); continue; }
        if (!strncmp(argv[i], "xbuf=", 5)) { xbuf = atoi(argv[i] + 5); bf.release(); bf.sh = Shape{}; continue; }
        if (!strncmp(argv[i], "nt=", 3)) { bf.nt = atoi(argv[i] + 3); continue; }
        if (!strncmp(argv[i], "xcd=", 4)) { bf.xcd = atoi(argv[i] + 4); continue; }
        if (!strncmp(argv[i], "xcdpct=", 7)) { bf.xcdpct = atoi(argv[i] + 7); continue; }       // share of the tiles dealt XCD-contiguously
        if (!strncmp(argv[i], "stagger=", 8)) { g_stagger = atoi(argv[i] + 8); continue; }
        if (!strncmp(argv[i], "variant=", 8)) { g_variant = atoi(argv[i] + 8); continue; }
        if (!strncmp(argv[i], "goff=", 5)) { g_goff = atol(argv[i] + 5); bf.release(); bf.sh = Shape{}; continue; }
        if (!strncmp(argv[i], "ldgpad=", 7)) { g_ldg_pad = atol(argv[i] + 7); bf.release(); bf.sh = Shape{}; continue; }
        if (!strncmp(argv[i], "tail=", 5)) { bf.tail_count = atoi(argv[i] + 5); const char *c = strchr(argv[i], ':'); if (c) bf.tail_nt = atoi(c + 1); continue; }   // tail=count:nt
        if (!strncmp(argv[i], "pat=", 4)) { bf.pat = atoi(argv[i] + 4); bf.release(); bf.sh = Shape{}; continue; }
        int v[7] = {4096, 200, 64, 0, 0, 0, 0};
        int nv = 0;
        for (char *tok = strtok(argv[i], ","); tok && nv < 7; tok = strtok(nullptr, ",")) v[nv++] = atoi(tok);
        const Shape sh = make_shape(v[0], v[1], v[5], v[6]);
        if (sh.B != bf.sh.B || sh.N != bf.sh.N || sh.mission != bf.sh.mission || sh.dtype != bf.sh.dtype || bf.dX.empty()) {
            // keep the X buffers together above the Infinity Cache only when the shape is large anyway
            fill(bf, sh, xbuf);
            const int nt_keep = bf.nt, xcd_keep = bf.xcd;
            const int tail_keep = bf.tail_count; bf.tail_count = 0;
            bf.nt = 1; bf.xcd = 0;
            tolfg::FgArgs r = make_args(bf, 64, 0, 0, 0, bf.dF2, bf.dG2);     // reference result of this shape
            bf.nt = nt_keep; bf.xcd = xcd_keep; bf.tail_count = tail_keep;
            CK(tolfg::launch_fg(r, sh.mission, tolfg::WIND_SHEAR, sh.dtype, sh.dtype == 0 ? 2 : 4, st));
            CK(hipStreamSynchronize(st));
        }
        const int vec = sh.dtype == 0 ? 2 : 4;
        tolfg::FgArgs a = make_args(bf, v[2], v[3], v[4], 0, bf.dF, bf.dG);
        CK(hipMemsetAsync(bf.dF, 0xff, bf.es() * sh.B * sh.ldf, st));
        CK(hipMemsetAsync(bf.dG, 0xff, bf.es() * sh.B * sh.ldg, st));
        CK(tolfg::launch_fg(a, sh.mission, tolfg::WIND_SHEAR, sh.dtype, vec, st));
        CK(hipStreamSynchronize(st));
        const bool approx = sh.dtype == 1 && a.nt > 64;
        const long bad = compare(bf, approx);
        for (int w = 0; w < 5; w++) {
            tolfg::FgArgs aw = make_args(bf, v[2], v[3], v[4], w, bf.dF, bf.dG);
            CK(tolfg::launch_fg(aw, sh.mission, tolfg::WIND_SHEAR, sh.dtype, vec, st));
        }
        CK(hipStreamSynchronize(st));
        static std::vector<hipEvent_t> kev;                 // per-launch events riding on the dispatches
        while ((int)kev.size() < 2 * reps) { hipEvent_t e; CK(hipEventCreate(&e)); kev.push_back(e); }
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; r++) {
            tolfg::FgArgs ar = make_args(bf, v[2], v[3], v[4], r, bf.dF, bf.dG);
            CK(tolfg::launch_fg(ar, sh.mission, tolfg::WIND_SHEAR, sh.dtype, vec, st, kev[2 * r], kev[2 * r + 1]));
        }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        double kern_us = 0;
        for (int r = 0; r < reps; r++) { float k; CK(hipEventElapsedTime(&k, kev[2 * r], kev[2 * r + 1])); kern_us += 1e3 * k / reps; }
        // the same launches without per-dispatch events (they cost 4-16 us per launch, profiles/r02_event_cost.md)
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; r++) {
            tolfg::FgArgs ar = make_args(bf, v[2], v[3], v[4], r, bf.dF, bf.dG);
            CK(tolfg::launch_fg(ar, sh.mission, tolfg::WIND_SHEAR, sh.dtype, vec, st));
        }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms_plain;
        CK(hipEventElapsedTime(&ms_plain, e0, e1));
        const double us_plain = 1e3 * ms_plain / reps;
        // repeat of the check after the timed launches: the counters must have been left at zero
        CK(hipMemsetAsync(bf.dF, 0xff, bf.es() * sh.B * sh.ldf, st));
        CK(hipMemsetAsync(bf.dG, 0xff, bf.es() * sh.B * sh.ldg, st));
        CK(tolfg::launch_fg(a, sh.mission, tolfg::WIND_SHEAR, sh.dtype, vec, st));
        CK(hipStreamSynchronize(st));
        const long bad2 = compare(bf, approx);
        const long neG_eff = bf.pat == tolfg::PATTERN_COMPACT ? sh.c0 + 46L * sh.N + (sh.mission == tolfg::MISSION_G7 ? 30 : 22) : sh.neG;
        const double bytes = (double)bf.es() * sh.B * ((double)sh.n + sh.neF + neG_eff);
        const double us = 1e3 * ms / reps;
        char tailtxt[48] = "";
        if (a.tail_count) snprintf(tailtxt, sizeof tailtxt, " tail %d x (%d x %d)", a.tail_count, a.tail_tiles, a.tail_nt);
        printf("| %d | %d | %s | %s | %d -> %d x %d | %d | %d%s%s%s | %.2f | %.2f | %.2f | %.0f | %.1f | %s |\n", sh.B, sh.N,
               sh.mission == 0 ? "S10" : (sh.mission == 1 ? "G7" : "mixed"), sh.dtype == 0 ? "f64" : "f32", v[2], a.tiles, a.nt, v[3], v[4],
               bf.nt ? " nt" : " plain", bf.xcd ? (bf.xcdpct == 100 ? " xcd" : " xcd-part") : "", tailtxt, us, kern_us, us_plain,
               bytes / (1e3 * us_plain), 100.0 * bytes / (1e3 * us_plain) / 8000.0, (bad || bad2) ? "MISMATCH" : "ok");
        fflush(stdout);
    }
    bf.release();
    return 0;
}
