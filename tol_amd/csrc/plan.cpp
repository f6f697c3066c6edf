// plan.cpp -- host-side launch planning for the fg kernel: tiling of a trajectory and the per-launch
// choices (tile size, resident-wave cap, store flavour, tile order).  Host code only, no HIP calls.
#include "kernels.h"

namespace tolfg {

void plan_tiles(int N, int dtype, int max_nt, int *tiles, int *nt)
{
    // ceil(N/max_nt) tiles of equal size, the size rounded up to 4 nodes so that every tile's x window
    // (11*k0 elements into the row) starts on a 16-byte boundary for both element sizes
    const int widest = dtype == 1 ? 2 * kTileNodes : kTileNodes;       // fp32: up to two nodes per lane (packed kernels)
    int cap = max_nt <= 0 ? kTileNodes : max_nt;
    cap = (cap < 4 ? 4 : (cap > widest ? widest : cap)) & ~3;
    const int t = (N + cap - 1) / cap;
    int per = (N + t - 1) / t;
    per = (per + 3) & ~3;
    if (per > cap) per = cap;
    *nt = per;
    *tiles = (N + per - 1) / per;
}

LaunchPlan plan_launch(const LaunchShape &sh)
{
    // Measured on MI355X with tools/fgbench.cpp (profiles/r02_tile_fused_sweep.md, r02_write_shapes.md, r03_plan.md):
    //  * tile size: lane-per-node tiles smaller than the wavefront lose at EVERY batch size, 64 trajectories included
    //    (a wave's life is mostly fixed latencies: B=128 12.7 / 12.4 / 13.1 / 13.9 / 18.2 us for 52 / 40 / 20 / 16 / 8
    //    nodes, B=1024 43.9 / 44.5 / 49.2 / 53.9 / 67.6, B=2048 89 / 94 / 106 / 116 / 145), so every shape uses the
    //    largest equal tiles <= 64 nodes.
    //  * fp32, two nodes per lane (packed kernels, tiles of up to 128 nodes): a mixed batch gains 3 % (8192: 170.1 ->
    //    165.7 us, 1024: 23.7 -> 22.8, 2048: 44.7 -> 42.2); single-mission batches lose 3 % (4096: 83.0 -> 85.5) and
    //    small launches lose more (B=128: 10.0 -> 11.3), so: mixed batches of at least 4096 64-node tiles, rows aligned.
    //  * outputs beyond the 256 MiB Infinity Cache (F+G > 192 MiB: B >= ~1100 at ts=200, fp64): the slab
    //    stream is non-temporal (plain: -7...-19 %) and the CU is capped at 8 (fp64, packed fp32) or 12 (fp32)
    //    resident tile waves -- fewer concurrent store streams suit the HBM write path (fp64, B=4096:
    //    cap 6 / 7 / 8 / 9 / none -> 173.9 / 166.3 / 162.2 / 163.5 / 160.6-163.8 us; fp32 8 / 12 -> 92.4 / 90.8;
    //    packed fp32, mixed 8192: 6 / 8 / none -> 179.2 / 166.9 / 168.1).
    //  * outputs that fit the cache: plain stores stay on-die (B=1024: 42.6 vs 51.5 us non-temporal,
    //    B=512: 23.8 vs 31.1 us) and want every wave that fits, so no cap.  All the launch's waves are then resident
    //    from the start and would run in step (load, then compute four to a SIMD, and only then the first store):
    //    from 12 waves per CU on, the waves take an issue priority from their slot on the SIMD (FgArgs::stagger), which
    //    lets one wave per SIMD run ahead of the next.  Alternating on/off on the SAME buffers (profiles/r03_plan.md):
    //    B=1024 fp64 43.5 -> 43.1 us, mixed fp64 42.0 -> 41.5, mixed fp32 2048 44.1 -> 43.6: 1-1.5 %, never slower; nothing
    //    at <= 8 waves per CU and nothing beyond the cache, where the waves are out of step anyway.  (A first A/B across
    //    two processes read 5-10 %: that was the allocation-to-allocation spread of these shapes, +-2 %.)
    //  * tiles are dealt to the XCDs in contiguous eighths (+1...+8 %, never slower).
    LaunchPlan p{};
    const bool beyond_cache = sh.out_bytes > 192.0 * 1024 * 1024;
    const long tiles64 = (long)sh.B * ((sh.N + kTileNodes - 1) / kTileNodes);
    const bool packed = sh.dtype == 1 && sh.mission == MISSION_MIXED && sh.aligned && sh.N > kTileNodes && tiles64 >= 4096;
    p.max_nt = packed ? 2 * kTileNodes : kTileNodes;
    p.nt_stores = beyond_cache ? 1 : 0;
    // fp32 compact slabs (184 bytes per node) are the one shape that wants every wave it can get beyond the cache
    // too: cap 12 / none -> 55.1 / 50.7 us at B=4096, 112.0 / 106.0 us for the mixed 8192 (profiles/r02_shape_sweep.md)
    p.waves_per_cu = beyond_cache ? (sh.dtype == 0 || packed ? 8 : (sh.pattern == PATTERN_COMPACT ? 0 : 12)) : 0;
    if (packed && sh.pattern == PATTERN_COMPACT) p.waves_per_cu = 0;
    p.xcd = 1;
    p.stagger = (!beyond_cache && tiles64 >= 12 * 256) ? 1 : 0;
    // one launch per evaluation, except for the compact pattern beyond the cache, where the two-launch form
    // measured 5 % faster (103.3 vs 108.0 us at B=4096: half the bytes per node, so the finalizing waves' tails weigh more)
    p.fused = (sh.pattern == PATTERN_COMPACT && beyond_cache) ? 0 : 1;
    return p;
}

}  // namespace tolfg
