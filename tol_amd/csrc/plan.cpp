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
    //  * outputs beyond the 256 MiB Infinity Cache (F+G > 240 MiB, see round 4 below: B >= ~1400 at ts=200, fp64): the slab
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
    //  * round 4, where "fits the cache" ends (tools/plan_ab.py, variants in turn on the same buffers, S10 fp64, us per evaluation,
    //    plain | non-temporal + cap 8): B=1088 (191 MiB) 47.1 | 48.2, 1152 (202 MiB) 49.1 | 50.4, 1280 (225 MiB) 53.4 | 54.1, 1408
    //    (247 MiB) 58.7 | 58.5, 1536 (270 MiB) 65.5 | 63.6, 1792 73.7 | 66.9, mixed 1536 63.9 | 59.3, mixed 2048 85.8 | 77.2: the two
    //    meet where the outputs reach the 256 MiB Infinity Cache, so the switch sits at 240 MiB (it was 192 MiB through round 3).
    //  * round 4, a tile's rows through LDS in two passes of 32 nodes (FgArgs::sub_nodes; 9.2 instead of 14.8 KB per fp64 wave: 16
    //    instead of 10 resident waves per CU): launches in the cache with 11-17 tile waves per CU gain 2-6 % (S10 768 / 896 / 1024 /
    //    1088: 31.6 -> 30.1, 34.7 -> 34.0, 42.1 -> 40.1 and 41.6 -> 40.7 on two boxes, 46.9 -> 45.5; G7 768 / 1024: 31.7 -> 29.9, 43.6 ->
    //    41.3; mixed 768 / 1024 / 1088: 31.5 -> 30.0, 43.7 -> 42.0 (one box: 38.5 -> 39.5), 47.0 -> 45.0); nothing at <= 10 waves per
    //    CU (B = 256, 512, 640: they all fit anyway) or from 18 up (1152 ... 2048), where the CU's store path, not residency, sets
    //    the pace (profiles/r04_incache_counters.md).  fp32 rows are half the size and never the limit.  With these kernels held to
    //    128 VGPRs (kernels.hip: min_waves_per_simd; the two-pass loop itself had pushed them to 133-135 = 12 waves per CU) the gain
    //    grew and the range widened: S10 704 / 896 / 1024 / 1088 / 1152 / 1280: 31.6 -> 28.2, 35.3 -> 34.0, 45.4 -> 40.1, 44.7 -> 41.1,
    //    47.6 -> 44.2, 50.6 -> 49.4 us; mixed 1024 45.1 -> 39.3, G7 1024 44.9 -> 38.7; nothing at 640 (10 per CU) and from 1408 on
    //    (beyond the cache): 11-20 tile waves per CU (profiles/r04/sub32_sweep_128vgpr.txt).
    //  * round 4, the SNOPT callback (B = 1) at ts >= 100 as tile workgroups on different CUs with the completion word instead of
    //    one workgroup of 2-4 waves, every tile wave fetching the finalizer's x values (dt, node 0, node N) at its start: ts = 200
    //    19.8 us per call as one workgroup, 17.2 / 16.8 / 16.4 / 16.2 as 4 / 5 / 7 / 8 tile workgroups; ts = 100 16.8 as one workgroup,
    //    15.0 as 5 tiles of 20, 15.5 as 4 tiles of 28 (profiles/r04_callback_tiles.md): tiles of <= 28 nodes, at least 5 of them.
    LaunchPlan p{};
    const bool beyond_cache = sh.out_bytes > 240.0 * 1024 * 1024;
    const long tiles64 = (long)sh.B * ((sh.N + kTileNodes - 1) / kTileNodes);
    const bool packed = sh.dtype == 1 && sh.mission == MISSION_MIXED && sh.aligned && sh.N > kTileNodes && tiles64 >= 4096;
    p.max_nt = packed ? 2 * kTileNodes : kTileNodes;
    // a few short trajectories: one workgroup per trajectory, one launch (fg_single_kernel) -- except the callback's single
    // trajectory of 100+ nodes, which is quicker as 5 tiles on 5 CUs
    p.single = (sh.B <= 8 && sh.N <= 256 && sh.mission != MISSION_MIXED) ? 1 : 0;
    // (with the Jacobian wanted: a call for F alone -- snOptA's line searches -- stores 13 KB and is quicker as one workgroup:
    // 12.0-13.6 against 14.1-14.6 us at ts = 200)
    if (p.single && sh.B == 1 && sh.N >= 100 && sh.needG) {
        p.single = 0;
        const int t = (sh.N + 27) / 28 > 5 ? (sh.N + 27) / 28 : 5;      // tiles of <= 28 nodes, at least 5 of them
        p.max_nt = ((sh.N + t - 1) / t + 3) & ~3;
    }
    const long cus = sh.cus > 0 ? sh.cus : 256;
    const long waves_per_cu_launched = (tiles64 + cus - 1) / cus;
    p.sub_nodes = (!beyond_cache && sh.dtype == 0 && !p.single && waves_per_cu_launched > 10 && waves_per_cu_launched <= 20) ? 32 : 0;
    p.nt_stores = beyond_cache ? 1 : 0;
    // fp32 compact slabs (184 bytes per node) are the one shape that wants every wave it can get beyond the cache
    // too: cap 12 / none -> 55.1 / 50.7 us at B=4096, 112.0 / 106.0 us for the mixed 8192 (profiles/r02_shape_sweep.md)
    // (round 4, on placed output buffers -- tolfg_batch_alloc_outputs -- the cap matters less: fp64 mixed 8192 cap 6 / 8 / 10 / 12 /
    // none -> 314.9 / 280.8 / 282.1 / 280.3 / 280.8 us; packed fp32 mixed 8192 cap 6 / 8 / 12 / 16 / none -> 181.0 / 158.8 / 154.7 / 154.7 /
    // 154.9 us: the packed kernels take 12 like the other fp32 launches, profiles/r04/plan_ab_placed.txt)
    p.waves_per_cu = beyond_cache ? (sh.dtype == 0 ? 8 : (sh.pattern == PATTERN_COMPACT ? 0 : 12)) : 0;
    if (packed && sh.pattern == PATTERN_COMPACT) p.waves_per_cu = 0;
    p.xcd = 1;
    p.stagger = (!beyond_cache && tiles64 >= 12 * cus) ? 1 : 0;
    // one launch per evaluation, except for the compact pattern beyond the cache, where the two-launch form
    // measured 5 % faster (103.3 vs 108.0 us at B=4096: half the bytes per node, so the finalizing waves' tails weigh more)
    p.fused = (sh.pattern == PATTERN_COMPACT && beyond_cache) ? 0 : 1;
    return p;
}

}  // namespace tolfg
