// plan.cpp -- host-side launch planning for the fg kernel: tiling of a trajectory and the per-launch
// choices (tile size, resident-wave cap, store flavour, tile order).  Host code only, no HIP calls.
#include "kernels.h"

namespace tolfg {

namespace {
constexpr int kTileNodes = 64;     // nodes per dynamics tile = wavefront width (kTileNodes in kernels.hip)
}

void plan_tiles(int N, int dtype, int max_nt, int *tiles, int *nt)
{
    // ceil(N/max_nt) tiles of equal size, the size rounded up to 4 nodes so that every tile's x window
    // (11*k0 elements into the row) starts on a 16-byte boundary for both element sizes
    (void)dtype;
    int cap = max_nt <= 0 ? kTileNodes : max_nt;
    cap = (cap < 4 ? 4 : (cap > kTileNodes ? kTileNodes : cap)) & ~3;
    const int t = (N + cap - 1) / cap;
    int per = (N + t - 1) / t;
    per = (per + 3) & ~3;
    if (per > cap) per = cap;
    *nt = per;
    *tiles = (N + per - 1) / per;
}

LaunchPlan plan_launch(double out_bytes, int dtype, int pattern)
{
    // Measured on MI355X with tools/fgbench.cpp (profiles/r02_tile_fused_sweep.md, r02_write_shapes.md):
    //  * tile size: lane-per-node tiles smaller than the wavefront lose -- the node arithmetic costs the
    //    same ~5000 cycles per wave whatever the number of active lanes (32 nodes: equal at best;
    //    16 nodes: 47-53 % of peak; 8 nodes: 33 %) -- so every shape uses the largest equal tiles <= 64.
    //  * outputs beyond the 256 MiB Infinity Cache (F+G > 192 MiB: B >= ~1100 at ts=200, fp64): the slab
    //    stream is non-temporal (plain: -7...-19 %) and the CU is capped at 8 (fp64) or 12 (fp32)
    //    resident tile waves -- fewer concurrent store streams suit the HBM write path (fp64, B=4096:
    //    cap 6 / 7 / 8 / 9 / none -> 173.9 / 166.3 / 162.2 / 163.5 / 160.6-163.8 us; fp32 8 / 12 -> 92.4 / 90.8).
    //  * outputs that fit the cache: plain stores stay on-die (B=1024: 42.6 vs 51.5 us non-temporal,
    //    B=512: 23.8 vs 31.1 us) and want every wave that fits, so no cap.
    //  * tiles are dealt to the XCDs in contiguous eighths (+1...+8 %, never slower).
    LaunchPlan p{};
    const bool beyond_cache = out_bytes > 192.0 * 1024 * 1024;
    p.max_nt = kTileNodes;
    p.nt_stores = beyond_cache ? 1 : 0;
    // fp32 compact slabs (184 bytes per node) are the one shape that wants every wave it can get beyond the cache
    // too: cap 12 / none -> 55.1 / 50.7 us at B=4096, 112.0 / 106.0 us for the mixed 8192 (tools/exp_r02ab.sh)
    p.waves_per_cu = beyond_cache ? (dtype == 0 ? 8 : (pattern == PATTERN_COMPACT ? 0 : 12)) : 0;
    p.xcd = 1;
    // one launch per evaluation, except for the compact pattern beyond the cache, where the two-launch form
    // measured 5 % faster (103.3 vs 108.0 us at B=4096: half the bytes per node, so the finalizing waves' tails weigh more)
    p.fused = (pattern == PATTERN_COMPACT && beyond_cache) ? 0 : 1;
    return p;
}

}  // namespace tolfg
