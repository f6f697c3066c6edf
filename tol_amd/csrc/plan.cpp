// plan.cpp -- host-side launch planning for the fg kernel: tiling of a trajectory and the per-launch
// choices (tile size, resident-wave cap, store flavour, tile order).  Host code only, no HIP calls.
#include "kernels.h"

namespace tolfg {

namespace {
constexpr int TILE = 64;           // nodes per dynamics tile = wavefront width (kernels.hip)
}

void plan_tiles(int N, int dtype, int max_nt, int *tiles, int *nt)
{
    // ceil(N/max_nt) tiles of equal size, the size rounded up to 4 nodes so that every tile's x window
    // (11*k0 elements into the row) starts on a 16-byte boundary for both element sizes
    (void)dtype;
    int cap = max_nt <= 0 ? TILE : max_nt;
    cap = (cap < 4 ? 4 : (cap > TILE ? TILE : cap)) & ~3;
    const int t = (N + cap - 1) / cap;
    int per = (N + t - 1) / t;
    per = (per + 3) & ~3;
    if (per > cap) per = cap;
    *nt = per;
    *tiles = (N + per - 1) / per;
}

int pick_tile_nodes(int B, int N, int dtype, int pattern)
{
    (void)B; (void)N; (void)dtype; (void)pattern;
    return TILE;
}

}  // namespace tolfg
