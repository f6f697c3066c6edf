// slab_table.h -- layout of one node's Jacobian slab, shared by the host pattern generator and the
// device store loop so that the two cannot drift apart.
//
// Reference pattern (what problem::countG produces, src/problem.cpp:813-919): 8 rows x 13 entries
// [dt | x y z Va gam chi phi CL dphi dCL T | next-node state] = 104 entries per node, 58 of them
// structural zeros that the reference keeps because dynamicsGradients raises Gnonzero for every
// column of the node (src/problem.cpp:1194-1198).
// Compact pattern: only the 46 entries dynamicsGradients ever assigns
// (src/problem.cpp:1080-1186) plus the next-node ones, in the same row-major order.
#ifndef TOLFG_SLAB_TABLE_H_
#define TOLFG_SLAB_TABLE_H_

namespace tolfg {

constexpr int SLAB_FULL = 104;
constexpr int SLAB_COMPACT = 46;
// LDS row slots: 0..31 computed values, then the three constants
constexpr int SL_ZERO = 32, SL_ONE = 33, SL_MONE = 34;

struct SlabTableFull { unsigned char c[SLAB_FULL]; };
struct SlabTableCompact { unsigned char c[SLAB_COMPACT]; unsigned char full_index[SLAB_COMPACT]; };

// which LDS slot feeds slab element e = 13*(row-1) + col
constexpr SlabTableFull make_slab_table()
{
    SlabTableFull t{};
    for (int i = 0; i < SLAB_FULL; i++) t.c[i] = SL_ZERO;
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

// the compact slab = the reference slab without its structural zeros, order preserved
constexpr SlabTableCompact make_compact_table()
{
    const SlabTableFull f = make_slab_table();
    SlabTableCompact t{};
    int n = 0;
    for (int e = 0; e < SLAB_FULL; e++)
        if (f.c[e] != SL_ZERO) {
            t.c[n] = f.c[e];
            t.full_index[n] = (unsigned char)e;
            n++;
        }
    return t;
}

constexpr int count_kept()
{
    const SlabTableFull f = make_slab_table();
    int n = 0;
    for (int e = 0; e < SLAB_FULL; e++) n += f.c[e] != SL_ZERO;
    return n;
}
static_assert(count_kept() == SLAB_COMPACT, "46 structural non-zeros per node");

}  // namespace tolfg
#endif
