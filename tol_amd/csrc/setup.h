// setup.h -- what problem construction hands to SNOPT: sizes, sparsity pattern, initial guess, bounds.
#ifndef TOLFG_SETUP_H_
#define TOLFG_SETUP_H_

#include "params.h"

namespace tolfg {

// ref: src/problem.cpp:151-152 (n, neF); neG is what problem::countG (src/problem.cpp:813-919)
// counts, in closed form; c0 = position in G of node 0's 104-entry slab.
struct Sizes {
    int mission;   // MISSION_S10 | MISSION_G7
    int pattern;   // PATTERN_REFERENCE | PATTERN_COMPACT (slab_table.h)
    int N, nb, n, neF, neG, c0, slab;
};
Sizes make_sizes(int mission, int N, int pattern = 0);

// (iGfun, jGvar), 0-based, in countG's row-major order -- generated in O(neG) instead of the
// reference's O(neF*n) probing (83 s / 20 GB at ts = 2000).
void make_pattern(const Sizes &sz, int *iGfun, int *jGvar);

struct Start { double xi, yi, zi; };

// ref: problemS10::InitialCond src/problemS10.cpp:19-219, problemG7::InitialCond + RotateYaw
// src/problemG7.cpp:19-217,520-542.  chi_d only matters for G7.
void initial_guess(const Sizes &sz, const aircraft &ac, const Start &st, double chi_d, double *x);

// ref: problem::setLimits src/problem.cpp:198-365 with the node-0 constants the constructor
// hard-codes (src/problem.cpp:80-134).
void set_limits(const Sizes &sz, const aircraft &ac, const limit &lm, const Start &st,
                double *xlow, double *xupp, double *Flow, double *Fupp);

}  // namespace tolfg
#endif
