#!/bin/bash
# VERDICT r3 task 6: the callback (B = 1) as ONE workgroup of 2-4 waves (fg_single_kernel, TOLFG_FORCE_SINGLE_LAUNCH=1) against the
# plan's choice since round 4 -- 5 tile workgroups on 5 CUs with the completion word -- and other tile sizes.
# Run on the GPU box from the repo root.
for rep in 1 2; do
echo "== pass $rep: one workgroup per trajectory (TOLFG_FORCE_SINGLE_LAUNCH=1)"
TOLFG_FORCE_SINGLE_LAUNCH=1 python3 tools/callback_rate.py | grep " 200 \| 100 "
echo "== pass $rep: the plan (5 tiles from ts = 100)"
python3 tools/callback_rate.py | grep " 200 \| 100 "
for nt in 52 32 28; do
    echo "== pass $rep: tile-per-workgroup, TOLFG_TILE_NODES=$nt"
    TOLFG_NO_SINGLE_LAUNCH=1 TOLFG_TILE_NODES=$nt python3 tools/callback_rate.py | grep " 200 \| 100 "
done
done
