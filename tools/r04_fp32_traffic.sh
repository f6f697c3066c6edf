#!/bin/bash
# VERDICT r3 task 4: where do the extra 4.6 % of HBM bytes of the fp32 launches come from (fp64: 2.1 %)?
# One counter per pass over the 8192-trajectory launch: mixed (the stated config), S10 only, G7 only (G7 rows in fp32 start 8 bytes
# off a 16-byte boundary: shifted streams), fp64 mixed for comparison.  Run on the GPU box from the repo root.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04_fp32
mkdir -p "$OUT"
COMMON="--ts 200 --batch 8192 --steps 30 --warmup 3 --min-warm-seconds 0 --no-calibration --no-cpu-baseline --no-configs"
for cfg in "mixed f32" "S10 f32" "G7 f32" "mixed f64"; do
    set -- $cfg
    tag=$1_$2
    echo "#### $tag"
    for c in FETCH_SIZE WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum; do
        timeout -k 5 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$tag/$c" -o pmc -- python3 bench.py $COMMON --mission $1 --dtype $2 > "$OUT/$tag.$c.log" 2>&1 \
            && python3 tools/pmc_avg.py $(find "$OUT/$tag/$c" -name "*counter_collection.csv" | head -1) || echo "$c: pass failed ($(grep -m1 -E 'Missing|rror' "$OUT/$tag.$c.log" | cut -c1-120))"
    done
    python3 - <<PY
import sys; sys.path.insert(0, ".")
import tol_amd, bench as BN
air = BN.AIRCRAFT5 if "$1" == "mixed" else ("tempest",)
bt = tol_amd.Batch("$1", air, ts=200, dtype="$2")
bt.set_trajectories(BN.make_trajectories(tol_amd, 8192, 0, "$1", len(air)))
n, neF, neG = bt.n, bt.neF, bt.neG
es = 8 if "$2" == "f64" else 4
print("algorithmic bytes per launch %.1f MB (x read %.1f MB, F+G written %.1f MB)" % (bt.algorithmic_bytes(8192) / 1e6, es * 8192 * n / 1e6, (bt.algorithmic_bytes(8192) - es * 8192 * n) / 1e6))
PY
done
find "$OUT" -name "*.csv" -size +2M -delete
