#!/bin/bash
# VERDICT r3 task 3: what binds the launches whose outputs fit the cache (configs[3], B = 1024 fp64: 0.56-0.60 of peak)?
# rocprofv3 kernel-trace + stats, then one counter per pass, for B = 1024 (plain stores, the plan's choice), B = 2048 forced
# plain, and B = 2048 as planned (non-temporal) for comparison.  Run on the GPU box from the repo root.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04_incache
mkdir -p "$OUT"
ARGS="--mission S10 --ts 200 --steps 40 --warmup 5 --min-warm-seconds 0 --no-calibration --no-cpu-baseline --no-configs"
run() {    # tag, batch, extra env
    local tag=$1 B=$2; shift 2
    echo "#### $tag"
    env "$@" true
    ( export "$@" 2>/dev/null; timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag/stats" -o stats -- python3 bench.py $ARGS --batch $B > "$OUT/$tag.stats.log" 2>&1 )
    grep -h "fg_kernel" $(find "$OUT/$tag/stats" -name "*kernel_stats.csv") | cut -c1-260
    for c in $COUNTERS; do
        ( export "$@" 2>/dev/null; timeout -k 5 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$tag/$c" -o pmc -- python3 bench.py $ARGS --batch $B > "$OUT/$tag.$c.log" 2>&1 ) \
            && python3 tools/pmc_avg.py $(find "$OUT/$tag/$c" -name "*counter_collection.csv" | head -1) | grep -v "^$" || echo "$c: pass failed ($(grep -m1 -E 'Missing|error|rror' "$OUT/$tag.$c.log" | cut -c1-120))"
    done
}
COUNTERS="FETCH_SIZE WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum TCC_WRITEBACK_sum TCC_TAG_STALL_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
run b1024_plain 1024 TOLFG_DUMMY=1
run b2048_plain 2048 TOLFG_NT_STORES=0 TOLFG_WAVES_PER_CU=0
COUNTERS="WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_sum TCC_WRITEBACK_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
run b2048_planned 2048 TOLFG_DUMMY=1
find "$OUT" -name "*.csv" -size +2M -delete
