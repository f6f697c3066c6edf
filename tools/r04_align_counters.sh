#!/bin/bash
# Write requests of the launch with the slab streams cut to 64-byte boundaries against the 16-byte form (TOLFG_STREAM_ALIGN16=1).
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04_align
mkdir -p "$OUT"
COMMON="--ts 200 --batch 8192 --steps 30 --warmup 3 --min-warm-seconds 0 --no-calibration --no-cpu-baseline --no-configs"
for dt in f64 f32; do
for a16 in 0 1; do
    export TOLFG_STREAM_ALIGN16=$a16
    tag=${dt}_align16_$a16
    echo "#### $tag"
    for c in WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TA_DATA_STALLED_BY_TC_CYCLES_sum; do
        timeout -k 5 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$tag/$c" -o pmc -- python3 bench.py $COMMON --dtype $dt > "$OUT/$tag.$c.log" 2>&1 \
            && python3 tools/pmc_avg.py $(find "$OUT/$tag/$c" -name "*counter_collection.csv" | head -1) || echo "$c: pass failed"
    done
done
done
find "$OUT" -name "*.csv" -size +2M -delete
