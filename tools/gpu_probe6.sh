#!/bin/bash
P=./tools/bin/fgprobe
for v in 1 513 1025 1537 2049 8193 9729; do timeout -k 5 60 $P 4096 200 30 $v | head -1; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st2 -o st -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-callback > gpurun_out/st2.log 2>&1
head -4 gpurun_out/st2/st_kernel_stats.csv | cut -c1-160
