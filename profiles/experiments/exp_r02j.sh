#!/bin/bash
# round-2 experiment J: GPU suite; compact-pattern launch choices; TA counters one per pass
mkdir -p gpurun_out/r02j
O=gpurun_out/r02j
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest_gpu.log | cut -c1-400
for v in "" "TOLFG_XCD=0" "TOLFG_FUSED=0" "TOLFG_WAVES_PER_CU=0" "TOLFG_WAVES_PER_CU=6" "TOLFG_WAVES_PER_CU=10" "TOLFG_NT_STORES=0" "TOLFG_XCD=0 TOLFG_FUSED=0"; do
echo "== compact $v"; env $v timeout -k 10 200 python bench.py --pattern compact --steps 100 --no-cpu-baseline --no-configs 2>/dev/null | python tools/show_bench.py | head -1
done
for c in TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_BUSY_avr; do
timeout -k 10 200 bash tools/pmc_pass.sh ta_$c "$c" > $O/pmc_$c.txt 2>&1; echo "pass $c exit $?"; tail -3 $O/pmc_$c.txt | cut -c1-200
done
