#!/bin/bash
# round-2 experiment I: GPU suite, native callback timing, bench, rocprofv3 stats + PMC passes
mkdir -p gpurun_out/r02i
O=gpurun_out/r02i
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest_gpu.log | cut -c1-400
echo "== callback"; timeout -k 10 120 python tools/callback_rate.py 2>&1 | tail -5
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; tail -2 $O/bench.err
python tools/show_bench.py $O/bench.json
timeout -k 10 600 bash tools/profile_gpu.sh r02 > $O/profile.log 2>&1; echo "profile exit $?"; tail -3 $O/profile.log
timeout -k 10 200 bash tools/pmc_pass.sh ta "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WRITE_WAVEFRONTS_sum" > $O/pmc_ta.txt 2>&1; echo "ta pass exit $?"; cat $O/pmc_ta.txt | tail -6
timeout -k 10 200 bash tools/pmc_pass.sh wr "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" > $O/pmc_wr.txt 2>&1; echo "wr pass exit $?"; cat $O/pmc_wr.txt | tail -6
