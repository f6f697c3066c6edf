#!/bin/bash
# scratch driver for one gpurun call: counter list, f32 bench, three PMC passes
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/counters.txt 2>&1
wc -l gpurun_out/counters.txt
timeout -k 10 200 python bench.py --dtype f32 --no-cpu-baseline --no-callback > gpurun_out/bench_f32.log 2>&1
tail -c 700 gpurun_out/bench_f32.log
echo
timeout -k 10 200 bash tools/pmc_pass.sh sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
timeout -k 10 200 bash tools/pmc_pass.sh sq2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT"
timeout -k 10 200 bash tools/pmc_pass.sh g1 "GRBM_GUI_ACTIVE GRBM_COUNT"
