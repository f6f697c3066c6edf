#!/bin/bash
# dynamic instruction mix and wait states: fp32 mixed reference pattern, fp64 compact, fp64 reference
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > gpurun_out/sq_counters.txt
bash tools/pmc_fgbench.sh r03_f32ref tools/bin/fgbench "reps=4 nt=1 xcd=1 8192,200,64,12,1,2,1"
bash tools/pmc_fgbench.sh r03_f64cmp tools/bin/fgbench "reps=4 nt=1 xcd=1 pat=1 4096,200,64,8,0,0,0"
bash tools/pmc_fgbench.sh r03_f64ref tools/bin/fgbench "reps=4 nt=1 xcd=1 4096,200,64,8,1,0,0"
