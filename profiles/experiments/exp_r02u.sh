#!/bin/bash
# round-2 experiment U: own fp32 sin/cos -- GPU suite, fp32 timings against the library routine (same box)
mkdir -p gpurun_out/r02u
O=gpurun_out/r02u
timeout -k 10 600 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log | cut -c1-300; grep "worst scaled error per class" $O/pytest_gpu.log | cut -c1-1500
for rep in 1 2; do for bin in fgbench fgbench_libsc; do
echo "== $bin (rep $rep)"
timeout -k 10 300 tools/bin/$bin reps=40 nt=1 xcd=1 4096,200,64,12,1,0,1 4096,200,64,12,1,1,1 8192,200,64,12,1,2,1 4096,200,16,0,1,0,1 nt=0 1024,200,64,0,1,0,1 2>&1 | tail -5
done; done
