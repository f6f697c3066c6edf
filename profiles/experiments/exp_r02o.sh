#!/bin/bash
# round-2 experiment O: LDS sized to the tile (11 waves per CU uncapped at ts=200); where the ~6 us between evaluations go
mkdir -p gpurun_out/r02o
O=gpurun_out/r02o
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log | cut -c1-300
for bin in fgbench fgbench_ntsmall; do
echo "== $bin"
timeout -k 10 300 tools/bin/$bin reps=40 nt=1 xcd=1 4096,200,64,8,1 4096,200,64,8,1 nt=0 1024,200,64,0,1 1024,200,64,10,1 1024,200,64,9,1 512,200,64,0,1 512,200,64,9,1 256,200,64,0,1 > $O/$bin.md 2>&1; cat $O/$bin.md
done
