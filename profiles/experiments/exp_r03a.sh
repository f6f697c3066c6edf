#!/bin/bash
# round 3, step A: table-driven slab stream vs round 2's build, same box; then the GPU suite
O=gpurun_out/r03a; mkdir -p $O
SHAPES="4096,200,64,8,1,0,0 1024,200,64,0,1,0,0 128,200,64,0,1,0,0 8192,200,64,8,1,2,0 8192,200,64,12,1,2,1 1024,200,64,0,1,2,1 4096,200,64,12,1,0,1 4096,200,64,12,1,1,1"
for exe in fgbench_r02 fgbench; do
  echo "== $exe reference pattern" >> $O/fgbench.md
  timeout -k 10 200 tools/bin/$exe reps=60 nt=1 xcd=1 4096,200,64,8,1,0,0 8192,200,64,8,1,2,0 8192,200,64,12,1,2,1 4096,200,64,12,1,0,1 4096,200,64,12,1,1,1 400,2000,64,8,1,0,0 nt=0 1024,200,64,0,1,0,0 128,200,64,0,1,0,0 1024,200,64,0,1,2,1 >> $O/fgbench.md 2>&1 || exit 1
  echo "== $exe compact pattern" >> $O/fgbench.md
  timeout -k 10 200 tools/bin/$exe reps=60 nt=1 xcd=1 pat=1 4096,200,64,8,0,0,0 4096,200,64,8,1,0,0 4096,200,64,0,0,0,1 4096,200,64,0,1,0,1 8192,200,64,0,0,2,1 8192,200,64,8,0,2,0 >> $O/fgbench.md 2>&1 || exit 1
done
cat $O/fgbench.md
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest_gpu.log
