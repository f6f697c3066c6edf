#!/bin/bash
# round-2 experiment H: fast sincos -- GPU suite, tile sizes again, callback rate, bench configs
mkdir -p gpurun_out/r02h
O=gpurun_out/r02h
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -6 $O/pytest_gpu.log | cut -c1-400
echo "== callback (default)"; timeout -k 10 120 python tools/callback_rate.py 2 2>&1 | tail -10
timeout -k 10 400 tools/bin/fgbench reps=40 nt=1 xcd=1 \
  4096,200,64,8,1 4096,200,32,12,1 4096,200,32,0,1 4096,200,16,0,1 4096,200,16,16,1 4096,200,8,0,1 \
  nt=0 1024,200,64,0,1 1024,200,32,0,1 1024,200,16,0,1 512,200,64,0,1 512,200,32,0,1 512,200,16,0,1 128,200,64,0,1 128,200,32,0,1 128,200,16,0,1 128,200,8,0,1 \
  nt=1 8192,200,64,8,1,2 8192,200,64,12,1,2,1 400,2000,64,8,1 \
  > $O/fgbench.md 2>&1; echo "fgbench exit $?"; cat $O/fgbench.md
for v in "" "TOLFG_TILE_NODES=32" "TOLFG_XCD=0" "TOLFG_NT_STORES=1"; do
echo "== bench B=1024 $v"; env $v timeout -k 10 200 python bench.py --batch 1024 --steps 100 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('ms/step %.4f kernel %.4f min %.4f frac %.3f'%(d['ms_per_step'], r['kernel_ms'], r['kernel_min_ms'], r['frac']))"
done
