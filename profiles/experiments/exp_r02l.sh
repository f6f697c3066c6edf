#!/bin/bash
# round-2 experiment L: round-1 build vs round-2 build on the SAME box (reference and compact patterns), interleaved
mkdir -p gpurun_out/r02l
O=gpurun_out/r02l
R1=tools/bin/r1tree
one() { python tools/show_bench.py | head -1 | cut -c1-220; }
r1line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('r1: value %.4g ms/step %.4f fg_kernel %.4f ms (fg only) -> %.3f of peak' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['frac']))"; }
for rep in 1 2; do
for pat in reference compact; do
echo "== $pat (rep $rep)"
(cd $R1 && timeout -k 10 200 python bench.py --pattern $pat --steps 100 --no-cpu-baseline --no-callback 2>/dev/null) | r1line
timeout -k 10 200 python bench.py --pattern $pat --steps 100 --no-cpu-baseline --no-configs 2>/dev/null | one
done
done
echo "== fp32 sweep"; timeout -k 10 300 python tests/fp32_sweep.py > $O/fp32_sweep.md 2>$O/fp32.err; echo "exit $?"; tail -3 $O/fp32.err; cat $O/fp32_sweep.md
echo "== rehearse 2 ranks (gloo, one GPU)"; bash tools/rehearse_ranks.sh 2>&1 | tail -3 | cut -c1-300
