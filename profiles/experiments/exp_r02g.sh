#!/bin/bash
# round-2 experiment G: GPU suite, callback rate, bench.py (new layout)
mkdir -p gpurun_out/r02g
O=gpurun_out/r02g
timeout -k 10 600 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -12 $O/pytest_gpu.log | cut -c1-400
grep "worst scaled error per class" $O/pytest_gpu.log | cut -c1-1200
echo "== callback (default)"; timeout -k 10 120 python tools/callback_rate.py 2>&1 | tail -5
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; tail -3 $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02g/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','backend','scaling')}, d['roofline']['frac'], d['roofline']['kernel_ms'])
for r in d.get('configs',[]):
    print(r['config'], r['mode'], r.get('batch'), r.get('dtype'), 'ms/step %.4f'%r.get('ms_per_step',0) if 'ms_per_step' in r else '', 'eval_us %.1f'%r['eval_us'] if 'eval_us' in r else 'us/call %.1f'%r['us_per_call'], '%.3g node-evals/s'%r['node_evals_per_s'], 'frac %.3f'%r['frac_of_hbm_peak'] if 'frac_of_hbm_peak' in r else '')
print(d.get('cpu_baseline',{}).get('value'), d.get('next_compact_pattern',{}).get('frac_of_hbm_peak'))
PY
