#!/bin/bash
timeout -k 10 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for r in 1 2; do timeout -k 5 60 ./tools/bin/fgprobe 4096 200 30 1; done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-callback 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
