#!/bin/bash
export TOLFG_IPB=1
timeout -k 10 200 python bench.py --no-cpu-baseline --no-callback 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ipb1', d['value'], d['ms_per_step'], d['roofline'])"
export TOLFG_IPB=2
timeout -k 10 200 python bench.py --no-cpu-baseline --no-callback 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ipb2', d['value'], d['ms_per_step'], d['roofline'])"
unset TOLFG_IPB
timeout -k 10 200 python bench.py --no-cpu-baseline --no-callback 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'], d['roofline'])"
timeout -k 5 60 ./tools/bin/fgprobe2 4096 200 30 1
export TOLFG_IPB=1
timeout -k 10 200 bash tools/pmc_pass.sh w1 "WRITE_SIZE"
timeout -k 10 200 bash tools/pmc_pass.sh f1 "FETCH_SIZE"
