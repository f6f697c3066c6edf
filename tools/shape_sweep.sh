#!/bin/bash
# throughput across problem shapes (not the headline; for the record)
run() { timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-configs --no-native-multi "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('| %s | %s | %d | %d | %s | %s | %.3g | %.0f | %.1f |' % (c['mission'], c['aircraft'], c['ts'], c['batch_per_gpu'], d['dtype'], c['pattern'], d['value'], r['achieved'], 100*r['frac']))"; }
echo "| mission | air-frame | ts | batch | dtype | pattern | node-evals/s (whole step) | evaluation GB/s | % of 8 TB/s |"
echo "|---|---|---|---|---|---|---|---|---|"
run --mission S10 --ts 100 --batch 8192
run --mission S10 --ts 200 --batch 4096
run --mission S10 --ts 200 --batch 512
run --mission S10 --ts 200 --batch 1024
run --mission S10 --ts 200 --batch 1536
run --mission S10 --ts 200 --batch 2048
run --mission S10 --ts 500 --batch 1600
run --mission S10 --ts 2000 --batch 400 --aircraft skywalker
run --mission G7 --ts 200 --batch 4096
run --mission G7 --ts 100 --batch 8192
run --mission S10 --ts 200 --batch 4096 --dtype f32
run --mission S10 --ts 200 --batch 4096 --pattern compact
run --mission S10 --ts 2000 --batch 400 --pattern compact --aircraft skywalker
run --mission G7 --ts 200 --batch 4096 --pattern compact --dtype f32
run --mission S10 --ts 200 --batch 256
run --mission S10 --ts 200 --batch 128
run --mission mixed --ts 200 --batch 8192
run --mission mixed --ts 200 --batch 8192 --dtype f32
run --mission G7 --ts 200 --batch 4096 --dtype f32
run --mission S10 --ts 201 --batch 4096
