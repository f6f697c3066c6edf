#!/bin/bash
# occupancy sweep of fg_kernel (stamped probe build; shares, not benchmark numbers); TOLFG_WAVES_PER_CU: measurement build only
export TOLFG_LIBRARY=${TOLFG_LIBRARY:-$PWD/tol_amd/lib/libtolfg_measure.so}
for r in 1 2; do
  for w in 0 8 7 6 5 4; do timeout -k 5 60 ./tools/bin/fgprobe 4096 200 30 1 $w | head -1; done
done
for w in 0 7 6 5; do timeout -k 5 60 ./tools/bin/fgprobe 512 2000 30 1 $w | head -1; done
for w in 0 7 6 5; do TOLFG_WAVES_PER_CU=$w timeout -k 10 120 python bench.py --steps 100 --no-cpu-baseline --no-configs 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench waves', '$w', '%.4g'%d['value'], '%.1f GB/s'%d['roofline']['achieved'], 'step %.1f us'%(1e3*d['ms_per_step']))"; done
for w in 0 8 6 4; do TOLFG_WAVES_PER_CU=$w timeout -k 10 120 python bench.py --steps 100 --pattern compact --no-cpu-baseline --no-configs 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('compact waves', '$w', '%.4g'%d['value'], '%.1f GB/s'%d['roofline']['achieved'])"; done
for w in 0 12 8 6; do TOLFG_WAVES_PER_CU=$w timeout -k 10 120 python bench.py --steps 100 --dtype f32 --no-cpu-baseline --no-configs 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('f32 waves', '$w', '%.4g'%d['value'], '%.1f GB/s'%d['roofline']['achieved'])"; done
