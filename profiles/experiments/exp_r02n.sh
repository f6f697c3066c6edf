#!/bin/bash
# round-2 experiment N: same-box A/B of the build before the shifted slab stream (3c61bd0) and HEAD; cost of the timing events
one() { python tools/show_bench.py | head -1 | cut -c1-230; }
for rep in 1 2 3; do
echo "== rep $rep: 3c61bd0 then HEAD"
(cd tools/bin/r2atree && timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline --no-configs 2>/dev/null) | one
timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline --no-configs 2>/dev/null | one
done
echo "== HEAD without timing events (TOLFG_BENCH_NO_EVENTS=1)"
TOLFG_BENCH_NO_EVENTS=1 timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline --no-configs 2>/dev/null | one
TOLFG_BENCH_NO_EVENTS=1 timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline --no-configs 2>/dev/null | one
