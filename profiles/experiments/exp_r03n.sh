#!/bin/bash
# clocks and power while the headline loop runs, then the same for a pure fill; then the rocprofv3 passes
O=gpurun_out/r03n; mkdir -p $O
sample() {  # $1 = tag: sample rocm-smi at ~4 Hz until the file $O/stop exists
  rm -f $O/stop
  while [ ! -f $O/stop ]; do
    { date +%s.%N; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power"; } >> $O/smi_$1.txt
    sleep 0.2
  done
}
rocm-smi --showclocks --showpower > $O/smi_idle.txt 2>&1
sample headline & SP=$!
timeout -k 10 200 python bench.py --steps 30000 --warmup 200 --no-configs --no-cpu-baseline > $O/bench_long.json 2> $O/bench_long.err; echo "bench exit $?"
touch $O/stop; wait $SP
sample s10 & SP=$!
timeout -k 10 200 python bench.py --mission S10 --batch 4096 --steps 60000 --warmup 200 --no-configs --no-cpu-baseline > $O/bench_s10.json 2> $O/bench_s10.err; echo "bench exit $?"
touch $O/stop; wait $SP
sample fill & SP=$!
timeout -k 10 120 python - > $O/fill.txt 2>&1 <<'PY'
import torch, time
x = torch.empty(200_000_000, dtype=torch.float32, device="cuda")      # 800 MB
for _ in range(20): x.fill_(1.0)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 40000
for _ in range(n): x.fill_(2.0)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("fill 800 MB x %d: %.1f us each = %.0f GB/s" % (n, 1e6 * dt / n, 0.8 * n / dt))
PY
touch $O/stop; wait $SP
cat $O/fill.txt
python tools/show_bench.py $O/bench_long.json | head -1; python tools/show_bench.py $O/bench_s10.json | head -1
for t in headline s10 fill; do echo "== $t"; grep -c sclk $O/smi_$t.txt; grep -E "sclk|Power" $O/smi_$t.txt | sed -n '20,26p'; done
timeout -k 10 600 bash tools/profile_gpu.sh r03 > $O/profile.log 2>&1; echo "profile exit $?"; tail -3 $O/profile.log
