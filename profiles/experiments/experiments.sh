#!/bin/bash
# All one-off GPU experiments of rounds 2-4 in one place: `profiles/experiments/experiments.sh <name>` runs ONE of them (one gpurun
# call each, from the repo root on the GPU box: `gpurun -- bash profiles/experiments/experiments.sh r03u`); `list` prints the names,
# the question each one asked and the note that holds the answer.  They are the record of how the numbers in profiles/*.md and
# tol_amd/csrc/plan.cpp were obtained, not product tooling (tools/README.md).  Round-2 entries expect the round-2 harness arguments.
# (The bodies are not indented: several hold here-documents.)
name=${1:-list}
# the TOLFG_* switches these experiments set exist in the measurement build only (tol_amd/csrc/knobs.h; round 5)
export TOLFG_LIBRARY=${TOLFG_LIBRARY:-$PWD/tol_amd/lib/libtolfg_measure.so}
case "$name" in
r02a)
# round-2 experiment A (one gpurun call): write-stream shapes, tile size x cap x fused sweep, callback trace
mkdir -p gpurun_out/r02a
O=gpurun_out/r02a
echo "== wrbench (mode 4: nt 16-B stores, S KiB sequential per wave)" > $O/wrbench.txt
for lds in 0 23400; do for S in 2 4 8 13 16 26 43; do timeout -k 5 60 tools/bin/wrbench 4 $S $lds >> $O/wrbench.txt 2>&1; done; done
echo "== wrbench mode 6 (read 1 KiB + S KiB nt stores per wave)" >> $O/wrbench.txt
for lds in 0 23400; do for S in 4 7 13; do timeout -k 5 60 tools/bin/wrbench 6 $S $lds 8 1 >> $O/wrbench.txt 2>&1; done; done
cat $O/wrbench.txt
echo "== fgbench"
timeout -k 10 400 tools/bin/fgbench reps=40 \
  4096,200,64,7,0 4096,200,64,7,1 4096,200,64,0,1 4096,200,64,8,1 4096,200,64,6,1 \
  4096,200,48,0,1 4096,200,48,8,1 4096,200,48,10,1 \
  4096,200,32,0,1 4096,200,32,8,1 4096,200,32,10,1 4096,200,32,12,1 4096,200,32,14,1 \
  4096,200,16,0,1 4096,200,16,8,1 4096,200,16,12,1 4096,200,16,16,1 4096,200,16,20,1 \
  4096,200,8,0,1 4096,200,8,16,1 \
  1024,200,64,0,0 1024,200,64,0,1 1024,200,64,7,1 1024,200,32,0,1 1024,200,32,12,1 1024,200,16,0,1 1024,200,16,16,1 1024,200,8,0,1 \
  512,200,64,0,0 512,200,64,0,1 512,200,32,0,1 512,200,16,0,1 512,200,8,0,1 \
  128,200,64,0,0 128,200,64,0,1 128,200,32,0,1 128,200,16,0,1 128,200,8,0,1 \
  400,2000,64,7,0 400,2000,64,7,1 400,2000,32,10,1 400,2000,16,16,1 \
  4096,200,64,7,0,1 4096,200,64,7,1,1 4096,200,16,16,1,1 \
  4096,200,64,8,0,0,1 4096,200,64,8,1,0,1 4096,200,32,12,1,0,1 4096,200,16,16,1,0,1 \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
echo "== callback trace"
timeout -k 10 120 python tools/trace_callback.py > $O/trace.out 2> $O/trace.err; echo "trace exit $?"; grep -A6 -- "---" $O/trace.err | head -40
echo "== gpu tests"
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest_gpu.log
;;
r02aa)
# compact pattern (46-entry slabs): resident-wave cap x tile size x fused, same box
O=gpurun_out/r02aa; mkdir -p $O
timeout -k 10 500 tools/bin/fgbench reps=60 pat=1 nt=1 xcd=1 \
  4096,200,64,8,0 4096,200,64,0,0 4096,200,64,10,0 4096,200,64,12,0 4096,200,64,16,0 \
  4096,200,64,8,1 4096,200,64,12,1 4096,200,64,0,1 \
  4096,200,32,0,0 4096,200,32,12,0 4096,200,32,16,0 4096,200,32,16,1 4096,200,40,12,0 4096,200,48,12,0 \
  nt=0 4096,200,64,8,0 4096,200,64,12,0 nt=1 \
  4096,200,64,8,0,0,1 4096,200,64,12,0,0,1 4096,200,64,16,0,0,1 4096,200,64,0,0,0,1 \
  400,2000,64,8,0 400,2000,64,12,0 400,2000,64,0,0 \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
;;
r02ab)
# fp32: resident-wave cap, reference and compact pattern, S10 / G7 / mixed, same box
O=gpurun_out/r02ab; mkdir -p $O
timeout -k 10 500 tools/bin/fgbench reps=60 nt=1 xcd=1 \
  4096,200,64,12,1,0,1 4096,200,64,16,1,0,1 4096,200,64,20,1,0,1 4096,200,64,0,1,0,1 4096,200,64,12,1,0,1 \
  4096,200,64,12,1,1,1 4096,200,64,0,1,1,1 8192,200,64,12,1,2,1 8192,200,64,0,1,2,1 \
  400,2000,64,12,1,0,1 400,2000,64,0,1,0,1 2048,200,64,12,1,0,1 2048,200,64,0,1,0,1 \
  pat=1 4096,200,64,12,0,0,1 4096,200,64,0,0,0,1 4096,200,64,0,1,0,1 8192,200,64,12,0,2,1 8192,200,64,0,0,2,1 8192,200,64,0,1,2,1 \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
;;
r02ac)
# s_setprio experiments: 1 = store phase at priority 3, 2 = load/compute phase at priority 3 (dropped to 0 for the stores)
O=gpurun_out/r02ac; mkdir -p $O
A="reps=60 nt=1 xcd=1 4096,200,64,8,1 400,2000,64,8,1 4096,200,64,12,1,0,1 8192,200,64,8,1,2 nt=0 1024,200,64,0,1"
{
for b in fgbench fgbench_prio1 fgbench_prio2 fgbench; do echo "== $b"; timeout -k 10 200 tools/bin/$b $A; done
} > $O/fgbench.md 2>&1
echo "exit $?"; grep -v "^|---\|^| B " $O/fgbench.md
;;
r02ad)
# do the XCDs finish their eighths at systematically different times?  (stamped build: timelines only)
O=gpurun_out/r02ad; mkdir -p $O
{
for rep in 1 2 3; do
  for a in "4096 200 20 0 8 1 1 1" "4096 200 20 0 8 0 1 1" "8192 200 10 0 8 1 1 1" "400 2000 20 0 8 1 1 1"; do
    echo "### fgprobe $a   (B N reps variant cap xcd fused nt)"
    timeout -k 10 120 tools/bin/fgprobe $a | grep "per XCC\|us/launch\|resident tile"
  done
done
} > $O/fgprobe.txt 2>&1
echo "exit $?"; cat $O/fgprobe.txt
;;
r02ae)
# is the odd/even XCD asymmetry a property of the XCD or of the eighth of the output it walks?  (stamped build: timelines only)
# variant 65536: XCD x walks eighth x^1;  131072: (x+4)%8;  262144: (x+2)%8
O=gpurun_out/r02ae; mkdir -p $O
{
for shape in "4096 200 20" "400 2000 20"; do
  for v in 0 65536 131072 262144 0; do
    echo "### fgprobe $shape $v 8 1 1 1   (B N reps variant cap xcd fused nt)"
    timeout -k 10 120 tools/bin/fgprobe $shape $v 8 1 1 1 | grep "per XCC\|us/launch"
  done
done
} > $O/fgprobe.txt 2>&1
echo "exit $?"; cat $O/fgprobe.txt
;;
r02ai)
# what attaching start/stop events to every dispatch costs: bench.py reports the uninstrumented timed region and the instrumented pass
O=gpurun_out/r02ai; mkdir -p $O
for a in "--batch 4096" "--batch 4096" "--batch 1024 --steps 500" "--batch 128 --steps 2000" "--batch 4096 --ts 2000 --aircraft skywalker --batch 400" "--batch 8192 --mission mixed" "--batch 4096 --dtype f32"; do
  timeout -k 10 200 python bench.py --no-configs --no-cpu-baseline $a > $O/b.json 2>/dev/null
  echo "bench.py $a"; python - <<PY
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("  uninstrumented: %.1f us/step wall, %.1f us/launch between two stream events = %.3f of peak | instrumented pass: %.1f us/step wall, dispatch events avg %.1f us (min %.1f) = %.3f of peak"
      % (1e3*d["ms_per_step"], 1e3*r["kernel_ms"], r["frac"], 1e3*r["instrumented_ms_per_step"], 1e3*r["instrumented_kernel_ms"], 1e3*r["instrumented_kernel_min_ms"],
         r["algorithmic_bytes_per_launch"]/(r["instrumented_kernel_ms"]*1e-3)/1e9/r["peak"]))
PY
done 2>&1 | tee $O/event_cost.txt
;;
r02aj)
# the launch-plan decisions again, timed without per-dispatch events: fused vs two launches, cap, store flavour, tile size
O=gpurun_out/r02aj; mkdir -p $O
timeout -k 10 800 tools/bin/fgbench reps=60 nt=1 xcd=1 \
  4096,200,64,8,1 4096,200,64,8,0 4096,200,64,0,1 4096,200,64,7,1 4096,200,64,10,1 4096,200,64,8,1 \
  xcd=0 4096,200,64,8,1 xcd=1 \
  400,2000,64,8,1 400,2000,64,8,0 400,2000,64,0,1 \
  4096,200,64,12,1,0,1 4096,200,64,12,0,0,1 4096,200,64,0,1,0,1 \
  8192,200,64,8,1,2 8192,200,64,8,0,2 \
  2048,200,64,8,1 2048,200,64,8,0 2048,200,64,0,1 nt=0 2048,200,64,0,1 2048,200,64,8,1 \
  nt=0 1024,200,64,0,1 1024,200,64,0,0 1024,200,48,0,1 1024,200,32,0,1 nt=1 1024,200,64,0,1 1024,200,64,8,1 \
  nt=0 512,200,64,0,1 512,200,64,0,0 512,200,32,0,1 128,200,64,0,1 128,200,64,0,0 128,200,32,0,1 \
  pat=1 nt=1 4096,200,64,8,1 4096,200,64,8,0 4096,200,64,0,1 4096,200,64,0,0 \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
;;
r02ak)
# phase shares and timeline of the compact pattern (stamped build: shares only) next to the reference pattern
O=gpurun_out/r02ak; mkdir -p $O
{
for a in "4096 200 20 0 8 1 1 1 0" "4096 200 20 0 8 1 0 1 1" "4096 200 20 0 0 1 0 1 1" "4096 200 20 0 8 1 1 1 1" "4096 200 20 256 8 1 0 1 1" "4096 200 20 2048 8 1 0 1 1"; do
  echo "### fgprobe $a   (B N reps variant cap xcd fused nt pattern)"
  timeout -k 10 120 tools/bin/fgprobe $a
done
} > $O/fgprobe.txt 2>&1
cat $O/fgprobe.txt
;;
r02al)
# compact pattern, fp64: resident-wave cap, uninstrumented time column
O=gpurun_out/r02al; mkdir -p $O
timeout -k 10 600 tools/bin/fgbench reps=60 pat=1 nt=1 xcd=1 \
  4096,200,64,8,0 4096,200,64,0,0 4096,200,64,12,0 4096,200,64,10,0 4096,200,64,8,0 4096,200,64,0,0 \
  400,2000,64,8,0 400,2000,64,0,0 8192,200,64,8,0,2 8192,200,64,0,0,2 4096,200,64,8,0,1 4096,200,64,0,0,1 \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
;;
r02an)
# how much would full 64-node tiles buy at ts = 200?  ts = 192 and 256 have them (3 / 4 tiles of 64), ts = 200 has 4 x 52
O=gpurun_out/r02an; mkdir -p $O
timeout -k 10 600 tools/bin/fgbench reps=60 nt=1 xcd=1 \
  4096,200,64,8,1 4266,192,64,8,1 3200,256,64,8,1 4096,200,64,8,1 4266,192,64,8,1 3200,256,64,8,1 \
  4096,200,64,12,1,0,1 4266,192,64,12,1,0,1 3200,256,64,12,1,0,1 \
  pat=1 4096,200,64,8,0 4266,192,64,8,0 3200,256,64,8,0 4096,200,64,0,0,0,1 4266,192,64,0,0,0,1 3200,256,64,0,0,0,1 \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
;;
r02ao)
# does the row stride of G (the spacing of the concurrent store fronts) matter?  ldgpad = extra elements between rows
O=gpurun_out/r02ao; mkdir -p $O
S=4096,200,64,8,1
timeout -k 10 700 tools/bin/fgbench reps=50 nt=1 xcd=1 \
  ldgpad=0 $S ldgpad=2 $S ldgpad=10 $S ldgpad=66 $S ldgpad=74 $S ldgpad=514 $S ldgpad=1090 $S ldgpad=2050 $S ldgpad=3138 $S ldgpad=11330 $S ldgpad=0 $S \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
;;
r02ap)
# cache-policy bits of the slab stores inside the fg kernel (TOLFG_STORE_FLAVOR builds): 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0 sc1 nt
O=gpurun_out/r02ap; mkdir -p $O
A="reps=50 nt=1 xcd=1 4096,200,64,8,1 400,2000,64,8,1"
{
for b in fgbench fgbench_fl1 fgbench_fl2 fgbench_fl3 fgbench_fl4 fgbench_fl5 fgbench; do echo "== $b"; timeout -k 10 200 tools/bin/$b $A | grep -v "^|---\|^| B "; done
} > $O/fgbench.md 2>&1
cat $O/fgbench.md
;;
r02aq)
# write-stream shapes, part 9: 52 KiB per wave written as 4 KiB chunks interleaved over a small group of G waves (mode 11), G = 4 ... 64
W=tools/bin/wrbench; O=gpurun_out/r02aq; mkdir -p $O
{
for rep in 1 2 3; do
  timeout -k 5 60 $W 4 52 23400
  for G in 4 8 16 64; do timeout -k 5 60 $W 11 52 23400 4 $G; done
  timeout -k 5 60 $W 11 52 23400 2 4
  timeout -k 5 60 $W 11 52 23400 13 4
done
} > $O/wrbench.txt 2>&1
cat $O/wrbench.txt
;;
r02as)
# do 128-byte-aligned slab regions matter?  ldgpad=2 makes the row stride a multiple of 128 B; goff=4 then puts every row's slab
# region (row + c0) on a 128-byte line, so that no line is shared between two tile waves
O=gpurun_out/r02as; mkdir -p $O
S=4096,200,64,8,1
timeout -k 10 700 tools/bin/fgbench reps=50 nt=1 xcd=1 \
  ldgpad=0 goff=0 $S ldgpad=2 goff=0 $S ldgpad=2 goff=4 $S ldgpad=2 goff=12 $S ldgpad=0 goff=0 $S ldgpad=2 goff=4 $S ldgpad=2 goff=0 $S ldgpad=2 goff=4 $S \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
;;
r02at)
# which 16-byte position of a row's slab region inside a 128-byte line is fast?  ldgpad=2: every row the same position; goff shifts it
O=gpurun_out/r02at; mkdir -p $O
S=4096,200,64,8,1
timeout -k 10 800 tools/bin/fgbench reps=50 nt=1 xcd=1 \
  ldgpad=0 goff=0 $S \
  ldgpad=2 goff=0 $S goff=2 $S goff=4 $S goff=6 $S goff=8 $S goff=10 $S goff=12 $S goff=14 $S \
  ldgpad=0 goff=0 $S \
  ldgpad=2 goff=0 $S goff=2 $S goff=4 $S goff=6 $S goff=8 $S goff=10 $S goff=12 $S goff=14 $S \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md | cut -d'|' -f11-13
;;
r02b)
# round-2 experiment B: cooperative write shapes (wrbench modes 7-10), fused variants
mkdir -p gpurun_out/r02b
O=gpurun_out/r02b
W=tools/bin/wrbench
{
echo "== mode 4 reference points"; $W 4 4 0; $W 4 52 23400; $W 4 52 0
echo "== mode 7: GRP waves interleave KiB chunks of one S KiB region (lds caps workgroups per CU)"
for lds in 0 23400 40000 80000; do for grp in 2 4 8; do $W 7 52 $lds $grp; done; done
for lds in 0 40000; do for grp in 2 4; do $W 7 26 $lds $grp; $W 7 13 $lds $grp; done; done
echo "== mode 8: GRP waves, wave w writes the w-th contiguous piece"
for lds in 0 40000 80000; do for grp in 4 8; do $W 8 52 $lds $grp; done; done
echo "== mode 9: idle (delay x 64 s_sleep(8)) then S KiB"
for d in 0 2 8 32; do $W 9 4 0 1 $d; $W 9 4 23400 1 $d; done
for d in 2 8; do $W 9 52 23400 1 $d; done
echo "== mode 10: mode 7 + 1 KiB read + idle per wave"
for d in 0 2 8; do for grp in 4 8; do $W 10 52 40000 $grp $d; $W 10 52 0 $grp $d; done; done
} > $O/wrbench.txt 2>&1
cat $O/wrbench.txt
echo "== fgbench fused variants"
timeout -k 10 300 tools/bin/fgbench reps=40 \
  4096,200,64,7,0 4096,200,64,7,1 4096,200,64,7,2 4096,200,64,8,2 4096,200,64,0,2 4096,200,64,7,0 4096,200,64,7,2 \
  1024,200,64,0,0 1024,200,64,0,1 1024,200,64,0,2 1024,200,32,0,2 1024,200,40,0,2 \
  512,200,64,0,2 512,200,32,0,2 128,200,64,0,2 \
  400,2000,64,7,0 400,2000,64,7,2 \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
;;
r02c)
# round-2 experiment C: XCD placement of write streams; plain vs non-temporal stores when the outputs fit the Infinity Cache
mkdir -p gpurun_out/r02c
O=gpurun_out/r02c
W=tools/bin/wrbench
{
echo "== XCD-contiguous (mode 3) vs launch-order (mode 4)"
for S in 4 13 52; do for lds in 0 23400; do $W 4 $S $lds; $W 3 $S $lds; done; done
echo "== plain stores, launch order (mode 0)"
for S in 4 52; do $W 0 $S 0; $W 0 $S 23400; done
} > $O/wrbench.txt 2>&1
cat $O/wrbench.txt
for bin in fgbench fgbench_plain; do
echo "== $bin"
timeout -k 10 300 tools/bin/$bin reps=40 \
  4096,200,64,7,0 4096,200,64,0,0 2048,200,64,7,0 2048,200,64,0,0 \
  1024,200,64,0,0 1024,200,64,7,0 1024,200,32,0,0 512,200,64,0,0 512,200,32,0,0 256,200,64,0,0 128,200,64,0,0 \
  > $O/$bin.md 2>&1
cat $O/$bin.md
done
;;
r02d)
# round-2 experiment D: XCD-contiguous tile order, nt vs plain slab stores, fused (polling) finalize, tile size
mkdir -p gpurun_out/r02d
O=gpurun_out/r02d
timeout -k 10 500 tools/bin/fgbench reps=40 \
  nt=1 xcd=0 4096,200,64,7,0 4096,200,64,7,1 4096,200,64,8,1 \
  xcd=1 4096,200,64,7,0 4096,200,64,7,1 4096,200,64,8,1 4096,200,64,0,1 4096,200,64,6,0 4096,200,64,8,0 4096,200,64,0,0 4096,200,32,12,0 4096,200,32,0,0 4096,200,16,0,0 \
  xcd=0 2048,200,64,7,0 xcd=1 2048,200,64,7,0 2048,200,64,7,1 2048,200,64,0,1 nt=0 2048,200,64,0,1 2048,200,64,7,1 \
  nt=0 xcd=0 1024,200,64,0,0 1024,200,64,0,1 1024,200,32,0,1 xcd=1 1024,200,64,0,0 1024,200,64,0,1 1024,200,32,0,1 1024,200,32,0,0 1024,200,40,0,1 1024,200,24,0,1 nt=1 1024,200,32,0,1 \
  nt=0 xcd=0 512,200,64,0,1 512,200,32,0,1 xcd=1 512,200,64,0,1 512,200,32,0,1 512,200,24,0,1 512,200,16,0,1 \
  xcd=0 128,200,64,0,1 128,200,32,0,1 xcd=1 128,200,64,0,1 128,200,32,0,1 128,200,16,0,1 \
  nt=1 xcd=0 400,2000,64,7,0 xcd=1 400,2000,64,7,0 400,2000,64,7,1 400,2000,64,8,1 \
  xcd=0 4096,200,64,7,0,1 xcd=1 4096,200,64,7,0,1 4096,200,64,7,1,1 \
  xcd=0 4096,200,64,8,0,0,1 xcd=1 4096,200,64,8,0,0,1 4096,200,64,8,1,0,1 4096,200,64,12,1,0,1 \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
timeout -k 10 300 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest_gpu.log
;;
r02e)
# round-2 experiment E: SNOPT-callback latency -- completion word, registered caller arrays, zero-copy limit
mkdir -p gpurun_out/r02e
O=gpurun_out/r02e
timeout -k 10 300 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest_gpu.log
echo "== default (flag + registered arrays)"; timeout -k 10 120 python tools/callback_tool.py rate 2>&1 | tail -6
echo "== TOLFG_NO_FLAG"; TOLFG_NO_FLAG=1 timeout -k 10 120 python tools/callback_tool.py rate 2>&1 | tail -6
echo "== TOLFG_NO_REGISTER"; TOLFG_NO_REGISTER=1 timeout -k 10 120 python tools/callback_tool.py rate 2>&1 | tail -6
echo "== TOLFG_NO_FLAG TOLFG_NO_REGISTER (round-1 behaviour)"; TOLFG_NO_FLAG=1 TOLFG_NO_REGISTER=1 timeout -k 10 120 python tools/callback_tool.py rate 2>&1 | tail -6
echo "== zero-copy limit 4 MB (ts=2000 direct)"; TOLFG_ZERO_COPY_LIMIT=4000000 timeout -k 10 120 python tools/callback_tool.py rate 2>&1 | tail -6
echo "== zero-copy limit 4 MB, nt stores"; TOLFG_NT_STORES=1 TOLFG_ZERO_COPY_LIMIT=4000000 timeout -k 10 120 python tools/callback_tool.py rate 2>&1 | tail -6
echo "== trace"; timeout -k 10 120 python tools/trace_callback.py > $O/trace.out 2> $O/trace.err; grep -A4 -- "---" $O/trace.err | grep -v amdgpu.ids | head -40
;;
r02f)
# round-2 experiment F: GPU suite after the mixed-mission / callback changes, callback rate, mixed + fp32-G7 timings
mkdir -p gpurun_out/r02f
O=gpurun_out/r02f
timeout -k 10 400 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -15 $O/pytest_gpu.log | cut -c1-300
echo "== callback (default)"; timeout -k 10 120 python tools/callback_tool.py rate 2>&1 | tail -5
echo "== callback TOLFG_NO_FLAG"; TOLFG_NO_FLAG=1 timeout -k 10 120 python tools/callback_tool.py rate 2>&1 | tail -5
timeout -k 10 300 tools/bin/fgbench reps=40 nt=1 xcd=1 \
  4096,200,64,7,1,1 4096,200,64,7,1,2 8192,200,64,7,1,2 \
  4096,200,64,8,1,1,1 4096,200,64,12,1,1,1 8192,200,64,8,1,2,1 8192,200,64,12,1,2,1 8192,200,64,8,1,0,1 \
  4096,200,64,7,1 4096,200,64,8,1 4096,200,64,7,1 4096,200,64,8,1 4096,200,64,6,1 4096,200,64,9,1 \
  > $O/fgbench.md 2>&1; echo "fgbench exit $?"; cat $O/fgbench.md
;;
r02g)
# round-2 experiment G: GPU suite, callback rate, bench.py (new layout)
mkdir -p gpurun_out/r02g
O=gpurun_out/r02g
timeout -k 10 600 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -12 $O/pytest_gpu.log | cut -c1-400
grep "worst scaled error per class" $O/pytest_gpu.log | cut -c1-1200
echo "== callback (default)"; timeout -k 10 120 python tools/callback_tool.py rate 2>&1 | tail -5
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; tail -3 $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02g/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','backend','scaling')}, d['roofline']['frac'], d['roofline']['kernel_ms'])
for r in d.get('configs',[]):
    print(r['config'], r['mode'], r.get('batch'), r.get('dtype'), 'ms/step %.4f'%r.get('ms_per_step',0) if 'ms_per_step' in r else '', 'eval_us %.1f'%r['eval_us'] if 'eval_us' in r else 'us/call %.1f'%r['us_per_call'], '%.3g node-evals/s'%r['node_evals_per_s'], 'frac %.3f'%r['frac_of_hbm_peak'] if 'frac_of_hbm_peak' in r else '')
print(d.get('cpu_baseline',{}).get('value'), d.get('next_compact_pattern',{}).get('frac_of_hbm_peak'))
PY
;;
r02h)
# round-2 experiment H: fast sincos -- GPU suite, tile sizes again, callback rate, bench configs
mkdir -p gpurun_out/r02h
O=gpurun_out/r02h
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -6 $O/pytest_gpu.log | cut -c1-400
echo "== callback (default)"; timeout -k 10 120 python tools/callback_tool.py rate 2 2>&1 | tail -10
timeout -k 10 400 tools/bin/fgbench reps=40 nt=1 xcd=1 \
  4096,200,64,8,1 4096,200,32,12,1 4096,200,32,0,1 4096,200,16,0,1 4096,200,16,16,1 4096,200,8,0,1 \
  nt=0 1024,200,64,0,1 1024,200,32,0,1 1024,200,16,0,1 512,200,64,0,1 512,200,32,0,1 512,200,16,0,1 128,200,64,0,1 128,200,32,0,1 128,200,16,0,1 128,200,8,0,1 \
  nt=1 8192,200,64,8,1,2 8192,200,64,12,1,2,1 400,2000,64,8,1 \
  > $O/fgbench.md 2>&1; echo "fgbench exit $?"; cat $O/fgbench.md
for v in "" "TOLFG_TILE_NODES=32" "TOLFG_XCD=0" "TOLFG_NT_STORES=1"; do
echo "== bench B=1024 $v"; env $v timeout -k 10 200 python bench.py --batch 1024 --steps 100 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('ms/step %.4f kernel %.4f min %.4f frac %.3f'%(d['ms_per_step'], r['kernel_ms'], r['kernel_min_ms'], r['frac']))"
done
;;
r02i)
# round-2 experiment I: GPU suite, native callback timing, bench, rocprofv3 stats + PMC passes
mkdir -p gpurun_out/r02i
O=gpurun_out/r02i
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest_gpu.log | cut -c1-400
echo "== callback"; timeout -k 10 120 python tools/callback_tool.py rate 2>&1 | tail -5
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; tail -2 $O/bench.err
python tools/show_bench.py $O/bench.json
timeout -k 10 600 bash tools/profile_gpu.sh r02 > $O/profile.log 2>&1; echo "profile exit $?"; tail -3 $O/profile.log
timeout -k 10 200 bash tools/pmc_pass.sh ta "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WRITE_WAVEFRONTS_sum" > $O/pmc_ta.txt 2>&1; echo "ta pass exit $?"; cat $O/pmc_ta.txt | tail -6
timeout -k 10 200 bash tools/pmc_pass.sh wr "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" > $O/pmc_wr.txt 2>&1; echo "wr pass exit $?"; cat $O/pmc_wr.txt | tail -6
;;
r02j)
# round-2 experiment J: GPU suite; compact-pattern launch choices; TA counters one per pass
mkdir -p gpurun_out/r02j
O=gpurun_out/r02j
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest_gpu.log | cut -c1-400
for v in "" "TOLFG_XCD=0" "TOLFG_FUSED=0" "TOLFG_WAVES_PER_CU=0" "TOLFG_WAVES_PER_CU=6" "TOLFG_WAVES_PER_CU=10" "TOLFG_NT_STORES=0" "TOLFG_XCD=0 TOLFG_FUSED=0"; do
echo "== compact $v"; env $v timeout -k 10 200 python bench.py --pattern compact --steps 100 --no-cpu-baseline --no-configs 2>/dev/null | python tools/show_bench.py | head -1
done
for c in TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_BUSY_avr; do
timeout -k 10 200 bash tools/pmc_pass.sh ta_$c "$c" > $O/pmc_$c.txt 2>&1; echo "pass $c exit $?"; tail -3 $O/pmc_$c.txt | cut -c1-200
done
;;
r02k)
# round-2 experiment K: compact pattern A/B (own vs library sin/cos, cap, xcd, fused)
mkdir -p gpurun_out/r02k
O=gpurun_out/r02k
for bin in fgbench fgbench_libsc; do
echo "== $bin"
timeout -k 10 300 tools/bin/$bin reps=40 pat=1 nt=1 xcd=0 4096,200,64,8,0 4096,200,64,0,0 xcd=1 4096,200,64,8,1 4096,200,64,0,1 4096,200,64,12,1 pat=0 xcd=1 4096,200,64,8,1 > $O/$bin.md 2>&1
cat $O/$bin.md
done
;;
r02l)
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
;;
r02m)
# round-2 experiment M: shifted slab stream -- GPU suite, fp32 / G7 / odd-ts timings
mkdir -p gpurun_out/r02m
O=gpurun_out/r02m
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest_gpu.log | cut -c1-400
timeout -k 10 300 tools/bin/fgbench reps=40 nt=1 xcd=1 \
  4096,200,64,8,1 4096,200,64,8,1,1 4096,200,64,12,1,0,1 4096,200,64,12,1,1,1 8192,200,64,12,1,2,1 8192,200,64,8,1,2 4096,201,64,8,1 4096,201,64,12,1,1,1 \
  > $O/fgbench.md 2>&1; echo "fgbench exit $?"; cat $O/fgbench.md
;;
r02n)
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
;;
r02o)
# round-2 experiment O: LDS sized to the tile (11 waves per CU uncapped at ts=200); where the ~6 us between evaluations go
mkdir -p gpurun_out/r02o
O=gpurun_out/r02o
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log | cut -c1-300
for bin in fgbench fgbench_ntsmall; do
echo "== $bin"
timeout -k 10 300 tools/bin/$bin reps=40 nt=1 xcd=1 4096,200,64,8,1 4096,200,64,8,1 nt=0 1024,200,64,0,1 1024,200,64,10,1 1024,200,64,9,1 512,200,64,0,1 512,200,64,9,1 256,200,64,0,1 > $O/$bin.md 2>&1; cat $O/$bin.md
done
;;
r02p)
# round-2 experiment P: where a launch's time goes at B = 128 ... 4096 (stamped build: shares and timeline, not benchmark numbers)
mkdir -p gpurun_out/r02p
O=gpurun_out/r02p
{
for cfg in "4096 200 20 0 8 1 1 1" "2048 200 20 0 8 1 1 1" "1024 200 20 0 0 1 1 0" "512 200 20 0 0 1 1 0" "256 200 20 0 0 1 1 0" "128 200 20 0 0 1 1 0" "1 200 20 0 0 1 1 0" "400 2000 20 0 8 1 1 1"; do
echo "### fgprobe $cfg   (B N reps variant cap xcd fused nt)"
timeout -k 5 60 tools/bin/fgprobe $cfg
echo
done
} > $O/fgprobe.txt 2>&1
cat $O/fgprobe.txt
;;
r02q)
# round-2 experiment Q: occupancy over time inside a launch (stamped build)
mkdir -p gpurun_out/r02q
O=gpurun_out/r02q
{
for cfg in "4096 200 20 0 8 1 1 1" "4096 200 20 0 8 0 1 1" "4096 200 20 0 8 1 0 1" "4096 200 20 0 0 1 1 1" "4096 200 20 0 12 1 1 1" "1024 200 20 0 0 1 1 0"; do
echo "### fgprobe $cfg   (B N reps variant cap xcd fused nt)"
timeout -k 5 60 tools/bin/fgprobe $cfg | grep -v "cycles  "
echo
done
} > $O/fgprobe.txt 2>&1
cat $O/fgprobe.txt
;;
r02r)
# round-2 experiment R: persistent workgroups with per-XCD tile queues vs one workgroup per tile
mkdir -p gpurun_out/r02r
O=gpurun_out/r02r
timeout -k 10 300 tools/bin/fgbench reps=40 nt=1 xcd=1 \
  persist=0 4096,200,64,8,1 persist=8 4096,200,64,8,1 persist=7 4096,200,64,7,1 persist=6 4096,200,64,6,1 persist=5 4096,200,64,5,1 persist=10 4096,200,64,10,1 persist=6 4096,200,64,8,1 \
  persist=0 4096,200,64,8,1 persist=6 4096,200,64,6,1 persist=7 4096,200,64,7,1 \
  persist=0 400,2000,64,8,1 persist=6 400,2000,64,6,1 persist=7 400,2000,64,7,1 \
  persist=0 8192,200,64,8,1,2 persist=6 8192,200,64,6,1,2 persist=7 8192,200,64,7,1,2 \
  persist=0 4096,200,64,12,1,0,1 persist=8 4096,200,64,8,1,0,1 persist=10 4096,200,64,10,1,0,1 persist=12 4096,200,64,12,1,0,1 \
  persist=0 2048,200,64,8,1 persist=6 2048,200,64,6,1 persist=7 2048,200,64,7,1 \
  > $O/fgbench.md 2>&1; echo "fgbench exit $?"; cat $O/fgbench.md
;;
r02s)
# round-2 experiment S: deal only part of the tiles XCD-contiguously, the launch's tail in id order (all XCDs share it)
mkdir -p gpurun_out/r02s
O=gpurun_out/r02s
timeout -k 10 300 tools/bin/fgbench reps=40 nt=1 xcd=1 \
  xcdpct=100 4096,200,64,8,1 xcdpct=95 4096,200,64,8,1 xcdpct=90 4096,200,64,8,1 xcdpct=85 4096,200,64,8,1 xcdpct=75 4096,200,64,8,1 xcdpct=50 4096,200,64,8,1 xcdpct=0 4096,200,64,8,1 \
  xcdpct=100 4096,200,64,8,1 xcdpct=90 4096,200,64,8,1 xcdpct=85 4096,200,64,8,1 \
  xcdpct=100 400,2000,64,8,1 xcdpct=90 400,2000,64,8,1 xcdpct=80 400,2000,64,8,1 \
  xcdpct=100 2048,200,64,8,1 xcdpct=85 2048,200,64,8,1 xcdpct=70 2048,200,64,8,1 \
  xcdpct=100 4096,200,64,12,1,0,1 xcdpct=90 4096,200,64,12,1,0,1 xcdpct=80 4096,200,64,12,1,0,1 \
  nt=0 xcdpct=100 1024,200,64,0,1 xcdpct=70 1024,200,64,0,1 xcdpct=50 1024,200,64,0,1 xcdpct=0 1024,200,64,0,1 \
  > $O/fgbench.md 2>&1; echo "fgbench exit $?"; cat $O/fgbench.md
;;
r02t)
# round-2 experiment T: final build -- GPU suite, callback timings incl. F-only calls
mkdir -p gpurun_out/r02t
O=gpurun_out/r02t
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; python tools/show_bench.py $O/bench.json
;;
r02u)
# round-2 experiment U: own fp32 sin/cos -- GPU suite, fp32 timings against the library routine (same box)
mkdir -p gpurun_out/r02u
O=gpurun_out/r02u
timeout -k 10 600 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log | cut -c1-300; grep "worst scaled error per class" $O/pytest_gpu.log | cut -c1-1500
for rep in 1 2; do for bin in fgbench fgbench_libsc; do
echo "== $bin (rep $rep)"
timeout -k 10 300 tools/bin/$bin reps=40 nt=1 xcd=1 4096,200,64,12,1,0,1 4096,200,64,12,1,1,1 8192,200,64,12,1,2,1 4096,200,16,0,1,0,1 nt=0 1024,200,64,0,1,0,1 2>&1 | tail -5
done; done
;;
r02v)
# round-2 experiment V: compact pattern -- persistent form and caps (fused / unfused)
mkdir -p gpurun_out/r02v
O=gpurun_out/r02v
timeout -k 10 300 tools/bin/fgbench_persist reps=40 pat=1 nt=1 xcd=1 \
  persist=0 4096,200,64,8,0 4096,200,64,0,0 4096,200,64,8,1 \
  persist=8 4096,200,64,8,1 persist=10 4096,200,64,10,1 persist=6 4096,200,64,6,1 persist=12 4096,200,64,0,1 \
  persist=0 4096,200,64,12,0,0,1 persist=12 4096,200,64,12,1,0,1 persist=16 4096,200,64,16,1,0,1 \
  > $O/fgbench.md 2>&1; echo "exit $?"; cat $O/fgbench.md
;;
r02w)
# round-2 experiment W: bench.py with the two-streams side record
mkdir -p gpurun_out/r02w
timeout -k 10 400 python bench.py --steps 100 --no-cpu-baseline > gpurun_out/r02w/bench.json 2> gpurun_out/r02w/bench.err; echo "bench exit $?"; tail -3 gpurun_out/r02w/bench.err; python tools/show_bench.py gpurun_out/r02w/bench.json | tail -4
;;
r02x)
# write-stream shapes, part 4: is it the LENGTH of a wave's stream or the compactness of the in-flight address window?
set -e
W=tools/bin/wrbench; O=gpurun_out/r02x; mkdir -p $O
{
echo "== reference points (mode 4)"
for S in 4 52; do timeout -k 5 60 $W 4 $S 23400; done
echo "== mode 13: short streams, segments dealt in scattered order"
for S in 4 8 52; do timeout -k 5 60 $W 13 $S 23400; done
echo "== mode 11: long-lived waves (52 KiB each), chunks interleaved over a super-group of G waves (grp = chunk KiB, delay = G, 0 = all)"
for G in 0 16384 2048 256 16; do for C in 1 4; do timeout -k 5 60 $W 11 52 23400 $C $G; done; done
for G in 0 2048; do timeout -k 5 60 $W 11 52 0 4 $G; done
} > $O/wrbench.txt 2>&1
cat $O/wrbench.txt
;;
r02x2)
# write-stream shapes, part 5: one store instruction spread over several places of the wave's region
W=tools/bin/wrbench; O=gpurun_out/r02x2; mkdir -p $O
{
echo "== reference points (mode 4)"
for S in 4 52; do timeout -k 5 60 $W 4 $S 23400; done
echo "== mode 14: grp = lanes per contiguous piece (64 = mode 4)"
for S in 52 48 32 16; do for LG in 64 32 16 8 4 2; do timeout -k 5 60 $W 14 $S 23400 $LG; done; done
timeout -k 5 60 $W 4 52 23400
} > $O/wrbench.txt 2>&1
cat $O/wrbench.txt
;;
r02x3)
# write-stream shapes, part 6: cache-policy bits of the 16-byte stores
W=tools/bin/wrbench; O=gpurun_out/r02x3; mkdir -p $O
{
for S in 52 4; do
  echo "== S = $S KiB per wave; policy 0 none, 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1, 5 sc0 nt, 6 sc1 nt, 7 sc0 sc1 nt"
  timeout -k 5 60 $W 4 $S 23400
  for P in 0 1 2 3 4 5 6 7; do timeout -k 5 60 $W 15 $S 23400 $P; done
done
} > $O/wrbench.txt 2>&1
cat $O/wrbench.txt
;;
r02y)
# finer tiles for the trajectories a launch reaches last (FgArgs::tail_count): same-box A/B with tools/fgbench
O=gpurun_out/r02y; mkdir -p $O
S=4096,200,64,8,1
timeout -k 10 500 tools/bin/fgbench reps=60 nt=1 xcd=1 \
  tail=0 $S tail=128:16 $S tail=256:16 $S tail=512:16 $S tail=1024:16 $S \
  tail=0 $S tail=128:32 $S tail=256:32 $S tail=512:32 $S tail=1024:32 $S \
  tail=0 $S tail=256:8 $S tail=256:24 $S tail=512:24 $S tail=4096:32 $S \
  tail=0 400,2000,64,8,1 tail=16:16 400,2000,64,8,1 tail=32:32 400,2000,64,8,1 tail=64:32 400,2000,64,8,1 \
  tail=0 4096,200,64,12,1,0,1 tail=256:16 4096,200,64,12,1,0,1 tail=512:32 4096,200,64,12,1,0,1 \
  nt=0 tail=0 1024,200,64,0,1 tail=128:16 1024,200,64,0,1 tail=256:32 1024,200,64,0,1 tail=128:32 1024,200,64,0,1 \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
;;
r02z)
# finer-tiled tail at the batch sizes whose outputs fit the Infinity Cache (plain stores, no cap) and at B=2048
O=gpurun_out/r02z; mkdir -p $O
timeout -k 10 500 tools/bin/fgbench reps=80 nt=0 xcd=1 \
  tail=0 1024,200,64,0,1 tail=256:32 1024,200,64,0,1 tail=384:32 1024,200,64,0,1 tail=512:32 1024,200,64,0,1 tail=1024:32 1024,200,64,0,1 \
  tail=256:40 1024,200,64,0,1 tail=512:40 1024,200,64,0,1 tail=1024:40 1024,200,64,0,1 tail=0 1024,200,64,0,1 \
  tail=0 512,200,64,0,1 tail=128:32 512,200,64,0,1 tail=256:32 512,200,64,0,1 tail=512:32 512,200,64,0,1 tail=256:16 512,200,64,0,1 \
  tail=0 128,200,64,0,1 tail=64:32 128,200,64,0,1 tail=128:32 128,200,64,0,1 tail=128:16 128,200,64,0,1 \
  tail=0 256,200,64,0,1 tail=128:32 256,200,64,0,1 tail=256:32 256,200,64,0,1 \
  nt=1 tail=0 2048,200,64,8,1 tail=128:32 2048,200,64,8,1 tail=256:32 2048,200,64,8,1 \
  nt=0 tail=0 2048,200,64,0,1 tail=256:32 2048,200,64,0,1 \
  nt=0 tail=0 1024,200,64,0,1,2 tail=256:32 1024,200,64,0,1,2 tail=0 1024,200,64,0,1,0,1 tail=256:32 1024,200,64,0,1,0,1 \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
;;
r03a)
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
;;
r03b)
# dynamic instruction mix and wait states: fp32 mixed reference pattern, fp64 compact, fp64 reference
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > gpurun_out/sq_counters.txt
bash tools/pmc_fgbench.sh r03_f32ref tools/bin/fgbench "reps=4 nt=1 xcd=1 8192,200,64,12,1,2,1"
bash tools/pmc_fgbench.sh r03_f64cmp tools/bin/fgbench "reps=4 nt=1 xcd=1 pat=1 4096,200,64,8,0,0,0"
bash tools/pmc_fgbench.sh r03_f64ref tools/bin/fgbench "reps=4 nt=1 xcd=1 4096,200,64,8,1,0,0"
;;
r03c)
# packed fp32 (two nodes per lane, 128-node tiles) vs one node per lane, same box
O=gpurun_out/r03c; mkdir -p $O
timeout -k 10 300 tools/bin/fgbench reps=60 nt=1 xcd=1 \
  8192,200,64,12,1,2,1 8192,200,128,8,1,2,1 8192,200,128,0,1,2,1 8192,200,128,6,1,2,1 \
  4096,200,64,12,1,0,1 4096,200,128,8,1,0,1 4096,200,128,0,1,0,1 \
  4096,200,64,12,1,1,1 4096,200,128,8,1,1,1 \
  400,2000,64,12,1,0,1 400,2000,128,8,1,0,1 \
  nt=0 1024,200,64,0,1,2,1 1024,200,128,0,1,2,1 1024,200,64,0,1,0,1 1024,200,128,0,1,0,1 128,200,64,0,1,0,1 128,200,128,0,1,0,1 \
  nt=1 pat=1 4096,200,64,0,1,0,1 4096,200,128,0,1,0,1 4096,200,128,0,0,0,1 8192,200,64,0,0,2,1 8192,200,128,0,0,2,1 8192,200,128,0,1,2,1 \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
;;
r03d)
# where does an fp32 launch's time go: ablations (results wrong by construction; MISMATCH expected), same box
O=gpurun_out/r03d; mkdir -p $O
A=tools/bin/fgbench_abl
for cfg in "8192,200,64,12,1,2,1" "8192,200,128,8,1,2,1" "4096,200,64,8,1,0,0"; do
  for v in 0 256 512 1024 1536 2048 4096 3584 3840; do
    timeout -k 10 60 $A reps=40 nt=1 xcd=1 variant=$v $cfg 2>/dev/null | tail -1 | sed "s/^/| variant $v /" >> $O/ablate.md || exit 1
  done
done
cat $O/ablate.md | cut -d'|' -f2-8,11-14
;;
r03e)
# tile size vs batch size for launches that fit the cache (plain stores, no cap): what should plan_launch choose?
O=gpurun_out/r03e; mkdir -p $O
F=tools/bin/fgbench
{
for B in 64 128 256 512 1024 2048; do
  timeout -k 10 120 $F reps=80 nt=0 xcd=1 $B,200,64,0,1,0,0 $B,200,52,0,1,0,0 $B,200,40,0,1,0,0 $B,200,36,0,1,0,0 $B,200,28,0,1,0,0 $B,200,20,0,1,0,0 $B,200,16,0,1,0,0 $B,200,12,0,1,0,0 $B,200,8,0,1,0,0 | tail -9 || exit 1
done
for B in 128 256 1024 2048; do
  timeout -k 10 120 $F reps=80 nt=0 xcd=1 $B,200,128,0,1,2,1 $B,200,64,0,1,2,1 $B,200,40,0,1,2,1 $B,200,28,0,1,2,1 $B,200,20,0,1,2,1 $B,200,16,0,1,2,1 $B,200,8,0,1,2,1 | tail -7 || exit 1
done
timeout -k 10 120 $F reps=80 nt=0 xcd=1 1024,200,64,0,1,2,0 1024,200,40,0,1,2,0 1024,200,28,0,1,2,0 50,2000,64,0,1,0,0 50,2000,32,0,1,0,0 50,2000,16,0,1,0,0 | tail -6
} > $O/tiles.md 2>&1
cut -d'|' -f2,3,4,5,6,8,11,13,14 $O/tiles.md
;;
r03f)
# pipelined stream loop (vs exp_r03e on another box: compare within this call only) and issue-priority stagger for launches within the cache
O=gpurun_out/r03f; mkdir -p $O
F=tools/bin/fgbench
{
for st in 0 1; do
timeout -k 10 200 $F reps=80 nt=0 xcd=1 stagger=$st 64,200,64,0,1,0,0 128,200,64,0,1,0,0 256,200,64,0,1,0,0 512,200,64,0,1,0,0 1024,200,64,0,1,0,0 2048,200,64,0,1,0,0 1024,200,64,0,1,2,0 \
   128,200,64,0,1,2,1 1024,200,64,0,1,2,1 1024,200,128,0,1,2,1 2048,200,64,0,1,2,1 2048,200,128,0,1,2,1 50,2000,64,0,1,0,0 \
   nt=1 4096,200,64,8,1,0,0 8192,200,64,12,1,2,1 8192,200,128,8,1,2,1 | tail -16 | sed "s/^/| stagger=$st /" || exit 1
done
} > $O/stagger.md 2>&1
cut -d'|' -f2,3,4,5,6,7,8,9,12,14,15 $O/stagger.md
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest_gpu.log
;;
r03h)
# the restructured bench.py: default 1-GPU line, then the 2-rank gloo rehearsal on one GPU
O=gpurun_out/r03h; mkdir -p $O
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; tail -3 $O/bench.err; python tools/show_bench.py $O/bench.json
bash tools/rehearse_ranks.sh > $O/rehearse.log 2>&1; echo "rehearse exit $?"; cat $O/rehearse.log | cut -c1-330
;;
r03l)
# compact pattern: where does a launch's time go (ablations; results wrong by construction), fp64 and fp32, same box
O=gpurun_out/r03l; mkdir -p $O
A=tools/bin/fgbench_abl
for cfg in "4096,200,64,8,0,0,0" "4096,200,64,0,0,0,1"; do
  for v in 0 256 512 1024 1536 2048 4096 3584 3840; do
    timeout -k 10 60 $A reps=40 nt=1 xcd=1 pat=1 variant=$v $cfg 2>/dev/null | tail -1 | sed "s/^/| variant $v /" >> $O/ablate.md || exit 1
  done
done
cut -d'|' -f2-8,11-14 $O/ablate.md
;;
r03m)
# x-window prefetch by leaving waves (FgArgs::prefetch = tiles ahead on the XCD's run): distance sweep, same box
O=gpurun_out/r03m; mkdir -p $O
F=tools/bin/fgbench
{
for pf in 0 128 256 320 384 512 768 0; do
  timeout -k 10 120 $F reps=60 nt=1 xcd=1 prefetch=$pf pat=1 4096,200,64,8,0,0,0 4096,200,64,0,0,0,1 pat=0 4096,200,64,8,1,0,0 8192,200,64,8,1,2,0 8192,200,64,12,1,2,1 8192,200,128,8,1,2,1 | tail -6 | sed "s/^/| pf=$pf /" || exit 1
done
} > $O/prefetch.md 2>&1
cut -d'|' -f2,3,4,5,6,7,8,9,12,14,15 $O/prefetch.md
;;
r03n)
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
;;
r03o)
# launches within the cache: resident-wave cap x issue-priority stagger (B = 1024 / 2048, fp64 and fp32), same box
O=gpurun_out/r03o; mkdir -p $O
F=tools/bin/fgbench
{
for st in 0 1; do
timeout -k 10 200 $F reps=100 nt=0 xcd=1 stagger=$st 1024,200,64,0,1,0,0 1024,200,64,8,1,0,0 1024,200,64,6,1,0,0 1024,200,64,5,1,0,0 1024,200,64,4,1,0,0 1024,200,64,0,1,0,0 \
   1024,200,64,0,1,2,0 1024,200,64,8,1,2,0 1024,200,64,0,1,2,1 1024,200,64,8,1,2,1 1024,200,64,12,1,2,1 1024,200,128,0,1,2,1 1024,200,128,4,1,2,1 \
   2048,200,64,0,1,0,0 2048,200,64,8,1,0,0 | tail -15 | sed "s/^/| stagger=$st /" || exit 1
done
} > $O/cap_stagger.md 2>&1
cut -d'|' -f2,3,4,5,6,7,8,9,12,14,15 $O/cap_stagger.md
;;
r03p)
# stagger on/off alternating on the SAME buffers (one process per shape), three fresh processes each: is the gain real or allocation luck?
O=gpurun_out/r03p; mkdir -p $O
F=tools/bin/fgbench
{
for rep in 1 2 3; do
for shape in 1024,200,64,0,1,0,0 1024,200,64,0,1,2,0 1024,200,128,0,1,2,1 2048,200,64,0,1,2,1 1536,200,64,0,1,0,0; do
timeout -k 10 100 $F reps=100 nt=0 xcd=1 stagger=0 $shape stagger=1 $shape stagger=0 $shape stagger=1 $shape stagger=0 $shape stagger=1 $shape | tail -6 | cut -d'|' -f2,4,5,6,11 | tr '\n' ' ' || exit 1
echo
done
done
} > $O/alt.md 2>&1
cat $O/alt.md
;;
r03q)
# round-3 full pass: GPU suite in its three forms, smoke, bench.py, shape sweep
O=gpurun_out/r03q; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
TOLFG_NO_SINGLE_LAUNCH=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_tiled_callback.log 2>&1; echo "pytest (callback through the tile-per-workgroup path) exit $?"; tail -2 $O/pytest_gpu_tiled_callback.log
TOLFG_FUSED=0 timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_two_launch.log 2>&1; echo "pytest (two-launch form) exit $?"; tail -2 $O/pytest_gpu_two_launch.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $O/smoke.log
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; python tools/show_bench.py $O/bench.json
timeout -k 10 600 bash tools/shape_sweep.sh > $O/shape_sweep.md 2>&1; echo "shape sweep exit $?"; cat $O/shape_sweep.md
;;
r03r)
# stream-table loads issued before the x window (one memory latency less per wave): previous build vs this one, alternating processes
O=gpurun_out/r03r; mkdir -p $O
{
for rep in 1 2; do
for exe in fgbench_prev fgbench; do
  timeout -k 10 200 tools/bin/$exe reps=80 nt=0 xcd=1 stagger=1 64,200,64,0,1,0,0 128,200,64,0,1,0,0 512,200,64,0,1,0,0 1024,200,64,0,1,0,0 1024,200,64,0,1,2,0 1024,200,128,0,1,2,1 2048,200,64,0,1,2,1 \
     stagger=0 nt=1 4096,200,64,8,1,0,0 8192,200,64,8,1,2,0 8192,200,128,8,1,2,1 pat=1 4096,200,64,8,0,0,0 | tail -11 | cut -d'|' -f2,4,5,6,7,11,13 | sed "s/^/| $exe /" || exit 1
done
done
} > $O/ab.md 2>&1
cat $O/ab.md
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
;;
r03s)
# with the shorter front end (table loads before the window), re-sweep the resident-wave cap of the launches beyond the cache
O=gpurun_out/r03s; mkdir -p $O
F=tools/bin/fgbench
{
timeout -k 10 300 $F reps=60 nt=1 xcd=1 4096,200,64,6,1,0,0 4096,200,64,7,1,0,0 4096,200,64,8,1,0,0 4096,200,64,9,1,0,0 4096,200,64,10,1,0,0 \
   8192,200,64,6,1,2,0 8192,200,64,7,1,2,0 8192,200,64,8,1,2,0 8192,200,64,10,1,2,0 \
   8192,200,128,5,1,2,1 8192,200,128,6,1,2,1 8192,200,128,7,1,2,1 8192,200,128,8,1,2,1 \
   8192,200,64,8,1,2,1 8192,200,64,10,1,2,1 8192,200,64,12,1,2,1 8192,200,64,14,1,2,1 \
   4096,200,64,8,1,0,1 4096,200,64,10,1,0,1 4096,200,64,12,1,0,1 4096,200,64,16,1,0,1 | tail -21 | cut -d'|' -f2,4,5,6,7,11,13
timeout -k 10 300 $F reps=60 nt=1 xcd=1 pat=1 4096,200,64,5,0,0,0 4096,200,64,6,0,0,0 4096,200,64,7,0,0,0 4096,200,64,8,0,0,0 4096,200,64,10,0,0,0 4096,200,64,0,0,0,0 4096,200,64,8,1,0,0 4096,200,64,6,1,0,0 \
   4096,200,64,0,0,0,1 4096,200,64,8,0,0,1 4096,200,64,12,0,0,1 | tail -11 | cut -d'|' -f2,4,5,6,7,8,11,13
} > $O/caps.md 2>&1
cat $O/caps.md
;;
r03t)
# table loads before (variant 0) vs after (variant 8192) the x window: the SAME buffers, alternating, ablation build (both orders compiled in)
O=gpurun_out/r03t; mkdir -p $O
A=tools/bin/fgbench_abl
{
for shape in "nt=0 128,200,64,0,1,0,0" "nt=0 1024,200,64,0,1,0,0" "nt=0 1024,200,128,0,1,2,1" "nt=1 4096,200,64,8,1,0,0" "nt=1 8192,200,64,8,1,2,0" "nt=1 8192,200,128,8,1,2,1" "nt=1 8192,200,64,12,1,2,1" "nt=1 pat=1 4096,200,64,8,0,0,0" "nt=1 pat=1 4096,200,64,0,0,0,1"; do
  set -- $shape
  last=${@: -1}; opts=${@:1:$#-1}
  timeout -k 10 200 $A reps=60 xcd=1 $opts variant=0 $last variant=8192 $last variant=0 $last variant=8192 $last variant=0 $last variant=8192 $last 2>/dev/null | tail -6 | cut -d'|' -f2,4,5,6,11 | tr '\n' ' ' || exit 1
  echo
done
} > $O/order.md 2>&1
cat $O/order.md
;;
r03u)
# slab stream: 1 / 2 / 3 rounds of LDS reads in flight ahead of the store (separate binaries: compare the small-B rows, where one wave
# is alone on its SIMD); and a pure store loop at low occupancy (wrbench) for the per-wave store rate
O=gpurun_out/r03u; mkdir -p $O
{
for rep in 1 2; do
for d in 1 2 3; do
  timeout -k 10 120 tools/bin/fgbench_a$d reps=200 nt=0 xcd=1 1,200,64,0,1,0,0 64,200,64,0,1,0,0 128,200,64,0,1,0,0 256,200,64,0,1,0,0 1024,200,64,0,1,0,0 nt=1 4096,200,64,8,1,0,0 | tail -6 | cut -d'|' -f2,4,5,11 | tr '\n' ' ' | sed "s/^/ahead=$d /" || exit 1
  echo
done
done
for lds in 65536 40960 20480 0; do timeout -k 5 60 tools/bin/wrbench 4 43 $lds; timeout -k 5 60 tools/bin/wrbench 0 43 $lds; done
} > $O/ahead.md 2>&1
cat $O/ahead.md
;;
r03v)
# compiler scheduling strategy: default (max occupancy) vs -mllvm -amdgpu-sched-strategy=max-ilp (separate binaries, two passes)
O=gpurun_out/r03v; mkdir -p $O
{
for rep in 1 2; do
for exe in fgbench fgbench_ilp; do
  timeout -k 10 120 tools/bin/$exe reps=200 nt=0 xcd=1 1,200,64,0,1,0,0 64,200,64,0,1,0,0 128,200,64,0,1,0,0 512,200,64,0,1,0,0 1024,200,64,0,1,0,0 1024,200,128,0,1,2,1 nt=1 4096,200,64,8,1,0,0 8192,200,128,8,1,2,1 | tail -8 | cut -d'|' -f2,4,5,11 | tr '\n' ' ' | sed "s/^/$exe /" || exit 1
  echo
done
done
} > $O/ilp.md 2>&1
cat $O/ilp.md
;;
r03w)
# finalize reads its 23 edge values at the wave's start (variant 0) vs at the end (variant 16384): same buffers, alternating; then the GPU suite
O=gpurun_out/r03w; mkdir -p $O
A=tools/bin/fgbench_abl
{
for shape in "nt=0 8,2000,64,0,1,0,0" "nt=0 64,200,64,0,1,0,0" "nt=0 128,200,64,0,1,0,0" "nt=0 256,200,64,0,1,0,0" "nt=0 1024,200,64,0,1,0,0" "nt=0 1024,200,128,0,1,2,1" "nt=1 4096,200,64,8,1,0,0" "nt=1 8192,200,64,8,1,2,0" "nt=1 8192,200,128,8,1,2,1"; do
  set -- $shape
  last=${@: -1}; opts=${@:1:$#-1}
  timeout -k 10 200 $A reps=100 xcd=1 $opts variant=0 $last variant=16384 $last variant=0 $last variant=16384 $last variant=0 $last variant=16384 $last 2>/dev/null | tail -6 | cut -d'|' -f2,3,4,5,11 | tr '\n' ' ' || exit 1
  echo
done
} > $O/edge.md 2>&1
cat $O/edge.md
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
;;
r03x)
# what do the per-wave loads of the stream table cost?  ablation: offsets made up in registers (variant 32768, results wrong) vs loaded (0); same buffers
O=gpurun_out/r03x; mkdir -p $O
A=tools/bin/fgbench_abl
{
for shape in "nt=0 128,200,64,0,1,0,0" "nt=0 1024,200,64,0,1,0,0" "nt=1 4096,200,64,8,1,0,0" "nt=1 8192,200,64,8,1,2,0" "nt=1 8192,200,128,8,1,2,1" "nt=1 8192,200,64,12,1,2,1" "nt=1 pat=1 4096,200,64,8,0,0,0"; do
  set -- $shape
  last=${@: -1}; opts=${@:1:$#-1}
  timeout -k 10 200 $A reps=100 xcd=1 $opts variant=0 $last variant=32768 $last variant=0 $last variant=32768 $last 2>/dev/null | tail -4 | cut -d'|' -f2,3,4,5,11 | tr '\n' ' ' || exit 1
  echo
done
} > $O/table_cost.md 2>&1
cat $O/table_cost.md
;;
r03y)
# dt = x[0] through a scalar load (variant 0) vs the vector load of rounds 1-2 (variant 65536): same buffers, alternating
O=gpurun_out/r03y; mkdir -p $O
A=tools/bin/fgbench_abl
{
for shape in "nt=0 128,200,64,0,1,0,0" "nt=0 1024,200,64,0,1,0,0" "nt=0 1024,200,128,0,1,2,1" "nt=1 4096,200,64,8,1,0,0" "nt=1 8192,200,64,8,1,2,0" "nt=1 8192,200,128,8,1,2,1" "nt=1 4096,200,64,12,1,0,1" "nt=1 pat=1 4096,200,64,8,0,0,0"; do
  set -- $shape
  last=${@: -1}; opts=${@:1:$#-1}
  timeout -k 10 200 $A reps=100 xcd=1 $opts variant=0 $last variant=65536 $last variant=0 $last variant=65536 $last variant=0 $last variant=65536 $last 2>/dev/null | tail -6 | cut -d'|' -f2,3,4,5,11,14 | tr '\n' ' ' || exit 1
  echo
done
} > $O/dt.md 2>&1
cat $O/dt.md
;;
r04_callback_tiles)
# VERDICT r3 task 6: the callback (B = 1) as ONE workgroup of 2-4 waves (fg_single_kernel, TOLFG_FORCE_SINGLE_LAUNCH=1) against the
# plan's choice since round 4 -- 5 tile workgroups on 5 CUs with the completion word -- and other tile sizes.
# Run on the GPU box from the repo root.
for rep in 1 2; do
echo "== pass $rep: one workgroup per trajectory (TOLFG_FORCE_SINGLE_LAUNCH=1)"
TOLFG_FORCE_SINGLE_LAUNCH=1 python3 tools/callback_tool.py rate | grep " 200 \| 100 "
echo "== pass $rep: the plan (5 tiles from ts = 100)"
python3 tools/callback_tool.py rate | grep " 200 \| 100 "
for nt in 52 32 28; do
    echo "== pass $rep: tile-per-workgroup, TOLFG_TILE_NODES=$nt"
    TOLFG_NO_SINGLE_LAUNCH=1 TOLFG_TILE_NODES=$nt python3 tools/callback_tool.py rate | grep " 200 \| 100 "
done
done
;;
r04_incache_counters)
# VERDICT r3 task 3: what binds the launches whose outputs fit the cache (configs[3], B = 1024 fp64: 0.56-0.60 of peak)?
# rocprofv3 kernel-trace + stats, then one counter per pass, for B = 1024 (plain stores, the plan's choice), B = 2048 forced
# plain, and B = 2048 as planned (non-temporal) for comparison.  Run on the GPU box from the repo root.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04_incache
mkdir -p "$OUT"
ARGS="--mission S10 --ts 200 --steps 40 --warmup 5 --min-warm-seconds 0 --no-calibration --no-cpu-baseline --no-configs"
run() {    # tag, batch, extra env
    local tag=$1 B=$2; shift 2
    echo "#### $tag"
    env "$@" true
    ( export "$@" 2>/dev/null; timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag/stats" -o stats -- python3 bench.py $ARGS --batch $B > "$OUT/$tag.stats.log" 2>&1 )
    grep -h "fg_kernel" $(find "$OUT/$tag/stats" -name "*kernel_stats.csv") | cut -c1-260
    for c in $COUNTERS; do
        ( export "$@" 2>/dev/null; timeout -k 5 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$tag/$c" -o pmc -- python3 bench.py $ARGS --batch $B > "$OUT/$tag.$c.log" 2>&1 ) \
            && python3 tools/pmc_avg.py $(find "$OUT/$tag/$c" -name "*counter_collection.csv" | head -1) | grep -v "^$" || echo "$c: pass failed ($(grep -m1 -E 'Missing|error|rror' "$OUT/$tag.$c.log" | cut -c1-120))"
    done
}
COUNTERS="FETCH_SIZE WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum TCC_WRITEBACK_sum TCC_TAG_STALL_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
run b1024_plain 1024 TOLFG_DUMMY=1
run b2048_plain 2048 TOLFG_NT_STORES=0 TOLFG_WAVES_PER_CU=0
COUNTERS="WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_sum TCC_WRITEBACK_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
run b2048_planned 2048 TOLFG_DUMMY=1
find "$OUT" -name "*.csv" -size +2M -delete
;;
r04_fp32_traffic)
# VERDICT r3 task 4: where do the extra 4.6 % of HBM bytes of the fp32 launches come from (fp64: 2.1 %)?
# One counter per pass over the 8192-trajectory launch: mixed (the stated config), S10 only, G7 only (G7 rows in fp32 start 8 bytes
# off a 16-byte boundary: shifted streams), fp64 mixed for comparison.  Run on the GPU box from the repo root.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04_fp32
mkdir -p "$OUT"
COMMON="--ts 200 --batch 8192 --steps 30 --warmup 3 --min-warm-seconds 0 --no-calibration --no-cpu-baseline --no-configs"
for cfg in "mixed f32" "S10 f32" "G7 f32" "mixed f64"; do
    set -- $cfg
    tag=$1_$2
    echo "#### $tag"
    for c in FETCH_SIZE WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum; do
        timeout -k 5 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$tag/$c" -o pmc -- python3 bench.py $COMMON --mission $1 --dtype $2 > "$OUT/$tag.$c.log" 2>&1 \
            && python3 tools/pmc_avg.py $(find "$OUT/$tag/$c" -name "*counter_collection.csv" | head -1) || echo "$c: pass failed ($(grep -m1 -E 'Missing|rror' "$OUT/$tag.$c.log" | cut -c1-120))"
    done
    python3 - <<PY
import sys; sys.path.insert(0, ".")
import tol_amd, bench as BN
air = BN.AIRCRAFT5 if "$1" == "mixed" else ("tempest",)
bt = tol_amd.Batch("$1", air, ts=200, dtype="$2")
bt.set_trajectories(BN.make_trajectories(tol_amd, 8192, 0, "$1", len(air)))
n, neF, neG = bt.n, bt.neF, bt.neG
es = 8 if "$2" == "f64" else 4
print("algorithmic bytes per launch %.1f MB (x read %.1f MB, F+G written %.1f MB)" % (bt.algorithmic_bytes(8192) / 1e6, es * 8192 * n / 1e6, (bt.algorithmic_bytes(8192) - es * 8192 * n) / 1e6))
PY
done
find "$OUT" -name "*.csv" -size +2M -delete
;;
r04_align_counters)
# Write requests of the launch with the slab streams cut to 64-byte boundaries against the 16-byte form (TOLFG_STREAM_ALIGN16=1).
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04_align
mkdir -p "$OUT"
COMMON="--ts 200 --batch 8192 --steps 30 --warmup 3 --min-warm-seconds 0 --no-calibration --no-cpu-baseline --no-configs"
for dt in f64 f32; do
for a16 in 0 1; do
    export TOLFG_STREAM_ALIGN16=$a16
    tag=${dt}_align16_$a16
    echo "#### $tag"
    for c in WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TA_DATA_STALLED_BY_TC_CYCLES_sum; do
        timeout -k 5 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$tag/$c" -o pmc -- python3 bench.py $COMMON --dtype $dt > "$OUT/$tag.$c.log" 2>&1 \
            && python3 tools/pmc_avg.py $(find "$OUT/$tag/$c" -name "*counter_collection.csv" | head -1) || echo "$c: pass failed"
    done
done
done
find "$OUT" -name "*.csv" -size +2M -delete
;;
r04_alloc_probe)
# One allocation probe per call ($2 = 1 ... 8, further arguments are the probe's own, e.g. "3 order"): where the output buffer lands decides
# the class the launch runs in (profiles/r04_allocation_classes.md).  Round 5 kept tools/alloc_probe.py (n = 1) only; probes 2 ... 10 are in the
# history (git show a2af350:tools/alloc_probeN.py), their outputs under profiles/r04/.
set -u
n=${2:-7}
shift; shift || true
tool=tools/alloc_probe$([ "$n" = 1 ] || echo "$n").py
[ -f "$tool" ] || { echo "$tool was removed in round 5: git show a2af350:$tool > $tool"; exit 1; }
mkdir -p gpurun_out/r04_alloc
timeout -k 10 400 python3 "$tool" "$@" 2>&1 | grep -v amdgpu.ids | tee "gpurun_out/r04_alloc/probe_$n.txt"
;;
r04_x0_time)
# Initial guesses on the device: serial walk vs the table-driven node-parallel kernel, and the bitwise test (profiles/r04_x0_kernel.md).
set -u
mkdir -p gpurun_out/r04_x0
timeout -k 10 240 python3 -m pytest tests/test_device_setup.py -m gpu -q 2>&1 | tail -2
timeout -k 10 200 python3 tools/x0_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_x0/x0_time.txt
;;
list|*)
cat <<'LIST'
r02a                   round-2 experiment A (one gpurun call): write-stream shapes, tile size x cap x fused sweep, callback trace  [r02_write_shapes.md]
r02aa                  compact pattern (46-entry slabs): resident-wave cap x tile size x fused, same box  [r02_tile_fused_sweep.md]
r02ab                  fp32: resident-wave cap, reference and compact pattern, S10 / G7 / mixed, same box  [r02_tile_fused_sweep.md]
r02ac                  s_setprio experiments: 1 = store phase at priority 3, 2 = load/compute phase at priority 3 (dropped to 0 for the stores)  [-]
r02ad                  do the XCDs finish their eighths at systematically different times?  (stamped build: timelines only)  [r02_xcd_balance.md]
r02ae                  is the odd/even XCD asymmetry a property of the XCD or of the eighth of the output it walks?  (stamped build: timelines only) variant 65536: XCD x walks eighth x^1;  131072: (x+4)%8;  262144: (x+2)%8  [r02_xcd_balance.md]
r02ai                  what attaching start/stop events to every dispatch costs: bench.py reports the uninstrumented timed region and the instrumented pass  [r02_event_cost.md]
r02aj                  the launch-plan decisions again, timed without per-dispatch events: fused vs two launches, cap, store flavour, tile size  [r02_tile_fused_sweep.md]
r02ak                  phase shares and timeline of the compact pattern (stamped build: shares only) next to the reference pattern  [r02_launch_timeline.md]
r02al                  compact pattern, fp64: resident-wave cap, uninstrumented time column  [r02_launch_timeline.md]
r02an                  how much would full 64-node tiles buy at ts = 200?  ts = 192 and 256 have them (3 / 4 tiles of 64), ts = 200 has 4 x 52  [DESIGN.md, r02_tile_fused_sweep.md]
r02ao                  does the row stride of G (the spacing of the concurrent store fronts) matter?  ldgpad = extra elements between rows  [r02_write_shapes.md]
r02ap                  cache-policy bits of the slab stores inside the fg kernel (TOLFG_STORE_FLAVOR builds): 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0 sc1 nt  [r02_write_shapes.md]
r02aq                  write-stream shapes, part 9: 52 KiB per wave written as 4 KiB chunks interleaved over a small group of G waves (mode 11), G = 4 ... 64  [r02_write_shapes.md]
r02as                  do 128-byte-aligned slab regions matter?  ldgpad=2 makes the row stride a multiple of 128 B; goff=4 then puts every row's slab region (row + c0) on a 128-byte line, so that no line is shared between two tile waves  [r02_write_shapes.md]
r02at                  which 16-byte position of a row's slab region inside a 128-byte line is fast?  ldgpad=2: every row the same position; goff shifts it  [r02_write_shapes.md]
r02b                   round-2 experiment B: cooperative write shapes (wrbench modes 7-10), fused variants  [r02_write_shapes.md]
r02c                   round-2 experiment C: XCD placement of write streams; plain vs non-temporal stores when the outputs fit the Infinity Cache  [r02_write_shapes.md]
r02d                   round-2 experiment D: XCD-contiguous tile order, nt vs plain slab stores, fused (polling) finalize, tile size  [-]
r02e                   round-2 experiment E: SNOPT-callback latency -- completion word, registered caller arrays, zero-copy limit  [r02_callback.md]
r02f                   round-2 experiment F: GPU suite after the mixed-mission / callback changes, callback rate, mixed + fp32-G7 timings  [r02_callback.md]
r02g                   round-2 experiment G: GPU suite, callback rate, bench.py (new layout)  [-]
r02h                   round-2 experiment H: fast sincos -- GPU suite, tile sizes again, callback rate, bench configs  [-]
r02i                   round-2 experiment I: GPU suite, native callback timing, bench, rocprofv3 stats + PMC passes  [r02_callback.md]
r02j                   round-2 experiment J: GPU suite; compact-pattern launch choices; TA counters one per pass  [r02_ta_counters.md]
r02k                   round-2 experiment K: compact pattern A/B (own vs library sin/cos, cap, xcd, fused)  [-]
r02l                   round-2 experiment L: round-1 build vs round-2 build on the SAME box (reference and compact patterns), interleaved  [r02_same_box_ab.md]
r02m                   round-2 experiment M: shifted slab stream -- GPU suite, fp32 / G7 / odd-ts timings  [-]
r02n                   round-2 experiment N: same-box A/B of the build before the shifted slab stream (3c61bd0) and HEAD; cost of the timing events  [-]
r02o                   round-2 experiment O: LDS sized to the tile (11 waves per CU uncapped at ts=200); where the ~6 us between evaluations go  [-]
r02p                   round-2 experiment P: where a launch's time goes at B = 128 ... 4096 (stamped build: shares and timeline, not benchmark numbers)  [r02_launch_timeline.md]
r02q                   round-2 experiment Q: occupancy over time inside a launch (stamped build)  [r02_launch_timeline.md]
r02r                   round-2 experiment R: persistent workgroups with per-XCD tile queues vs one workgroup per tile  [r02_persistent.md]
r02s                   round-2 experiment S: deal only part of the tiles XCD-contiguously, the launch's tail in id order (all XCDs share it)  [DESIGN.md, r02_persistent.md]
r02t                   round-2 experiment T: final build -- GPU suite, callback timings incl. F-only calls  [r02_callback.md]
r02u                   round-2 experiment U: own fp32 sin/cos -- GPU suite, fp32 timings against the library routine (same box)  [DESIGN.md]
r02v                   round-2 experiment V: compact pattern -- persistent form and caps (fused / unfused)  [r02_persistent.md]
r02w                   round-2 experiment W: bench.py with the two-streams side record  [-]
r02x                   write-stream shapes, part 4: is it the LENGTH of a wave's stream or the compactness of the in-flight address window?  [r02_write_shapes.md]
r02x2                  write-stream shapes, part 5: one store instruction spread over several places of the wave's region  [r02_write_shapes.md]
r02x3                  write-stream shapes, part 6: cache-policy bits of the 16-byte stores  [-]
r02y                   finer tiles for the trajectories a launch reaches last (FgArgs::tail_count): same-box A/B with tools/fgbench  [r02_tail_tiles.md]
r02z                   finer-tiled tail at the batch sizes whose outputs fit the Infinity Cache (plain stores, no cap) and at B=2048  [r02_tail_tiles.md]
r03a                   round 3, step A: table-driven slab stream vs round 2's build, same box; then the GPU suite  [-]
r03b                   dynamic instruction mix and wait states: fp32 mixed reference pattern, fp64 compact, fp64 reference  [-]
r03c                   packed fp32 (two nodes per lane, 128-node tiles) vs one node per lane, same box  [-]
r03d                   where does an fp32 launch's time go: ablations (results wrong by construction; MISMATCH expected), same box  [-]
r03e                   tile size vs batch size for launches that fit the cache (plain stores, no cap): what should plan_launch choose?  [-]
r03f                   pipelined stream loop (vs exp_r03e on another box: compare within this call only) and issue-priority stagger for launches within the cache  [-]
r03h                   the restructured bench.py: default 1-GPU line, then the 2-rank gloo rehearsal on one GPU  [-]
r03l                   compact pattern: where does a launch's time go (ablations; results wrong by construction), fp64 and fp32, same box  [see profiles/README.md]
r03m                   x-window prefetch by leaving waves (FgArgs::prefetch = tiles ahead on the XCD's run): distance sweep, same box  [see profiles/README.md]
r03n                   clocks and power while the headline loop runs (VERDICT r2 item 7), then the same for a pure fill; then the rocprofv3 passes  [see profiles/README.md]
r03o                   launches within the cache: resident-wave cap x issue-priority stagger (B = 1024 / 2048, fp64 and fp32), same box  [see profiles/README.md]
r03p                   stagger on/off alternating on the SAME buffers (one process per shape), three fresh processes each: is the gain real or allocation luck?  [see profiles/README.md]
r03q                   round-3 full pass: GPU suite in its three forms, smoke, bench.py, shape sweep  [see profiles/README.md]
r03r                   stream-table loads issued before the x window (one memory latency less per wave): previous build vs this one, alternating processes  [see profiles/README.md]
r03s                   with the shorter front end (table loads before the window), re-sweep the resident-wave cap of the launches beyond the cache  [see profiles/README.md]
r03t                   table loads before (variant 0) vs after (variant 8192) the x window: the SAME buffers, alternating, ablation build (both orders compiled in)  [see profiles/README.md]
r03u                   slab stream: 1 / 2 / 3 rounds of LDS reads in flight ahead of the store (separate binaries: compare the small-B rows, where one wave is alone on its SIMD); and a pure store loop at low occupancy (wrbench) for the per-wav  [see profiles/README.md]
r03v                   compiler scheduling strategy: default (max occupancy) vs -mllvm -amdgpu-sched-strategy=max-ilp (separate binaries, two passes)  [see profiles/README.md]
r03w                   finalize reads its 23 edge values at the wave's start (variant 0) vs at the end (variant 16384): same buffers, alternating; then the GPU suite  [see profiles/README.md]
r03x                   what do the per-wave loads of the stream table cost?  ablation: offsets made up in registers (variant 32768, results wrong) vs loaded (0); same buffers  [see profiles/README.md]
r03y                   dt = x[0] through a scalar load (variant 0) vs the vector load of rounds 1-2 (variant 65536): same buffers, alternating  [see profiles/README.md]
r04_callback_tiles     the callback (B = 1) as one workgroup vs tile workgroups of several sizes  [r04_callback_tiles.md]
r04_incache_counters   what binds launches whose outputs fit the cache: rocprofv3 stats + one counter per pass, B = 1024 / 2048  [r04_incache_counters.md]
r04_fp32_traffic       where the extra 4.6 % of HBM bytes of the fp32 launches come from: write / read request counters, mixed / S10 / G7 / fp64  [r04_fp32_traffic.md]
r04_align_counters     write requests with the slab streams cut to 64-byte boundaries (r04_stream_align64.patch applied) vs the 16-byte form  [r04_fp32_traffic.md]
r04_alloc_probe N      one of round 4's allocation probes (N = 1: tools/alloc_probe.py; 2 ... 8 live in the history since round 5; further arguments are the probe's own)  [r04_allocation_classes.md]
r04_x0_time            initial guesses on the device: serial walk vs the table-driven node-parallel kernel, bitwise test first  [r04_x0_kernel.md]
LIST
;;
esac
