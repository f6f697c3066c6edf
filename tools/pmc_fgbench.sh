#!/bin/bash
# usage: tools/pmc_fgbench.sh <tag> <fgbench binary> "<fgbench args>"     (on the GPU box, repo root)
# SQ counter passes (kernel-trace only) over the native harness; averages per kernel name into gpurun_out/pmc_<tag>/summary.txt.
# One group of <= 7 SQ counters per pass (8 SQ slots on gfx950); the harness is the program after `--`.
TAG=$1; EXE=$2; ARGS=$3
OUT=$PWD/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/p$i" -o pmc -- $EXE $ARGS > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; grep -m3 -E "Missing|error|fatal|rror" "$OUT/p$i.log"; tail -3 "$OUT/p$i.log"; continue; }
done
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        k = r["Kernel_Name"]
        if "fg_kernel" not in k and "finalize" not in k: continue
        m = re.search(r"(fg_kernel|finalize_kernel)<([^>]*)>", k)
        acc[m.group(0) if m else k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-26s avg %16.1f  n=%d" % (c, sum(v) / len(v), len(v)))
PY
