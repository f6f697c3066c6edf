#!/bin/bash
# the restructured bench.py: default 1-GPU line, then the 2-rank gloo rehearsal on one GPU
O=gpurun_out/r03h; mkdir -p $O
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; tail -3 $O/bench.err; python tools/show_bench.py $O/bench.json
bash tools/rehearse_ranks.sh > $O/rehearse.log 2>&1; echo "rehearse exit $?"; cat $O/rehearse.log | cut -c1-330
