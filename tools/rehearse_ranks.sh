#!/bin/bash
# two ranks sharing ONE GPU over gloo: rehearses bench.py's multi-rank step loop (weak and strong scaling)
export HSA_ENABLE_IPC_MODE_LEGACY=0
for extra in "--batch 1024" "--global-batch 1024" "--mission mixed --global-batch 1001"; do
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 20 --warmup 3 $extra --backend gloo 2>&1 | tail -1 | python tools/show_bench.py | head -1 | cut -c1-250
done
