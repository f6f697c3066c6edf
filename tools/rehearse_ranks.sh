#!/bin/bash
# two ranks sharing ONE GPU over gloo: rehearses bench.py's multi-rank step loop -- the default line (weak scaling of the
# mixed headline + the stated configs[3] / configs[4] as strong-scaling records), then a strong-scaling run with a ragged split
export HSA_ENABLE_IPC_MODE_LEGACY=0
for extra in "--batch 1024" "--mission S10 --global-batch 1001 --no-configs"; do
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 20 --warmup 3 $extra --backend gloo 2>&1 | tail -1 | python tools/show_bench.py | cut -c1-330
done
