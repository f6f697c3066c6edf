#!/bin/bash
# rehearse the multi-rank step loop on the one-GPU box: 2 ranks share GPU 0, gloo gather
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 3 --batch 1024 --backend gloo 2>&1 | tail -3
# and the single-process torchrun path the driver uses for N=1? (plain python) plus nccl init with world=1 via torchrun
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-callback 2>&1 | tail -1 | cut -c1-300
