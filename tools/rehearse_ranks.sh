#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 20 --warmup 3 --batch 1024 --backend gloo 2>&1 | tail -1 | cut -c1-200
# NCCL path with a single rank group (world=1 via torchrun -> no dist) and plain run
timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-configs 2>&1 | tail -1 | cut -c1-160
