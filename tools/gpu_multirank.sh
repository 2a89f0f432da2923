#!/bin/bash
# rehearsal of bench.py's N = 2 path on the one-GPU box (gloo, both ranks on cuda:0)
export UCF_BENCH_BACKEND=gloo UCF_BENCH_ONE_DEVICE=1
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --nt 256 2>&1 | tail -3
