#!/bin/bash
# bench.py --gpus N in rehearsal mode: N ranks share the box's one GPU (gloo + host staging instead of RCCL): the plumbing of both
# multi-GPU splits and the bytes on the wire, not a scaling figure.   usage (GPU box, repo root): tools/rehearsal_bench.sh [N=2] [photons per rank]
n=${1:-2}
RRT_BENCH_REHEARSAL=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus $n --steps 6 --warmup 2 --photons-per-rank ${2:-1000000}
