#!/bin/bash
# N ranks of bench.py on ONE GPU over gloo (the RCCL path needs N GPUs): checks that the N>1 code path of bench.py runs and prints
# the JSON line.  usage: tools/rehearse_multi.sh N LOG2N [extra bench args]
N=$1; L=$2; shift; shift
export NGRAVS_BENCH_BACKEND=gloo NGRAVS_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) bench.py --gpus $N --steps 2 --warmup 1 --log2n $L --no-cpu-baseline "$@"
