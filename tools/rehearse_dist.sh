#!/bin/bash
# Rehearsal of bench.py's N > 1 code paths on ONE GPU: every rank on cuda:0, coupled by the shared-memory TEST transport (OPMGPU_COMM_TRANSPORT=shm),
# gloo for the barrier.  The rates mean nothing; what is checked is that every leg (weak, strong along j with its wells, SPE10-like with its wells,
# the self-launch without torch.distributed.run) builds, converges and prints its JSON line.   gpurun -- 'bash tools/rehearse_dist.sh <tag>'
TAG=${1:-rehearsal}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export OPMGPU_COMM_TRANSPORT=shm
COMMON="--steps 8 --warmup 2 --no-cpu-baseline"
run() { name=$1; shift; echo "== $name: $*"; timeout -k 10 400 "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "   exit $?"; tail -c 400 $OUT/$name.json; echo; }
run weak2 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --nx 40 --ny 40 --nz 40 $COMMON
run strong2 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 --nx 40 --ny 40 --nz 40 --scaling strong $COMMON
run strong4 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 4 --nx 40 --ny 40 --nz 40 --scaling strong $COMMON
run selflaunch2 python bench.py --gpus 2 --nx 40 --ny 40 --nz 40 $COMMON
run spe10_2 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --deck spe10like $COMMON
