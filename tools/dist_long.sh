#!/bin/bash
# decomposed runs over many time steps (shared-memory test transport, every rank on cuda:0): chopped attempts and iterations over ~150 Newton calls
export OPMGPU_COMM_TRANSPORT=shm
run() { echo "== $*"; timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $2 bench.py --gpus $1 --steps 150 --warmup 2 --stat-calls 150 --no-cpu-baseline --detail gpurun_out/dist_long_detail.json ${@:3} 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['config']['linear_solver'], '| its/solve', d['config']['linear_its_per_solve'], d['per_time_step'])"; }
run 2 29901 --krylov gmres
run 2 29902 --krylov bicgstab
run 4 29903 --nx 60 --ny 60 --nz 60 --krylov gmres
run 2 29904 --deck spe10like
run 4 29905 --deck spe10like
run 4 29906 --deck spe10like --krylov gmres --stage2-relax 1.0
run 2 29907 --dt-days 20 --krylov bicgstab
