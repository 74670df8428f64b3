#!/bin/bash
# the Norne-like deck (irregular graph: 60 % inactive cells, NNCs, 36 wells) decomposed into slabs of j-rows, 2 and 4 real ranks over the test transport
export OPMGPU_COMM_TRANSPORT=shm
run1() { echo "== one GPU: $*"; timeout -k 10 300 python bench.py --only-main --no-cpu-baseline --steps 60 --warmup 2 --stat-calls 120 --deck nornelike --detail gpurun_out/norne_detail.json "$@" 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['config']['linear_solver'], '| its/solve', d['config']['linear_its_per_solve'], d['per_time_step'])"; }
run() { echo "== $1 ranks: ${@:3}"; timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $2 bench.py --gpus $1 --deck nornelike --steps 120 --warmup 2 --stat-calls 120 --no-cpu-baseline --detail gpurun_out/norne_detail.json ${@:3} 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['config']['linear_solver'], '| its/solve', d['config']['linear_its_per_solve'], d['config']['cells_per_gpu'], d['per_time_step'])"; }
run1 --krylov bicgstab
run1 --krylov gmres
run 2 29921 --krylov bicgstab
run 2 29922 --krylov gmres
run 4 29923 --krylov bicgstab
run 4 29924 --krylov gmres
run 4 29925 --krylov bicgstab --solver ilu0
