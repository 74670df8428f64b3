#!/bin/bash
# SPE10-like deck on 2 real ranks (shared-memory transport, both on cuda:0): what the decomposed preconditioner costs there and what helps
export OPMGPU_COMM_TRANSPORT=shm
run() { echo "== $*"; env $1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $2 bench.py --gpus 2 --deck spe10like --steps 12 --warmup 2 --no-cpu-baseline --detail gpurun_out/dist_spe10_detail.json ${@:3} 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['linear_solver'], d['config']['linear_its_per_solve'], d['per_time_step'])"; }
run X=1 29701
run X=1 29702 --stage2-relax 0.9
run X=1 29703 --krylov bicgstab
run X=1 29704 --krylov bicgstab --stage2-relax 0.9
run OPMGPU_COARSE=0 29705 --stage2-relax 0.9
echo "== one rank (reference)"
unset OPMGPU_COMM_TRANSPORT
python bench.py --deck spe10like --only-main --no-cpu-baseline --steps 12 --warmup 2 --detail gpurun_out/dist_spe10_detail.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['linear_solver'], d['config']['linear_its_per_solve'], d['per_time_step'])"
python bench.py --deck spe10like --only-main --no-cpu-baseline --steps 12 --warmup 2 --stage2-relax 0.9 --detail gpurun_out/dist_spe10_detail.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['linear_solver'], d['config']['linear_its_per_solve'], d['per_time_step'])"
