#!/bin/bash
# A/B of the decomposed pressure cycle with level 0 on the global matrix (OPMGPU_CPR_L0_HALO) -- ranks share cuda:0 over the shm test transport
export OPMGPU_COMM_TRANSPORT=shm
run() { echo "== $*"; env $1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $2 --master-addr 127.0.0.1 --master-port $3 bench.py --gpus $2 --steps 12 --warmup 2 --no-cpu-baseline --detail gpurun_out/dist_l0_detail.json ${@:4} 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['linear_its_per_solve'], d['config']['cells_per_gpu'], d['per_time_step'])"; }
run OPMGPU_CPR_L0_HALO=0 4 29741 --deck spe10like --stage2-relax 1.0 --krylov gmres
run OPMGPU_CPR_L0_HALO=1 4 29742 --deck spe10like --stage2-relax 1.0 --krylov gmres
run OPMGPU_CPR_L0_HALO=0 4 29743 --deck spe10like --stage2-relax 0.9 --krylov bicgstab
run OPMGPU_CPR_L0_HALO=1 4 29744 --deck spe10like --stage2-relax 0.9 --krylov bicgstab
run OPMGPU_CPR_L0_HALO=1 4 29745 --deck spe10like --stage2-relax 1.0 --krylov bicgstab
run OPMGPU_CPR_L0_HALO=0 2 29746 --deck spe10like --stage2-relax 1.0 --krylov gmres
run OPMGPU_CPR_L0_HALO=1 2 29747 --deck spe10like --stage2-relax 1.0 --krylov gmres
run OPMGPU_CPR_L0_HALO=0 2 29748
run OPMGPU_CPR_L0_HALO=1 2 29749
