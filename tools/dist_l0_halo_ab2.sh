#!/bin/bash
# which of the exchanges around the decomposed pressure cycle are needed once level 0 is global: down-leg exchange (L0_HALO=2 drops it), x_p halo behind the cycle (HALO_XP=0 drops it)
export OPMGPU_COMM_TRANSPORT=shm
run() { echo "== $*"; env $1 $2 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $3 --master-addr 127.0.0.1 --master-port $4 bench.py --gpus $3 --steps 12 --warmup 2 --no-cpu-baseline --detail gpurun_out/dist_l0_detail.json ${@:5} 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['linear_its_per_solve'], d['config']['cells_per_gpu'], d['per_time_step'])"; }
P=29760
for cfg in "OPMGPU_CPR_L0_HALO=1 OPMGPU_CPR_HALO_XP=1" "OPMGPU_CPR_L0_HALO=2 OPMGPU_CPR_HALO_XP=1" "OPMGPU_CPR_L0_HALO=1 OPMGPU_CPR_HALO_XP=0" "OPMGPU_CPR_L0_HALO=2 OPMGPU_CPR_HALO_XP=0"; do
  run $cfg 4 $((P++)) --deck spe10like --stage2-relax 1.0 --krylov bicgstab
  run $cfg 4 $((P++)) --deck spe10like --stage2-relax 1.0 --krylov gmres
  run $cfg 2 $((P++))
done
