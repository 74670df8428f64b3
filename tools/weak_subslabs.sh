#!/bin/bash
# 2-rank weak deck, copies side by side along j: coarse space of the pressure stage with m sub-slabs per rank (they keep the vertical wells whole)
run2() { echo "== 2 ranks: $*"; env $2 OPMGPU_COMM_TRANSPORT=shm timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 2 --steps 12 --warmup 2 --no-cpu-baseline --weak-axis 1 --detail gpurun_out/weak_detail.json ${@:3} 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['value'], d['ms_per_step'], d['config']['linear_its_per_solve'], d['per_time_step'])"; }
P=29830
for m in 0 4 8; do
  run2 $((P++)) OPMGPU_COARSE_SUBSLABS=$m --krylov gmres
  run2 $((P++)) OPMGPU_COARSE_SUBSLABS=$m --krylov bicgstab
done
run2 $((P++)) OPMGPU_COARSE_SUBSLABS=8 --krylov gmres --no-wells
run2 $((P++)) OPMGPU_COARSE_SUBSLABS=0 --krylov gmres --no-wells
