#!/bin/bash
# the weak-scaling deck of 2 ranks (two copies of the 100^3 workload, one 5-spot each): single-domain iteration counts on ONE GPU against the
# decomposed run, copies stacked along k (axis 2: the cut goes through the strong vertical coupling) or side by side along j (axis 1)
run1() { echo "== one GPU: $*"; timeout -k 10 500 python bench.py --only-main --no-cpu-baseline --steps 12 --warmup 2 --detail gpurun_out/weak_detail.json "$@" 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['value'], d['ms_per_step'], d['config']['linear_its_per_solve'], d['per_time_step'])"; }
run2() { echo "== 2 ranks: $*"; OPMGPU_COMM_TRANSPORT=shm timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 2 --steps 12 --warmup 2 --no-cpu-baseline --detail gpurun_out/weak_detail.json ${@:2} 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['value'], d['ms_per_step'], d['config']['linear_its_per_solve'], d['per_time_step'])"; }
P=29810
for ax in 1 2; do
  for kry in gmres bicgstab; do
    run1 --stack 2 --weak-axis $ax --krylov $kry
    run2 $((P++)) --weak-axis $ax --krylov $kry
  done
done
