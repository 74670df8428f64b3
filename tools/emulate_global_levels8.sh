#!/bin/bash
# the weak-scaling deck of 8 ranks on one GPU (8 M cells), preconditioner cut into 8 slabs
run1() { echo "== $*"; env $1 $2 $3 $4 timeout -k 10 900 python bench.py --only-main --no-cpu-baseline --steps 10 --warmup 2 --stat-calls 30 --detail gpurun_out/weak_detail.json ${@:5} 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['linear_its_per_solve'], d['per_time_step'])"; }
run1 A=0 B=0 C=0 D=0 --stack 8 --krylov gmres
run1 OPMGPU_EMULATE_RANKS=8 OPMGPU_EMULATE_WHAT=3 OPMGPU_EMULATE_GLOBAL_LEVELS=0 D=0 --stack 8 --krylov gmres
run1 OPMGPU_EMULATE_RANKS=8 OPMGPU_EMULATE_WHAT=3 OPMGPU_EMULATE_GLOBAL_LEVELS=1 D=0 --stack 8 --krylov gmres
run1 OPMGPU_EMULATE_RANKS=8 OPMGPU_EMULATE_WHAT=1 C=0 D=0 --stack 8 --krylov gmres
run1 OPMGPU_EMULATE_RANKS=8 OPMGPU_EMULATE_WHAT=3 OPMGPU_EMULATE_GLOBAL_LEVELS=0 D=0 --stack 8 --krylov bicgstab
run1 A=0 B=0 C=0 D=0 --stack 8 --krylov bicgstab
