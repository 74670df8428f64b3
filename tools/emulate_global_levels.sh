#!/bin/bash
# how many of the COARSEST levels of the pressure hierarchy have to be global (uncut Galerkin operators) before the decomposed preconditioner
# gets the single-domain iteration counts back -- emulated on one GPU (both stages built from a copy of the matrix cut into slabs)
run1() { echo "== $*"; env $1 $2 $3 $4 timeout -k 10 500 python bench.py --only-main --no-cpu-baseline --steps 12 --warmup 2 --detail gpurun_out/weak_detail.json ${@:5} 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['linear_its_per_solve'], d['per_time_step'])"; }
for kry in gmres bicgstab; do
  run1 A=0 B=0 C=0 D=0 --stack 2 --krylov $kry
  for q in 0 1 2 3 4; do
    run1 OPMGPU_EMULATE_RANKS=2 OPMGPU_EMULATE_WHAT=3 OPMGPU_EMULATE_GLOBAL_LEVELS=$q D=0 --stack 2 --krylov $kry
  done
  run1 OPMGPU_EMULATE_RANKS=2 OPMGPU_EMULATE_WHAT=3 OPMGPU_EMULATE_GLOBAL_LEVELS=2 OPMGPU_EMULATE_L0_GLOBAL=1 --stack 2 --krylov $kry
done
for q in 0 1 2 3; do
  run1 OPMGPU_EMULATE_RANKS=4 OPMGPU_EMULATE_WHAT=3 OPMGPU_EMULATE_GLOBAL_LEVELS=$q D=0 --deck spe10like --krylov bicgstab --stage2-relax 0.9
done
run1 OPMGPU_EMULATE_RANKS=4 OPMGPU_EMULATE_WHAT=3 OPMGPU_EMULATE_GLOBAL_LEVELS=2 OPMGPU_EMULATE_L0_GLOBAL=1 --deck spe10like --krylov bicgstab --stage2-relax 0.9
