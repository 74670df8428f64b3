#!/bin/bash
# which stage of the decomposed preconditioner costs the columns on the SMOOTH weak-scaling deck (2 copies stacked along k, one GPU, preconditioner
# built from a copy of the matrix cut into 2 slabs): WHAT 1 = only the ILU0's copy is cut, 2 = only the AMG's, 3 = both; L0_GLOBAL = level 0 uncut
run1() { echo "== $*"; env $1 $2 $3 timeout -k 10 500 python bench.py --only-main --no-cpu-baseline --steps 12 --warmup 2 --stack 2 --detail gpurun_out/weak_detail.json ${@:4} 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['linear_its_per_solve'], d['per_time_step'])"; }
for kry in gmres bicgstab; do
  run1 X=0 Y=0 Z=0 --krylov $kry
  run1 OPMGPU_EMULATE_RANKS=2 OPMGPU_EMULATE_WHAT=1 Z=0 --krylov $kry
  run1 OPMGPU_EMULATE_RANKS=2 OPMGPU_EMULATE_WHAT=2 Z=0 --krylov $kry
  run1 OPMGPU_EMULATE_RANKS=2 OPMGPU_EMULATE_WHAT=2 OPMGPU_EMULATE_L0_GLOBAL=1 --krylov $kry
  run1 OPMGPU_EMULATE_RANKS=2 OPMGPU_EMULATE_WHAT=3 Z=0 --krylov $kry
  run1 OPMGPU_EMULATE_RANKS=2 OPMGPU_EMULATE_WHAT=3 OPMGPU_EMULATE_L0_GLOBAL=1 --krylov $kry
done
