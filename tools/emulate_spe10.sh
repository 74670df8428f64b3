#!/bin/bash
# which stage of the decomposed preconditioner costs the iterations on the SPE10-like deck?  One GPU, the preconditioner built from a matrix copy
# cut into N index-range slabs (OPMGPU_EMULATE_RANKS; OPMGPU_EMULATE_WHAT: 1 = only the ILU0's copy is cut, 2 = only the AMG's, 3 = both)
for cfg in "OPMGPU_EMULATE_RANKS=1" "OPMGPU_EMULATE_RANKS=4 OPMGPU_EMULATE_WHAT=1" "OPMGPU_EMULATE_RANKS=4 OPMGPU_EMULATE_WHAT=2" "OPMGPU_EMULATE_RANKS=4 OPMGPU_EMULATE_WHAT=3" "OPMGPU_EMULATE_RANKS=4 OPMGPU_EMULATE_WHAT=3 OPMGPU_COARSE=0"; do
  for extra in "" "--stage2-relax 0.9"; do
    echo "== $cfg $extra"
    env $cfg python bench.py --deck spe10like --only-main --no-cpu-baseline --krylov bicgstab --steps 12 --warmup 2 --stat-calls 40 $extra --detail gpurun_out/emu_detail.json 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['linear_its_per_solve'], d['per_time_step'])"
  done
done
