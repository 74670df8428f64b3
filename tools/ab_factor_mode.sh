#!/bin/bash
# A/B on ONE box: where the early ILU0 factorisation starts (OPMGPU_FACTOR_EARLY: 0 = at the solve, 1 = behind the assembly, 2 = behind the convergence check's kernels)
for mode in 1 2 0 1 2; do
  for kry in gmres bicgstab; do
    OPMGPU_FACTOR_EARLY=$mode python bench.py --only-main --no-cpu-baseline --krylov $kry --steps 40 --stat-calls 80 --detail gpurun_out/ab_detail.json 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mode $mode $kry', d['value'], d['ms_per_solving_iteration_median'], d['ms_per_solving_iteration_mean'], d['breakdown_ms'], d['config']['linear_its_per_solve'], d['per_time_step']['ms_per_converged_time_step'])"
  done
done
