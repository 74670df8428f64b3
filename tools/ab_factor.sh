#!/bin/bash
# A/B on ONE box: workgroup cap of the side-stream ILU0 factorisation (OPMGPU_FACTOR_GRID; 0 = uncapped), headline and BiCGStab
for cap in 0 256 512 1024 2048 0 512; do
  for kry in gmres bicgstab; do
    OPMGPU_FACTOR_GRID=$cap python bench.py --only-main --no-cpu-baseline --krylov $kry --steps 40 --stat-calls 80 --detail gpurun_out/ab_detail.json 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cap $cap $kry', d['value'], d['ms_per_solving_iteration_median'], d['ms_per_solving_iteration_mean'], d['breakdown_ms'], d['config']['linear_its_per_solve'], d['per_time_step']['ms_per_converged_time_step'])"
  done
done
