#!/bin/bash
# A/B on ONE box: the early ILU0 factorisation start (the entry-layout experiments of round 4 -- transposed slot byte, per-entry g dz word --
# are measured in profiles/r04_f_ab.log and no longer switchable: their code paths cost registers in the assembly kernel)
for cfg in "OPMGPU_FACTOR_EARLY=1" "OPMGPU_FACTOR_EARLY=0" "OPMGPU_FACTOR_EARLY=1" "OPMGPU_FACTOR_EARLY=0"; do
  echo "== $cfg"
  for kry in gmres bicgstab; do
    env $cfg python bench.py --only-main --no-cpu-baseline --krylov $kry --steps 40 --stat-calls 80 --detail gpurun_out/ab_detail.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_solving_iteration_median','ms_per_solving_iteration_mean','breakdown_ms')}, d['config']['linear_its_per_solve'], d['per_time_step'])"
  done
done
