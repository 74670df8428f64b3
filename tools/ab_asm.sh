#!/bin/bash
# A/B on ONE box: assembly entry layout (transposed slot byte / depth plane) and the early ILU0 factorisation start
for cfg in "" "OPMGPU_ASM_TSLOT=0" "OPMGPU_ASM_GDZ=1" "OPMGPU_ASM_TSLOT=0 OPMGPU_ASM_GDZ=1" "" "OPMGPU_ASM_TSLOT=0 OPMGPU_ASM_GDZ=1"; do
  env $cfg python tools/ab_kernels.py 100 2>/dev/null | cut -c1-260
done
for cfg in "OPMGPU_FACTOR_EARLY=1" "OPMGPU_FACTOR_EARLY=0" "OPMGPU_FACTOR_EARLY=1" "OPMGPU_FACTOR_EARLY=0"; do
  echo "== $cfg"
  env $cfg python bench.py --only-main --no-cpu-baseline --steps 40 --stat-calls 80 --detail gpurun_out/ab_detail.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_solving_iteration_median','ms_per_solving_iteration_mean','breakdown_ms')}, d['config']['linear_its_per_solve'], d['per_time_step'])"
  env $cfg python bench.py --only-main --no-cpu-baseline --krylov bicgstab --steps 40 --stat-calls 80 --detail gpurun_out/ab_detail.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_solving_iteration_median','ms_per_solving_iteration_mean','breakdown_ms')}, d['config']['linear_its_per_solve'], d['per_time_step'])"
done
