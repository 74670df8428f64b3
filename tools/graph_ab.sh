#!/bin/bash
# V-cycle through a captured hipGraph (OPMGPU_AMG_GRAPH=1) against plain launches at the size one rank of the 8-GPU SPE10 leg holds
# (60 x 28 x 85 = 142 800 cells) and at 100^3
for dims in "60 28 85" "100 100 100"; do
  set -- $dims
  for cfg in "OPMGPU_AMG_GRAPH=0" "OPMGPU_AMG_GRAPH=1" "OPMGPU_AMG_GRAPH=0" "OPMGPU_AMG_GRAPH=1"; do
    for kry in gmres bicgstab; do
      env $cfg python bench.py --only-main --no-cpu-baseline --nx $1 --ny $2 --nz $3 --rate 100 --krylov $kry --steps 40 --stat-calls 40 --detail gpurun_out/graph_ab_detail.json 2>/dev/null \
        | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$dims $cfg $kry', d['value'], d['ms_per_solving_iteration_median'], d['ms_per_solving_iteration_mean'], d['config']['linear_its_per_solve'], d['breakdown_ms'])"
    done
  done
done
