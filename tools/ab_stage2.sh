#!/bin/bash
# stage-2 damping (cpr_stage2_relax 1.0 = the reference's form, 0.9 = rounds 1-3) on one GPU: headline deck at dt = 5 / 20 / 30 d, both Krylov methods
for dt in 5 30; do
  for kry in gmres bicgstab; do
    for rel in 1.0 0.9; do
      python bench.py --only-main --no-cpu-baseline --krylov $kry --dt-days $dt --stage2-relax $rel --steps 30 --stat-calls 60 --detail gpurun_out/ab_detail.json 2>/dev/null \
        | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['per_time_step']; print('dt $dt $kry stage2 $rel:', d['value'], d['config']['linear_its_per_solve'], 'newton/step', p['newton_iterations_per_time_step'], 'ms/step', p['ms_per_converged_time_step'], 'ms/day', p['ms_per_simulated_day'], 'chops', p['chopped_attempts'], 'steps', p['time_steps'])"
    done
  done
done
