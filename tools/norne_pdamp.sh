#!/bin/bash
# Norne-like deck (60 % random inactive cells, NNCs, 36 wells), 120 days through the adaptive time stepper: the scaling of the pressure stage's
# coarse-grid corrections (OPMGPU_AMG_PDAMP; default 1.9 / 2.3 by the per-time-step policy) against failed sub-steps
for cfg in "A=0" "OPMGPU_AMG_ADAPT=0" "OPMGPU_AMG_PDAMP=1.6" "OPMGPU_AMG_PDAMP=1.3" "OPMGPU_AMG_PDAMP=1.0"; do
  for c in cpr_bicgstab cpr_gmres; do
    env $cfg timeout -k 10 300 python tools/long_run.py --deck nornelike --days 120 --configs $c 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('cpr'):
        n, j = l.split(' ', 1); d = json.loads(j)
        print('$cfg', n, 'substeps', d['substeps'], 'failed', d['failed_substeps'], 'wall', d['wall_s'], 'newton', d['newton_iterations'], 'linear', d['linear_iterations'])"
  done
done
