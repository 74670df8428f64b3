set -e
export TMPDIR=/tmp
for env in "A=1" "OPMGPU_GMRES_LAG=0" "OPMGPU_POLL=0"; do
  echo "== $env"
  env $env python3 bench.py --only-main --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['linear_iterations_per_newton'], d['breakdown_ms_per_step'], d['config']['time_steps_not_converged'], d['config']['linear_solver'])"
done
