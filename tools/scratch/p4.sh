set -e
export TMPDIR=/tmp
for k in bicgstab gmres; do
  echo "== $k"
  python3 bench.py --only-main --no-cpu-baseline --krylov $k 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['linear_iterations_per_newton'], d['breakdown_ms_per_step'], d['config']['time_steps_not_converged'])"
done
