set -e
export TMPDIR=/tmp
for v in "20000,400000" "0,0" "2000,400000" "20000,2000000"; do
  echo "== SUB=$v"
  OPMGPU_AMG_SUB=$v python3 bench.py --only-main --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['linear_iterations_per_newton'], d['breakdown_ms_per_step'])"
done
OPMGPU_AMG_TIME=1 OPMGPU_AMG_GRAPH=0 python3 bench.py --only-main --no-cpu-baseline 2>&1 >/dev/null | grep "\[amg\]"
