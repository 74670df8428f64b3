#!/bin/bash
# A/B of the pressure-stage knobs on the HEADLINE configuration (bench deck with its 5-spot, CPR in double + GMRES(40)): one bench.py --only-main
# run per setting, one line each (Mcell-updates/s, median ms, linear iterations).    gpurun -- 'bash tools/sweep_headline.sh > gpurun_out/sweep.log'
run() {
  env "$@" python bench.py --only-main --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('%-70s value %6.1f  ms %.3f  its %.2f  chopped %s' % ('$*', d['value'], d['ms_per_solving_iteration_median'], d['config'].get('linear_iterations_per_solving_iteration') or 0, d['config'].get('time_steps_chopped')), flush=True)
"
}
if [ "$1" = "adapt" ]; then       # the two-setting policy for the correction factor (OPMGPU_AMG_ADAPT), on against off
  rund() {
    tag="$1"; shift
    python bench.py --only-main --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
c=d['config']
print('%-44s value %6.1f  ms %.3f  mean-all %6.1f its %.2f  chopped %s notconv %s' % ('$tag', d['value'], d['ms_per_solving_iteration_median'], d['value_all_calls_mean'], c.get('linear_iterations_per_solving_iteration') or 0, c.get('time_steps_chopped'), c.get('time_steps_not_converged')), flush=True)
"
  }
  for ad in 1 0; do
  export OPMGPU_AMG_ADAPT=$ad
  rund "adapt=$ad cart100 (headline)"
  rund "adapt=$ad cart100 steps 60" --steps 60
  rund "adapt=$ad spe10like gmres" --deck spe10like
  rund "adapt=$ad spe10like bicgstab" --deck spe10like --krylov bicgstab
  rund "adapt=$ad 200^3" --nx 200 --ny 200 --nz 200
  rund "adapt=$ad cart100 bicgstab" --krylov bicgstab
  rund "adapt=$ad cart100 f32" --precision f32
  rund "adapt=$ad cart100 dt1" --dt-days 1
  rund "adapt=$ad 150^3" --nx 150 --ny 150 --nz 150
done
  exit 0
fi
if [ "$1" = "decks" ]; then       # the correction factor across decks and solver variants
  rund() {
    tag="$1"; shift
    python bench.py --only-main --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
c=d['config']
print('%-44s value %6.1f  ms %.3f  mean-all %6.1f its %.2f  chopped %s notconv %s' % ('$tag', d['value'], d['ms_per_solving_iteration_median'], d['value_all_calls_mean'], c.get('linear_iterations_per_solving_iteration') or 0, c.get('time_steps_chopped'), c.get('time_steps_not_converged')), flush=True)
"
  }
  for pd in default 2.2 2.4; do
    if [ $pd = default ]; then unset OPMGPU_AMG_PDAMP; else export OPMGPU_AMG_PDAMP=$pd; fi
    rund "pd=$pd spe10like gmres" --deck spe10like
    rund "pd=$pd spe10like bicgstab" --deck spe10like --krylov bicgstab
    rund "pd=$pd 200^3" --nx 200 --ny 200 --nz 200
    rund "pd=$pd cart100 no wells" --wells none
    rund "pd=$pd cart100 bicgstab" --krylov bicgstab
    rund "pd=$pd cart100 f32" --precision f32
    rund "pd=$pd cart100 dt1" --dt-days 1
  done
  exit 0
fi
if [ "$1" = "2" ]; then
  for pd in 2.0 2.2 2.4 2.6 3.0; do run OPMGPU_AMG_PDAMP=$pd; done
  run OPMGPU_AMG_PDAMP=2.2 OPMGPU_AMG_NPOST=1 OPMGPU_AMG_NPOST0=1
  run OPMGPU_AMG_PDAMP=2.4 OPMGPU_AMG_NPOST=1 OPMGPU_AMG_NPOST0=1
  run OPMGPU_AMG_PDAMP=2.2 OPMGPU_AMG_PDAMP0=1.9
  run OPMGPU_AMG_PDAMP=2.6 OPMGPU_AMG_PDAMP0=2.2
  exit 0
fi
run OPMGPU_NOP=1
for pd in 1.6 2.2; do run OPMGPU_AMG_PDAMP=$pd; done
for om in 0.8 1.0; do run OPMGPU_AMG_OMEGA=$om; done
run OPMGPU_AMG_NPOST=1
run OPMGPU_AMG_NPOST0=1
run OPMGPU_AMG_NPOST0=1 OPMGPU_AMG_NPOST=1
run OPMGPU_AMG_NPRE=2
run OPMGPU_AMG_NPOST0=3
run OPMGPU_AMG_PDAMP0=2.2
run OPMGPU_AMG_PDAMP0=1.6
run OPMGPU_NOP=2
