#!/bin/bash
# linear_solver_reduction against the cost of a TIME STEP (headline deck, dt = 5 d and 20 d): dune's left-preconditioned GMRES stops on ||M^-1 r||,
# so its 1e-2 is a looser solve than BiCGStab's 1e-2 and the Newton loop pays for it
run() { timeout -k 10 400 python bench.py --only-main --no-cpu-baseline --steps 20 --warmup 3 --stat-calls 80 --detail gpurun_out/red_detail.json "$@" 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['per_time_step']; print('$*', '| Mcell-upd/s', d['value'], '| its/solve', d['config']['linear_its_per_solve'], '| newton/step', p['newton_iterations_per_time_step'], '| ms/step', p['ms_per_converged_time_step'], '| ms/day', p['ms_per_simulated_day'], '| chopped', p['chopped_attempts'])"; }
for dt in 5 20; do
  for kry in gmres bicgstab; do
    for red in 1e-2 3e-3 1e-3 1e-4; do
      run --krylov $kry --reduction $red --dt-days $dt
    done
  done
done
