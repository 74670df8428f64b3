#!/bin/bash
# random irregular decks (tools/robust_sweep.py's generator) DECOMPOSED into slabs of j-rows: 2 ranks CPR + BiCGStab, 4 ranks CPR + GMRES, 60 Newton calls each,
# real ranks over the test transport; one GPU for comparison
export OPMGPU_COMM_TRANSPORT=shm
P=30000
one() { timeout -k 10 200 python bench.py --only-main --no-cpu-baseline --steps 60 --warmup 2 --stat-calls 60 --deck random --seed $1 --krylov $2 --detail gpurun_out/rs_detail.json 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['per_time_step']; print('  1 rank  $2: its/solve', d['config']['linear_its_per_solve'], 'steps', p['time_steps'], 'chopped', p['chopped_attempts'], 'lin/step', p['linear_iterations_per_time_step'])" || echo "  1 rank $2: FAILED"; }
many() { timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $((P++)) bench.py --gpus $1 --deck random --seed $2 --krylov $3 --steps 60 --warmup 2 --stat-calls 60 --no-cpu-baseline --detail gpurun_out/rs_detail.json 2>gpurun_out/rs_last.err \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['per_time_step']; print('  $1 ranks $3: its/solve', d['config']['linear_its_per_solve'], 'steps', p['time_steps'], 'chopped', p['chopped_attempts'], 'lin/step', p['linear_iterations_per_time_step'], 'cells/rank', d['config']['cells_per_gpu'])" || { echo "  $1 ranks $3: FAILED"; grep -v "amdgpu\|Warning" gpurun_out/rs_last.err | grep "Error\|error" | tail -3 | cut -c1-300; }; }
for seed in $(seq ${1:-8000} ${2:-8011}); do
  echo "== deck $seed"
  one $seed bicgstab
  many 2 $seed bicgstab
  many 4 $seed gmres
done
