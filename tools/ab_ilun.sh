#!/bin/bash
# block ILU(n) (cpr_ilu_n / ilu_fillin_level) on the headline deck and on SPE10-like: iterations per solve and ms per Newton iteration
run() { echo "== $*"; timeout -k 10 500 python bench.py --only-main --no-cpu-baseline --steps 20 --warmup 3 --detail gpurun_out/ilun_detail.json "$@" 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['linear_its_per_solve'], d['per_time_step'])"; }
for deck in cart spe10like; do
  for n in 0 1 2; do
    run --ilu-fill $n --deck $deck --solver cpr --krylov gmres
    run --ilu-fill $n --deck $deck --solver cpr --krylov bicgstab
    run --ilu-fill $n --deck $deck --solver ilu0 --krylov bicgstab
  done
done
