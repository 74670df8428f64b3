#!/bin/bash
# mixed-precision preconditioner (opmgpu_params.preconditioner_single) against the pure double solve, same box; float copy written by the
# assembly (default) or converted before the solve (OPMGPU_MIXED_DUALWRITE=0)
python -m pytest tests/test_gpu_linsolver.py tests/test_gpu_wells.py tests/test_gpu_assembly.py -x -q 2>&1 | tail -n 3
for dw in 1 0; do
  echo "== OPMGPU_MIXED_DUALWRITE=$dw"
  OPMGPU_MIXED_DUALWRITE=$dw python bench.py --no-cpu-baseline --no-other-decks --steps 40 --stat-calls 80 --detail gpurun_out/ab_mixed_detail.json 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(len(json.dumps(d)), d['value'], d['breakdown_ms'], d['per_time_step']); [print(k, v) for k, v in d['variants'].items() if 'precond' in k or k in ('cpr_f64_bicgstab', 'cpr_f32_gmres')]"
done
python tools/ab_kernels.py 100 2>/dev/null | cut -c1-300
