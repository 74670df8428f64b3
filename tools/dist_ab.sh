#!/bin/bash
# A/B of the decomposed GMRES path's fused operations over the shared-memory test transport (ranks share cuda:0): iteration counts and the
# collective-operation count per Newton iteration.  gpurun -- 'bash tools/dist_ab.sh'
export OPMGPU_COMM_TRANSPORT=shm
for cfg in "OPMGPU_CS_FUSED=1" "OPMGPU_CS_FUSED=0"; do
  for np in 2 4; do
    echo "== $cfg, $np ranks (strong, 40^3 + wells; SPE10-like for 2)"
    env $cfg timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $np --master-addr 127.0.0.1 --master-port $((29600 + np)) bench.py --gpus $np --nx 40 --ny 40 --nz 40 --scaling strong --steps 16 --warmup 2 --no-cpu-baseline --detail gpurun_out/dist_ab_detail.json 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['linear_its_per_solve'], d['per_time_step'])"
  done
  env $cfg timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29610 bench.py --gpus 2 --deck spe10like --steps 12 --warmup 2 --no-cpu-baseline --detail gpurun_out/dist_ab_detail.json 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('spe10like', d['value'], d['config']['linear_its_per_solve'], d['per_time_step'])"
  env $cfg python -m pytest tests/test_gpu_dist_shm.py -q -s -k "collective_operations or cpr_gmres" 2>&1 | grep -E "collective operations|passed|failed"
done
