#!/bin/bash
# SPE10-like deck on 2 / 4 real ranks (shared-memory transport, all on cuda:0): coarse unknowns per rank of the decomposed pressure stage as
# sub-slabs along the cut direction (OPMGPU_COARSE_SUBSLABS: they keep the vertical wells whole), against the default one unknown per rank
export OPMGPU_COMM_TRANSPORT=shm
run() { echo "== $*"; env $1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $2 --master-addr 127.0.0.1 --master-port $3 bench.py --gpus $2 --deck spe10like --steps 12 --warmup 2 --no-cpu-baseline --detail gpurun_out/dist_spe10_detail.json ${@:4} 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['linear_its_per_solve'], d['config']['cells_per_gpu'], d['per_time_step'])"; }
run OPMGPU_COARSE_SUBSLABS=0 2 29721
run OPMGPU_COARSE_SUBSLABS=4 2 29722
run OPMGPU_COARSE_SUBSLABS=6 2 29723
run OPMGPU_COARSE_SUBSLABS=0 4 29724
run OPMGPU_COARSE_SUBSLABS=4 4 29725
run OPMGPU_COARSE_SUBSLABS=4 4 29726 --krylov bicgstab
run OPMGPU_COARSE_SUBSLABS=4 4 29727 --stage2-relax 0.9
