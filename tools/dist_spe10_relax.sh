#!/bin/bash
export OPMGPU_COMM_TRANSPORT=shm
run() { echo "== $*"; env $1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $2 --master-addr 127.0.0.1 --master-port $3 bench.py --gpus $2 --deck spe10like --steps 12 --warmup 2 --no-cpu-baseline --detail gpurun_out/dist_spe10_detail.json ${@:4} 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['linear_its_per_solve'], d['config']['cells_per_gpu'], d['per_time_step'])"; }
run X=0 4 29731 --stage2-relax 0.9
run X=0 4 29732 --stage2-relax 0.8
run X=0 4 29733 --stage2-relax 0.9 --krylov bicgstab
run X=0 2 29734 --stage2-relax 0.8
run X=0 2 29735 --stage2-relax 0.9 --dt-days 2
run X=0 4 29736 --stage2-relax 0.9 --dt-days 2
