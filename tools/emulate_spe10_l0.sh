for cfg in "OPMGPU_EMULATE_RANKS=4 OPMGPU_EMULATE_WHAT=2" "OPMGPU_EMULATE_RANKS=4 OPMGPU_EMULATE_WHAT=2 OPMGPU_EMULATE_L0_GLOBAL=1" "OPMGPU_EMULATE_RANKS=4 OPMGPU_EMULATE_WHAT=3 OPMGPU_EMULATE_L0_GLOBAL=1" "OPMGPU_EMULATE_RANKS=8 OPMGPU_EMULATE_WHAT=3 OPMGPU_EMULATE_L0_GLOBAL=1" "OPMGPU_EMULATE_RANKS=8 OPMGPU_EMULATE_WHAT=3"; do
  for extra in "" "--stage2-relax 0.9"; do
    echo "== $cfg $extra"
    env $cfg python bench.py --deck spe10like --only-main --no-cpu-baseline --krylov bicgstab --steps 12 --warmup 2 --stat-calls 40 $extra --detail gpurun_out/emu_detail.json 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['linear_its_per_solve'], d['per_time_step'])"
  done
done
