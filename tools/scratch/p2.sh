set -e
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r02_b; mkdir -p $OUT
OPMGPU_AMG_TIME=1 OPMGPU_AMG_GRAPH=0 python3 bench.py --only-main --no-cpu-baseline > $OUT/amg_time.json 2> $OUT/amg_time.err
grep "\[amg\]" $OUT/amg_time.err | head -40
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 $ROOT/bench.py --only-main --no-cpu-baseline > $OUT/line.json 2> $OUT/rocprof.err
echo done
