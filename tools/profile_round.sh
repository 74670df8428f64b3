#!/bin/bash
# One measurement campaign on the GPU box: the bench line, the same command under rocprofv3 --kernel-trace --stats, and the PMC passes
# (HBM bytes, L2 hit rate) over the assembly kernels and the SpMV -- one counter group per pass, --kernel-trace only, as the pool requires.
#   gpurun -- 'bash tools/profile_round.sh r02_a'      results under gpurun_out/<tag>/
set -e -o pipefail
TAG=${1:-prof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench.err
echo "bench done"; tail -c 600 $OUT/bench_line.json; echo
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_line_under_rocprof.json 2> $OUT/rocprof.err
echo "trace done"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $grp | tr ' ' '_')
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$name -o run -- python3 $ROOT/tools/pmc_assembly.py 100 > /dev/null 2> $OUT/pmc_$name.err
    echo "pmc $name done"
done
cd $ROOT
python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.json
cat $OUT/pmc_summary.json
ls $OUT/trace | head
