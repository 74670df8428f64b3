#!/bin/bash
# One measurement campaign on the GPU box: the bench line, the same command under rocprofv3 --kernel-trace --stats, a short kernel TRACE of
# the headline alone (timeline of one Newton iteration), and the PMC passes (HBM bytes, L2 hit rate) over the assembly kernels, the SpMV
# and the pressure stage's set-up kernels -- one counter group per pass, --kernel-trace only, as the pool requires.
#   gpurun -- 'bash tools/profile_round.sh r04_a'      results under gpurun_out/<tag>/ (raw traces of the long run are dropped: gpurun
#   copies back at most 64 MiB)
set -e -o pipefail
TAG=${1:-prof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --detail $OUT/bench_detail.json > $OUT/bench_line.json 2> $OUT/bench.err
echo "bench done"; tail -c 600 $OUT/bench_line.json; echo
cd /tmp
# (the profiled command leaves out the CPU baseline and the other decks: the per-kernel averages then belong to the 100^3 deck alone, which is what
# bench.py's roofline object times -- with the SPE9-like / Norne-like legs in, k_spmv's average is a mix of 9 000-cell and 1 M-cell launches)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 $ROOT/bench.py --no-cpu-baseline --no-other-decks --detail $OUT/bench_detail_under_rocprof.json > $OUT/bench_line_under_rocprof.json 2> $OUT/rocprof.err
find $OUT/trace -name 'run_kernel_trace.csv' -delete          # hundreds of MB for the whole run; the per-kernel statistics stay
echo "stats done"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_main -o run -- python3 $ROOT/bench.py --only-main --no-cpu-baseline --stat-calls 20 --detail $OUT/bench_detail_trace_main.json > $OUT/bench_line_trace_main.json 2> $OUT/rocprof_main.err
echo "trace done"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $grp | tr ' ' '_')
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$name -o run -- python3 $ROOT/tools/pmc_assembly.py 100 > /dev/null 2> $OUT/pmc_$name.err
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmcamg_$name -o run -- python3 $ROOT/tools/pmc_amg_setup.py > /dev/null 2> $OUT/pmcamg_$name.err
    echo "pmc $name done"
done
cd $ROOT
python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.json
cat $OUT/pmc_summary.json
du -sh $OUT
