#!/bin/bash
# Run GPU steps one after the other on the GPU box; every step under its own `timeout -k 10`.  A step that FAILS (assertion, non-zero exit)
# does not stop the sequence, a step that is KILLED (timeout: 124 / 137) does -- no further GPU step after a kill.
#   gpurun -- 'bash tools/gpu_steps.sh "300 cmd1" "600 cmd2" ...'        (first word = the step's time limit in seconds)
mkdir -p gpurun_out
n=0
for step in "$@"; do
    n=$((n + 1))
    lim=${step%% *}; cmd=${step#* }
    echo "== step $n (limit ${lim}s): $cmd"
    timeout -k 10 "$lim" bash -o pipefail -c "$cmd"
    rc=$?
    echo "== step $n exit code $rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== step $n was killed at its limit: stopping"; exit $rc; fi
done
exit 0
