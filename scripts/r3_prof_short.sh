#!/bin/bash
# kernel-stats profile of a few pipelined 'B' blocks (no torch): usage scripts/r3_prof_short.sh TAG
set -o pipefail
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
PROBE_DEPTH=6 timeout -k 10 300 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/r3_${TAG}_prof -o r3 -- python3 $ROOT/scripts/dev/pipe_notorch.py 10 > $ROOT/gpurun_out/r3_${TAG}_prof.log 2>&1 || { tail -5 $ROOT/gpurun_out/r3_${TAG}_prof.log; exit 1; }
python3 $ROOT/scripts/rocpd_stats.py $ROOT/gpurun_out/r3_${TAG}_prof/r3_results.db > $ROOT/gpurun_out/r3_${TAG}_stats.txt
grep "k_gm" $ROOT/gpurun_out/r3_${TAG}_stats.txt
