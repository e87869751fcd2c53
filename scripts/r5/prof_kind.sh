#!/bin/bash
# kernel statistics of the transform alone on one workload: gpurun_out/r05_prof/bwt_kernel_stats_<kind>.csv
# usage: prof_kind.sh kind [MiB] [reps]
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_prof
KIND=${1:-realtext}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
REPS=${3:-3} timeout -k 10 420 rocprofv3 --kernel-trace -d "$OUT/bwt_$KIND" -o s -- python3 "$ROOT/scripts/r5/workloads.py" ${2:-256} $KIND > "$OUT/bwt_$KIND.log" 2>&1 || { tail -5 "$OUT/bwt_$KIND.log"; exit 1; }
python3 "$ROOT/scripts/rocpd_stats.py" "$OUT/bwt_$KIND/s_results.db" "$OUT/bwt_kernel_stats_$KIND.csv" || exit 1
rm -rf "$OUT/bwt_$KIND"
tail -2 "$OUT/bwt_$KIND.log"
