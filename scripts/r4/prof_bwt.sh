#!/bin/bash
# kernel statistics of the transform alone (256 MiB text block): gpurun_out/r04_prof/bwt_kernel_stats.csv
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/bwt" -o s -- python3 "$ROOT/scripts/r4/bwt_only.py" ${1:-6} ${2:-256} ${3:-t} > "$OUT/bwt_${3:-t}.log" 2>&1 || { tail -5 "$OUT/bwt_${3:-t}.log"; exit 1; }
python3 "$ROOT/scripts/rocpd_stats.py" "$OUT/bwt/s_results.db" "$OUT/bwt_kernel_stats_${3:-t}.csv" || exit 1
rm -rf "$OUT/bwt"
tail -2 "$OUT/bwt_${3:-t}.log"
