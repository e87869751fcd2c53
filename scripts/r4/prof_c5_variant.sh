#!/bin/bash
# kernel statistics of the 1 GiB text block's transform under the environment given on the command line (NAME=VALUE ...)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c5v
mkdir -p "$OUT"
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/stats" -o s -- python3 "$ROOT/scripts/r4/bwt_only.py" 3 1024 t > "$OUT/stats.log" 2>&1 || { tail -5 "$OUT/stats.log"; exit 1; }
python3 "$ROOT/scripts/rocpd_stats.py" "$OUT/stats/s_results.db" "$OUT/kernel_stats.csv" || exit 1
rm -rf "$OUT/stats"
echo done
