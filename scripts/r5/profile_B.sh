#!/bin/bash
# Round-5 profiles of the pipelined 'B' route: kernel statistics of a 12-block stream, and two PMC passes
# (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only) of a 4-block stream -> gpurun_out/r05_prof_B/.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_prof_B
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PROBE_DEPTH=6 timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/stats" -o s -- python3 "$ROOT/scripts/dev/pipe_notorch.py" 12 > "$OUT/stats.log" 2>&1 || { tail -5 "$OUT/stats.log"; exit 1; }
python3 "$ROOT/scripts/rocpd_stats.py" "$OUT/stats/s_results.db" "$OUT/kernel_stats.csv" || exit 1
PROBE_DEPTH=4 timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o f -- python3 "$ROOT/scripts/dev/pipe_notorch.py" 4 > "$OUT/fetch.log" 2>&1 || { tail -5 "$OUT/fetch.log"; exit 1; }
PROBE_DEPTH=4 timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o w -- python3 "$ROOT/scripts/dev/pipe_notorch.py" 4 > "$OUT/write.log" 2>&1 || { tail -5 "$OUT/write.log"; exit 1; }
python3 "$ROOT/scripts/pmc_summary.py" "$OUT/fetch" "$OUT/write" "$OUT/pmc_traffic.json" "256 MiB text blocks, BWT (code-key long sort + finisher) + 'B' coder with the models on the device, 4 blocks" > "$OUT/pmc_traffic.txt" || exit 1
rm -rf "$OUT/stats"
echo done
