#!/bin/sh
# Round-2 profiles on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench (BWT + 'B' coder), CSV
#   2. two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only) over the
#      transform alone (bench.py --bwt-only), for scripts/pmc_summary.py
# Outputs land in gpurun_out/r02_prof/.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_prof
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o b -- \
    python3 "$ROOT/bench.py" --steps 8 --warmup 1 --depth 4 --no-cpu-baseline > "$OUT/bench_profiled.json" 2> "$OUT/stats.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o f -- \
    python3 "$ROOT/bench.py" --bwt-only --steps 2 --warmup 1 --blocks 1 --no-cpu-baseline > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o w -- \
    python3 "$ROOT/bench.py" --bwt-only --steps 2 --warmup 1 --blocks 1 --no-cpu-baseline > /dev/null 2> "$OUT/write.err"
ls -R "$OUT" | head -30
