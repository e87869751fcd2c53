#!/bin/sh
# Round-2 PMC passes over the whole bench step (BWT + 'B' device half), for the stream kernels'
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 runs, kernel trace only.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_prof_wavelet
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o f -- \
    python3 "$ROOT/bench.py" --steps 2 --warmup 1 --depth 4 --blocks 1 --no-cpu-baseline > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o w -- \
    python3 "$ROOT/bench.py" --steps 2 --warmup 1 --depth 4 --blocks 1 --no-cpu-baseline > /dev/null 2> "$OUT/write.err"
ls "$OUT/fetch" "$OUT/write"
