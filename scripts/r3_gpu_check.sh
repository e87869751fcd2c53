#!/bin/bash
# Round-3 GPU check: 'B' coder parity tests, a default bench line, a kernel-stats profile of a short run.
# usage: scripts/r3_gpu_check.sh TAG [pytest -k expression]
set -o pipefail
TAG=${1:-x}
KEXPR=${2:-wavelet}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_encode.py -x -q -k "$KEXPR" > gpurun_out/r3_${TAG}_tests.log 2>&1 || { tail -30 gpurun_out/r3_${TAG}_tests.log; exit 1; }
tail -2 gpurun_out/r3_${TAG}_tests.log
timeout -k 10 400 python bench.py --steps 48 --no-cpu-baseline > gpurun_out/r3_${TAG}_bench.json 2> gpurun_out/r3_${TAG}_bench.err || { tail -20 gpurun_out/r3_${TAG}_bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3_${TAG}_prof -o r3 -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3_${TAG}_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3_${TAG}_prof.err || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/r3_${TAG}_prof.err; exit 1; }
echo done
