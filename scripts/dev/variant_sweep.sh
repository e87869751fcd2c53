#!/bin/sh
# Development: benches other builds of the library (build_variants/<name>/libbwtc_hip.so, made with
# `make -C bwtc_amd/csrc OUTDIR=../../build_variants/<name> EXTRA=-D...`) beside the default one.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/sweep
for v in default "$@"; do
  if [ "$v" = default ]; then unset BWTC_HIP_LIB; else export BWTC_HIP_LIB=$ROOT/build_variants/$v/libbwtc_hip.so; fi
  python3 $ROOT/bench.py --bwt-only --steps 5 --warmup 1 --blocks 2 --no-cpu-baseline > $ROOT/gpurun_out/sweep/$v.bwt.json 2> $ROOT/gpurun_out/sweep/$v.bwt.err || exit 1
  python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/sweep/$v.full.json 2> $ROOT/gpurun_out/sweep/$v.full.err || exit 1
  python3 - "$v" $ROOT/gpurun_out/sweep/$v.bwt.json $ROOT/gpurun_out/sweep/$v.full.json <<'PY'
import json, sys
b = json.load(open(sys.argv[2])); f = json.load(open(sys.argv[3]))
print("%-10s bwt %.2f ms   full %.2f ms/step (bwt part %.2f)" % (sys.argv[1], b["ms_per_step"], f["ms_per_step"], f["device_ms_bwt"]), flush=True)
PY
done
