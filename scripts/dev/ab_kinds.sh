#!/bin/sh
# Development: the transform's device time on the three synthetic kinds, default build vs
# build_variants/<name>, alternating, on the same box.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for round in 1 2; do
  for v in default "$@"; do
    if [ "$v" = default ]; then unset BWTC_HIP_LIB; else export BWTC_HIP_LIB=$ROOT/build_variants/$v/libbwtc_hip.so; fi
    echo "== $v"
    python3 scripts/dev/bwt_kinds.py
  done
done
