#!/bin/sh
# Development: average duration of the kernels matching a pattern, default build vs build_variants/<name>
# (rocprofv3 kernel stats of a short bench run each).  usage: ab_kernel.sh <pattern> <variant>...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PAT=$1; shift
cd /tmp; export TMPDIR=/tmp
for v in default "$@"; do
  if [ "$v" = default ]; then unset BWTC_HIP_LIB; else export BWTC_HIP_LIB=$ROOT/build_variants/$v/libbwtc_hip.so; fi
  rm -rf /tmp/abk_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk_$v -o t -- python3 $ROOT/bench.py --steps 4 --warmup 1 --depth 4 --no-cpu-baseline > /dev/null 2>&1
  echo "== $v"
  python3 - "$PAT" /tmp/abk_$v/t_kernel_stats.csv <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[2])):
    if sys.argv[1] in r["Name"]:
        print("%-70s calls %4s avg %9.1f us" % (re.sub(r"[(].*", "", r["Name"])[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
