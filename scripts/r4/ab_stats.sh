#!/bin/bash
# kernel statistics of a short 'B' stream under one environment switch per run: ab_stats.sh NAME=VALUE ...
# -> gpurun_out/ab/kernel_stats_<NAME>_<VALUE>.csv
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/ab
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do
  tag=$(echo "$kv" | tr '=' '_')
  env_name=${kv%%=*}; env_val=${kv#*=}
  export $env_name=$env_val
  PROBE_DEPTH=4 timeout -k 10 200 rocprofv3 --kernel-trace -d $O/s_$tag -o s -- python3 $ROOT/scripts/dev/pipe_notorch.py 6 > $O/$tag.log 2>&1 || exit 1
  unset $env_name
  python3 $ROOT/scripts/rocpd_stats.py $O/s_$tag/s_results.db $O/kernel_stats_$tag.csv || exit 1
  rm -rf $O/s_$tag
done
echo done
