#!/bin/bash
# one line per environment setting: the transform alone on the given workloads
# usage: sweep_env.sh "kinds" "VAR=val[,VAR2=val2]" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
KINDS=$1; shift
for cfg in "$@"; do
  echo -n "[$cfg] "
  env $(echo "$cfg" | tr ',' ' ') REPS=${REPS:-3} timeout -k 10 240 python3 $ROOT/scripts/r5/workloads.py 256 $KINDS 2>/dev/null | python3 -c "
import sys, json
print(' '.join('%s %.2f ms (rounds %d)' % (d['workload'], d['device_ms_bwt'], d['rounds']) for d in map(json.loads, sys.stdin)))" || exit 1
done
