#!/bin/bash
# finisher knobs on the real-text workloads: one line per setting
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for w in 2 3 4; do for g in "1024 256" "1024 512" "2048 512" "2048 1024"; do
  set -- $g
  echo -n "words=$w window=$1 group=$2: "
  BWTC_HIP_FIN_WORDS=$w BWTC_HIP_FIN_WINDOW=$1 BWTC_HIP_FIN_GROUP=$2 REPS=2 timeout -k 10 200 python3 $ROOT/scripts/r5/workloads.py 256 ${KINDS:-realtext pycorpus} 2>/dev/null | python3 -c "
import sys, json
print(' '.join('%s %.2f ms (rounds %d)' % (d['workload'], d['device_ms_bwt'], d['rounds']) for d in map(json.loads, sys.stdin)))"
done; done
