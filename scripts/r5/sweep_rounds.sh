#!/bin/bash
# rounds of sixteen characters per finisher pass, after "every round's characters read in one go": one line per setting
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r05_sweep
for r in ${ROUNDS:-3 4 2}; do
  echo -n "fin_rounds=$r: "
  BWTC_HIP_FIN_ROUNDS=$r REPS=3 timeout -k 10 240 python3 $ROOT/scripts/r5/workloads.py 256 ${KINDS:-c3 realtext pycorpus} 2>$ROOT/gpurun_out/r05_sweep/rounds_$r.err | tee $ROOT/gpurun_out/r05_sweep/rounds_$r.jsonl | python3 -c "
import sys, json
print(' '.join('%s %.2f ms (rounds %d)' % (d['workload'], d['device_ms_bwt'], d['rounds']) for d in map(json.loads, sys.stdin)))" || exit 1
done
