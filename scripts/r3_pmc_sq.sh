#!/bin/bash
# Round-3 PMC pass: SQ instruction / wait counters per kernel of a few pipelined 'B' blocks.
# usage: scripts/r3_pmc_sq.sh TAG "COUNTER LIST"
set -o pipefail
TAG=${1:-sq}
CTRS=${2:-"SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PROBE_DEPTH=4 timeout -k 10 500 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT" -o p -- \
    python3 "$ROOT/scripts/dev/pipe_notorch.py" 5 > "$OUT/run.log" 2> "$OUT/run.err" || { tail -5 "$OUT/run.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-46:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen: seen.add(key); n[k] += 1
names = sorted({c for k in acc for c in acc[k]})
print("%-48s %5s " % ("kernel", "n") + " ".join("%14s" % c[-14:] for c in names))
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    print("%-48s %5d " % (k, n[k]) + " ".join("%14.4g" % (acc[k][c] / n[k]) for c in names))
PY
