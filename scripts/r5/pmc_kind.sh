#!/bin/bash
# PMC passes over the transform alone on one workload (separate runs, kernel trace only):
# HBM traffic per kernel and the SQ counters that say what a kernel waits for -> gpurun_out/r05_pmc_<kind>/
# usage: pmc_kind.sh kind [MiB]
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
KIND=${1:-realtext}
OUT=$ROOT/gpurun_out/r05_pmc_$KIND
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {   # name counters...
  local name=$1; shift
  REPS=1 timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- python3 "$ROOT/scripts/r5/workloads.py" ${MIB:-256} $KIND > "$OUT/$name.log" 2>&1 || { tail -5 "$OUT/$name.log"; exit 1; }
}
MIB=${2:-256}
pass fetch FETCH_SIZE || exit 1
pass write WRITE_SIZE || exit 1
python3 "$ROOT/scripts/pmc_summary.py" "$OUT/fetch" "$OUT/write" "$OUT/pmc_traffic.json" "$MIB MiB, $KIND, transform alone, 1 warm-up + 1 block" > "$OUT/pmc_traffic.txt" || exit 1
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES || exit 1
pass sq2 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE || exit 1
for d in sq1 sq2; do cp "$OUT"/$d/*/*counter_collection.csv "$OUT/$d.csv" 2>/dev/null || cp "$OUT"/$d/*counter_collection.csv "$OUT/$d.csv"; done
rm -rf "$OUT/fetch" "$OUT/write" "$OUT/sq1" "$OUT/sq2"
echo done
