#!/bin/bash
# Round-5 profiles kept under profiles/r05_*: kernel statistics of the transform alone on the three texts and the 1 GiB
# block, the 'B' route's kernel statistics and PMC traffic (two passes, kernel trace only) -> gpurun_out/r05_final/
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_final
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for spec in "c3 256 6" "realtext 256 3" "pycorpus 256 3" "c3 1024 3"; do
  set -- $spec
  REPS=$3 timeout -k 10 420 rocprofv3 --kernel-trace -d "$OUT/t_$1_$2" -o s -- python3 "$ROOT/scripts/r5/workloads.py" $2 $1 > "$OUT/bwt_$1_$2.log" 2>&1 || { tail -5 "$OUT/bwt_$1_$2.log"; exit 1; }
  python3 "$ROOT/scripts/rocpd_stats.py" "$OUT/t_$1_$2/s_results.db" "$OUT/kernel_stats_bwt_only_$1_$2.csv" || exit 1
  rm -rf "$OUT/t_$1_$2"
done
"$ROOT/scripts/r5/profile_B.sh" || exit 1
cp "$ROOT"/gpurun_out/r05_prof_B/kernel_stats.csv "$OUT/kernel_stats_B_default.csv"
cp "$ROOT"/gpurun_out/r05_prof_B/pmc_traffic.json "$OUT/pmc_traffic_text256.json"
cp "$ROOT"/gpurun_out/r05_prof_B/pmc_traffic.txt "$OUT/pmc_traffic_text256.txt"
echo done
