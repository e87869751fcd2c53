#!/bin/bash
# C5 (1 GiB single text block, transform alone): kernel statistics and the two PMC passes -> gpurun_out/r04_prof_c5/
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_prof_c5
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/stats" -o s -- python3 "$ROOT/scripts/r4/bwt_only.py" 4 1024 t > "$OUT/stats.log" 2>&1 || { tail -5 "$OUT/stats.log"; exit 1; }
python3 "$ROOT/scripts/rocpd_stats.py" "$OUT/stats/s_results.db" "$OUT/kernel_stats.csv" || exit 1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o f -- python3 "$ROOT/scripts/r4/bwt_only.py" 2 1024 t > "$OUT/fetch.log" 2>&1 || { tail -5 "$OUT/fetch.log"; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o w -- python3 "$ROOT/scripts/r4/bwt_only.py" 2 1024 t > "$OUT/write.log" 2>&1 || { tail -5 "$OUT/write.log"; exit 1; }
python3 "$ROOT/scripts/pmc_summary.py" "$OUT/fetch" "$OUT/write" "$OUT/pmc_traffic.json" "C5: 1 GiB text block, transform alone (long-key sort with 16-byte items + finisher + text rounds), 2 blocks" > "$OUT/pmc_traffic.txt" || exit 1
rm -rf "$OUT/stats"
tail -1 "$OUT/stats.log"
echo done
