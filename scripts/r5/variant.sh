#!/bin/bash
# builds the library with extra compiler flags into /tmp on the GPU box and times the three texts with it
# usage: variant.sh "<flags>" [workloads...]     e.g. variant.sh "-DBWTC_RADIX_TPB=1024" c3 realtext
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
FLAGS=$1; shift
KINDS=${@:-c3 realtext pycorpus}
D=/tmp/variant_$$
mkdir -p $D/csrc $D/lib
cp -r $ROOT/bwtc_amd/csrc/. $D/csrc/ && mkdir -p $D/include && cp -r $ROOT/include/. $D/include/
sed -i "s|-I../../include|-I$D/include|; s|../../include/bwtc_hip.h|$D/include/bwtc_hip.h|" $D/csrc/Makefile
make -C $D/csrc -j16 EXTRA="$FLAGS" OUTDIR=$D/lib > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
echo -n "variant [$FLAGS]: "
BWTC_HIP_LIB=$D/lib/libbwtc_hip.so REPS=3 timeout -k 10 400 python3 $ROOT/scripts/r5/workloads.py 256 $KINDS 2>$D/err.log | python3 -c "
import sys, json
print(' '.join('%s %.2f' % (d['workload'], d['device_ms_bwt']) for d in map(json.loads, sys.stdin)))" || { tail -5 $D/err.log; exit 1; }
# PROF=<kind>: kernel statistics of the variant on that workload -> gpurun_out/r05_prof/bwt_kernel_stats_<kind>.csv
if [ -n "$PROF" ]; then BWTC_HIP_LIB=$D/lib/libbwtc_hip.so bash $ROOT/scripts/r5/prof_kind.sh $PROF > /dev/null && grep -E "k_finish" $ROOT/gpurun_out/r05_prof/bwt_kernel_stats_$PROF.csv | awk -F, '{printf "  %s calls %s total %.2f ms max %.2f ms\n", substr($1,1,50), $2, $3/1e6, $7/1e6}'; fi
