set -o pipefail
D=/tmp/checked_$$; mkdir -p $D/csrc $D/lib $D/include
cp -r $GRAFT_REPO_ROOT/bwtc_amd/csrc/. $D/csrc/ && cp -r $GRAFT_REPO_ROOT/include/. $D/include/
sed -i "s|-I../../include|-I$D/include|; s|../../include/bwtc_hip.h|$D/include/bwtc_hip.h|" $D/csrc/Makefile
make -C $D/csrc -j16 EXTRA=-DBWTC_RADIX_CHECKED OUTDIR=$D/lib > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
BWTC_HIP_LIB=$D/lib/libbwtc_hip.so timeout -k 10 700 python -m pytest tests/test_gpu_bwt.py -x -q -m gpu 2>&1 | tail -3
