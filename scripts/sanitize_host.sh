#!/bin/sh
# Host code of the 'B' coder under the sanitizers (CPU build only; there is no GPU sanitizer on
# this pool): the pipeline test with the library's host sources compiled in, under
# ThreadSanitizer and under AddressSanitizer + UBSan, and the decoder test under ASan + UBSan.
# Needs oracle/liboracle.so and bwtc_amd/lib/libbwtc_hip.so (python -c "import __graft_entry__ as g; g.build()").
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/bwtc_amd/csrc
OUT=${TMPDIR:-/tmp}/bwtc_sanitize
mkdir -p "$OUT"
HOSTSRC="$SRC/wavelet_pipeline.cpp $SRC/wavelet_host.cpp $SRC/wavelet_simd.cpp $SRC/wavelet_rc.cpp $SRC/entropy_host.cpp"
LINK="-L$ROOT/oracle -loracle -lpthread -Wl,-rpath,$ROOT/oracle"
for SAN in thread address,undefined; do
  g++ -O1 -g -std=c++17 -fsanitize=$SAN -fno-sanitize-recover=all -I"$ROOT/include" -I"$SRC" \
      -o "$OUT/host_pipeline_$SAN" "$ROOT/tests/cpp/host_pipeline_test.cpp" $HOSTSRC $LINK
  echo "== host_pipeline_test under -fsanitize=$SAN"
  "$OUT/host_pipeline_$SAN"
done
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=all -I"$ROOT/include" -I"$ROOT/bwtc_amd/host" \
    -o "$OUT/wavelet_decoder_asan" "$ROOT/tests/cpp/wavelet_decoder_test.cpp" -L"$ROOT/bwtc_amd/lib" -lbwtc_hip $LINK \
    -Wl,-rpath,"$ROOT/bwtc_amd/lib"
echo "== wavelet_decoder_test under -fsanitize=address,undefined"
"$OUT/wavelet_decoder_asan"
