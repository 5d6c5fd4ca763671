#!/bin/bash
# Runs the CPU tiers that execute the product's own C++ — the device headers compiled for the host (tests/emul)
# and the api.Verifier mirror (consensus_amd/host) — under AddressSanitizer + UndefinedBehaviorSanitizer.
# The sanitised builds replace the two .so files for the run and are restored afterwards.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
SAN="-O1 -g -fsanitize=undefined,address -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
PRE="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)"
python -c "import sys; sys.path.insert(0, 'tests'); import test_emul_device_algo" >/dev/null 2>&1 || true
cp tests/emul/libsbv_emul.so /tmp/libsbv_emul.so.keep 2>/dev/null || true
cp consensus_amd/libsbv_host.so /tmp/libsbv_host.so.keep
restore() {
  [ -f /tmp/libsbv_emul.so.keep ] && cp /tmp/libsbv_emul.so.keep tests/emul/libsbv_emul.so
  cp /tmp/libsbv_host.so.keep consensus_amd/libsbv_host.so
}
trap restore EXIT
g++ $SAN -std=c++17 -fPIC -shared -pthread -Wno-misleading-indentation -DSBV_F29_CHECK -DSBV_F25_CHECK -DSBV_K256_CHECK tests/emul/emul.cc -o tests/emul/libsbv_emul.so
( cd consensus_amd/host && g++ $SAN -std=c++17 -fPIC -Wall -Wno-misleading-indentation -pthread -shared p256_host.cc ed25519_host.cc k256_host.cc \
    formats.cc verifier.cc chain_emul.cc capi.cc -o ../libsbv_host.so -L.. -lsbv -Wl,-rpath,'$ORIGIN' )
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$PRE" python -m pytest tests/test_emul_device_algo.py tests/test_emul_fe29.py tests/test_ed25519_cpu.py \
    tests/test_host_verifier.py tests/test_datagen.py tests/test_emul_sign.py tests/test_chain_emul.py tests/test_k256_cpu.py -q
