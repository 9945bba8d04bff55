#!/bin/bash
# Run the emulator-backed parity suites with the product's host code (table.cpp, retain.cpp,
# match_core.hpp through tests/emu/emu.cpp) built with AddressSanitizer + UBSan.  CPU only.
#   bash tools/asan_emu.sh
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
SO=tests/emu/libemu.so
[ -f "$SO" ] && cp "$SO" /tmp/libemu_plain.so
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -I include -I rmqtt_amd/csrc \
    tests/emu/emu.cpp rmqtt_amd/csrc/table.cpp rmqtt_amd/csrc/retain.cpp -o "$SO"
trap '[ -f /tmp/libemu_plain.so ] && cp /tmp/libemu_plain.so "$SO" && touch "$SO"' EXIT
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
python -m pytest tests/test_parity.py tests/test_retain_parity.py tests/test_deliver_parity.py tests/test_snapshot.py \
    tests/test_golden_fixtures.py tests/test_hypothesis_parity.py tests/test_retain_tiers.py tests/test_publish_packets.py -x -q -m "not gpu" -p no:cacheprovider

# The Raft snapshot reader (host-only code in the Router mirror's library) under the same sanitizers:
# truncations, byte flips and hand-made compressed streams of tests/test_raft_snapshot.py.
HOST=rmqtt_amd/librmqtt_host_router.so
python -c "from rmqtt_amd import build; build.build_gpu(); build.build_host_router()"
cp "$HOST" /tmp/libhost_plain.so
g++ -O1 -g -std=c++17 -fPIC -shared -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -I include -I rmqtt_amd/host \
    rmqtt_amd/host/gpu_router.cpp rmqtt_amd/host/gpu_retain.cpp rmqtt_amd/host/raft_snapshot.cpp rmqtt_amd/host/router_capi.cpp \
    -o "$HOST" -L rmqtt_amd -lrmqtt_gpu_router -lz -ldl -Wl,-rpath,'$ORIGIN'
trap '[ -f /tmp/libemu_plain.so ] && cp /tmp/libemu_plain.so "$SO" && touch "$SO"; cp /tmp/libhost_plain.so "$HOST" && touch "$HOST"' EXIT
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
python -m pytest tests/test_raft_snapshot.py -x -q -m "not gpu" -p no:cacheprovider
