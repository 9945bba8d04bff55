#!/bin/bash
# The host simulation of the compact expansion kernels (tests/hipsim) under ThreadSanitizer: GPU threads are OS threads there, so a
# missing barrier in the kernel SOURCE is a reported data race.  CPU only.   tools/hipsim_tsan.sh > profiles/rNN_hipsim_tsan.txt
set -eu
cd "$(dirname "$0")/.."
CC=/opt/rocm/lib/llvm/bin/clang++
$CC -O1 -g -std=c++17 -pthread -fsanitize=thread -I include -I rmqtt_amd/csrc -I tests/hipsim tests/hipsim/tsan_main.cpp -o /tmp/hipsim_tsan
echo "# $CC -O1 -g -fsanitize=thread tests/hipsim/tsan_main.cpp  (rmqtt_amd/csrc/expand_compact.inc on the host)"
TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" /tmp/hipsim_tsan 2>&1 | tee /tmp/hipsim_tsan.log | grep -v "^$" | head -80
echo "# ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' /tmp/hipsim_tsan.log || true)"
