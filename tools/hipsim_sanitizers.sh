#!/bin/bash
# The host simulation of the compact expansion kernels (tests/hipsim) under ThreadSanitizer and AddressSanitizer: GPU threads are OS
# threads there, so a missing barrier in the kernel SOURCE is a reported data race, and an access outside an allocation (arrays have
# their exact product sizes) a reported overflow.  CPU only.   tools/hipsim_sanitizers.sh > profiles/rNN_hipsim_sanitizers.txt
set -eu
cd "$(dirname "$0")/.."
CC=/opt/rocm/lib/llvm/bin/clang++
for san in thread address; do
  $CC -O1 -g -std=c++17 -pthread -fsanitize=$san -I include -I rmqtt_amd/csrc -I tests/hipsim tests/hipsim/tsan_main.cpp -o /tmp/hipsim_$san
  echo "# $CC -O1 -g -fsanitize=$san tests/hipsim/tsan_main.cpp  (rmqtt_amd/csrc/expand_compact.inc on the host)"
  TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" ASAN_OPTIONS="detect_leaks=0" /tmp/hipsim_$san > /tmp/hipsim_$san.log 2>&1 || echo "# exit status $?"
  grep -v "^$" /tmp/hipsim_$san.log | head -60
  echo "# sanitizer reports ($san): $(grep -c 'WARNING: ThreadSanitizer\|ERROR: AddressSanitizer' /tmp/hipsim_$san.log || true)"
  # (r5) the tuple / delivery expansions: expand_kernel<true>, the early-load and the lean variants (wave-private lists, last-wave count word)
  $CC -O1 -g -std=c++17 -pthread -fsanitize=$san -I include -I rmqtt_amd/csrc -I tests/hipsim tests/hipsim/tsan_tuple_main.cpp -o /tmp/hipsim_tuple_$san
  echo "# $CC -O1 -g -fsanitize=$san tests/hipsim/tsan_tuple_main.cpp  (rmqtt_amd/csrc/expand_tuple.inc on the host)"
  TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" ASAN_OPTIONS="detect_leaks=0" /tmp/hipsim_tuple_$san > /tmp/hipsim_tuple_$san.log 2>&1 || echo "# exit status $?"
  grep -v "^$" /tmp/hipsim_tuple_$san.log | head -60
  echo "# sanitizer reports, tuple expansions ($san): $(grep -c 'WARNING: ThreadSanitizer\|ERROR: AddressSanitizer' /tmp/hipsim_tuple_$san.log || true)"
  # (r6) the v5 dedup passes: tile pass + classification, the topic pass (double hashing; lists fetched 256 entries at a time)
  $CC -O1 -g -std=c++17 -pthread -fsanitize=$san -I include -I rmqtt_amd/csrc -I tests/hipsim tests/hipsim/tsan_dedup_main.cpp -o /tmp/hipsim_dedup_$san
  echo "# $CC -O1 -g -fsanitize=$san tests/hipsim/tsan_dedup_main.cpp  (rmqtt_amd/csrc/dedup.inc on the host)"
  TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" ASAN_OPTIONS="detect_leaks=0" /tmp/hipsim_dedup_$san > /tmp/hipsim_dedup_$san.log 2>&1 || echo "# exit status $?"
  grep -v "^$" /tmp/hipsim_dedup_$san.log | head -60
  echo "# sanitizer reports, dedup passes ($san): $(grep -c 'WARNING: ThreadSanitizer\|ERROR: AddressSanitizer' /tmp/hipsim_dedup_$san.log || true)"
done
