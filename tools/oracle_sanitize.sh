#!/bin/bash
# The checker under AddressSanitizer + UBSan (CPU only; GPU sanitizers are not available on this pool): oracle/apus_oracle.c is
# rebuilt instrumented IN PLACE, the CPU tests that drive it without torch run under the preloaded runtimes, the normal build is put
# back.  (tests/test_oracle_vs_refloops.py is left out: the reference's loops are loaded once per server with dlmopen, which the
# ASan runtime aborts on -- a property of that harness, not a finding.)
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
make -C oracle liboracle.so -B CFLAGS="-O1 -g -fPIC -Wall -Wextra -std=gnu11 -fsanitize=address,undefined -fno-omit-frame-pointer" > /dev/null || exit 1
LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_oracle_vs_ref.py tests/test_trace_oracle.py tests/test_snapshot_stream.py tests/test_members_host.py \
  -q -s -m "not gpu" -p no:cacheprovider      # (-s: a report is printed where it can be seen, then the run stops)
rc=$?
make -C oracle liboracle.so -B > /dev/null
exit $rc
