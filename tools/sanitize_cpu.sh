#!/bin/bash
# tools/sanitize_cpu.sh -- the CPU-side code under gcc's AddressSanitizer + UndefinedBehaviorSanitizer.
# CPU box only: GPU sanitizers are not available on the pool, and nothing here touches a GPU.
#   1. oracle/csmp_oracle.c + tools/sanitize/oracle_driver.c: every oracle entry point on small seeded problems, output buffers of
#      exactly the documented sizes, leak detection ON;
#   2. the same oracle as a sanitized shared object under the CPU test suites that exercise it (tests/test_oracle.py,
#      tests/test_host.py; LD_PRELOAD of the sanitizer runtime, leak detection off: the interpreter's own allocations are not ours);
#   3. the context-free exports of include/csmp.h (host/hostonly.hpp: dictionary files, the sharded gather's wire layout) as a
#      host-only object driven by tools/sanitize/hostonly_driver.cpp over tests/golden/dict_*.csmp.
# Exit status 0 = every part ran clean.  Usage: tools/sanitize_cpu.sh [log file]   (default: profiles/r06_sanitize_cpu.txt)
set -u
cd "$(dirname "$0")/.."
LOG="${1:-profiles/r06_sanitize_cpu.txt}"
OUT="$(mktemp -d /tmp/csmp_sanitize.XXXXXX)"
trap 'rm -rf "$OUT"' EXIT
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -g"
status=0
step() {  # step <title> <command...>
    local title="$1"; shift
    echo "== $title" | tee -a "$LOG"
    echo "   \$ $*" >> "$LOG"
    if "$@" >> "$LOG" 2>&1; then echo "   ok" | tee -a "$LOG"; else echo "   FAILED (exit $?)" | tee -a "$LOG"; status=1; fi
}
{
    echo "tools/sanitize_cpu.sh   $(date -u +%Y-%m-%dT%H:%M:%SZ)   $(gcc --version | head -1)"
    echo "flags: $SAN   (gcc -O1; ASAN_OPTIONS / UBSAN_OPTIONS as printed per step)"
} > "$LOG"

step "1a. build: oracle + its driver, sanitized" \
    gcc -std=c11 -O1 $SAN -fopenmp -mavx2 -mfma -Wall -Wextra -o "$OUT/oracle_driver" tools/sanitize/oracle_driver.c oracle/csmp_oracle.c -lm
step "1b. run: every oracle entry point, exactly-sized buffers, leak detection on" \
    env ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 OMP_NUM_THREADS=4 "$OUT/oracle_driver"

step "2a. build: the oracle as a sanitized shared object" \
    gcc -std=c11 -O1 $SAN -fopenmp -mavx2 -mfma -fPIC -shared -Wall -Wextra -o "$OUT/libcsmp_oracle_san.so" oracle/csmp_oracle.c -lm
ASAN_RT="$(gcc -print-file-name=libasan.so)"
UBSAN_RT="$(gcc -print-file-name=libubsan.so)"
step "2b. run: tests/test_oracle.py + tests/test_host.py against it (CSMP_ORACLE_SO; sanitizer runtime preloaded)" \
    env LD_PRELOAD="$ASAN_RT:$UBSAN_RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
        CSMP_ORACLE_SO="$OUT/libcsmp_oracle_san.so" OMP_NUM_THREADS=4 \
        python -m pytest tests/test_oracle.py tests/test_host.py -q -m "not gpu" -p no:cacheprovider

mkdir -p "$OUT/scratch"
step "3a. build: the context-free exports (host/hostonly.hpp) + their driver, sanitized" \
    g++ -std=c++17 -O1 $SAN -Wall -Wextra -o "$OUT/hostonly_driver" tools/sanitize/hostonly_driver.cpp
step "3b. run: dictionary files (tests/golden/dict_*.csmp, damaged copies, bad arguments), shard ranges, the gather's wire layout" \
    env ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 "$OUT/hostonly_driver" tests/golden "$OUT/scratch"

echo "== result: $([ $status -eq 0 ] && echo 'all parts clean' || echo 'FAILURES above')" | tee -a "$LOG"
exit $status
