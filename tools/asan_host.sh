#!/bin/bash
# The HOST side of libnlx.so / libnlx_synth.so under AddressSanitizer + UndefinedBehaviorSanitizer (the device side cannot be:
# no GPU sanitizers on the pool), then the CPU test suite through that build.  Run on the CPU box:
#   bash tools/asan_host.sh            -> profiles/r04_asan_host.txt
# Left out: the two tests that force a failed host allocation (ASan's operator new aborts instead of throwing bad_alloc) and the
# two that build their own host binaries with gcc -fsanitize (gcc's libasan cannot share a process tree with the preloaded clang runtime).
set -e
cd "$(dirname "$0")/.."
export NLX_BUILD_VARIANT=asan
NLX_EXTRA_FLAGS="-Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer" \
NLX_EXTRA_LDFLAGS="-fsanitize=address,undefined -shared-libsan" python near-light-client_amd/build.py > /tmp/asan_build.log 2>&1
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
rm -f /tmp/nlx_asan_report*
set +e
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:log_path=/tmp/nlx_asan_report \
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:log_path=/tmp/nlx_asan_report \
timeout 3000 python -m pytest tests -q -s -m "not gpu" -k "not exceptions_do_not_cross and not failed_allocations and not row_code_built_for_the_host and not device_witness_code_on_the_host" > /tmp/asan_suite.log 2>&1
rc=$?
{
  echo "libnlx_asan.so / libnlx_synth_asan.so: host side built with -fsanitize=address,undefined (hipcc -Xarch_host), CPU suite through it"
  echo "(LD_PRELOAD=libclang_rt.asan-x86_64.so, detect_leaks=0, halt_on_error=1; excluded: the two forced-allocation-failure tests and the two tests that run gcc-sanitized binaries of their own)"
  echo "pytest exit code: $rc"
  tail -3 /tmp/asan_suite.log
  echo "sanitizer reports: $(ls /tmp/nlx_asan_report* 2>/dev/null | wc -l)"
  for f in /tmp/nlx_asan_report*; do [ -f "$f" ] && head -20 "$f"; done
} > profiles/r04_asan_host.txt
cat profiles/r04_asan_host.txt
