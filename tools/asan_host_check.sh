#!/bin/bash
# CPU sanitizer run of the host code (SURVEY.md §5 "race detection / sanitizers").  Build container only — GPU
# sanitizers are not available on this pool and nothing here touches a GPU.
#   make -C csrc asan    the host-only translation units (CPython-exact MT19937, cosine schedule + error channel, the replay ring's
#                        deque bookkeeping) with -fsanitize=address,undefined -> libgcrl_host_asan.so
#   then tests/test_host_abi.py against that library in a Python process with the sanitizer runtime preloaded.
# Writes the log to profiles/ when a file name is given:  tools/asan_host_check.sh profiles/r04_asan_host_check.txt
set -e
cd "$(dirname "$0")/.."
make -C goal-conditioned-rl-framework_amd/csrc asan
lib=$PWD/goal-conditioned-rl-framework_amd/libgcrl_host_asan.so
rt=$(g++ -print-file-name=libasan.so)
out=${1:-/dev/stdout}
{
  echo "# $(date -u +%FT%TZ)  g++ $(g++ -dumpversion), -fsanitize=address,undefined -fno-sanitize-recover=undefined"
  echo "# LD_PRELOAD=$rt  GCRL_HOST_ASAN_LIB=$lib  python -m pytest tests/test_host_abi.py -q"
  LD_PRELOAD=$rt ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 GCRL_HOST_ASAN_LIB=$lib \
    python -m pytest tests/test_host_abi.py -q -p no:cacheprovider 2>&1 | tail -15
} > "$out"
[ "$out" = /dev/stdout ] || cat "$out"
