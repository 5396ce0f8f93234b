# The CPU builds (oracle/*.c and the host build of the engine's headers, oracle/engine_host.cpp) under
# AddressSanitizer + UndefinedBehaviorSanitizer, then the whole `-m "not gpu"` suite on them:
#   bash tools/sanitize_cpu.sh          (GPU sanitizers are not available on this pool)
# The normal build is put back afterwards.
set -e
cd "$(dirname "$0")/../oracle"
O=$(pwd)
B=$(mktemp -d)
cp -r _build "$B/keep" 2>/dev/null || true
# absolute paths: the trap runs from wherever the script is when it exits (round 5: it ran from the repository root, left the
# sanitizer build in oracle/_build and the kept one in ./_build)
trap 'rm -rf "$O/_build"; [ -d "$B/keep" ] && cp -r "$B/keep" "$O/_build" && touch "$O"/_build/*; rm -rf "$B"' EXIT
make clean > /dev/null
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -O1 -g -march=x86-64-v3 -ffp-contract=off -fno-fast-math -fPIC -Wno-unused-function"
make CFLAGS="$SAN -std=gnu11" CXXFLAGS="$SAN -std=c++17 -Wno-unknown-pragmas" > /dev/null
cd ..
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider
