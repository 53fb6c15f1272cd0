#!/bin/bash
# Host logic under ASan + UBSan on the CPU (GPU sanitizers are not available on this pool).
# Usage: bash tests/manual/sanitize_host.sh
set -e
cd "$(dirname "$0")/../.."
OUT=$(mktemp -d)
g++ -O1 -g -std=c++17 -fPIC -shared -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude \
    lvb_amd/csrc/host_api.cpp lvb_amd/csrc/proposals.cpp lvb_amd/csrc/anneal.cpp lvb_amd/csrc/anneal_chains.cpp lvb_amd/csrc/refsearch.cpp \
    lvb_amd/csrc/program.cpp -x c tests/cpu_double/lvbgpu_double.c oracle/fitch_oracle.c -Wl,-Bsymbolic \
    -o "$OUT/liblvbhost_double_asan.so"
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \
    python tests/manual/sanitize_host.py "$OUT/liblvbhost_double_asan.so"
rm -rf "$OUT"
