#!/bin/bash
# Builds the diagnostics variant of the library that counts the matrix-core matcher's accumulator blocks (-DTOD_K4X_COUNT_WALKS, an
# atomic per block: timings of this build mean nothing) into tod_amd/libtodhip_walks.so -- run HERE (hipcc cross-compiles); on the GPU
# box: python tools/k4x_on_chained_db.py (product build: timings), then TODHIP_LIB_PATH=$PWD/tod_amd/libtodhip_walks.so python
# tools/k4x_on_chained_db.py (walk fractions).
set -e
cd "$(dirname "$0")/../tod_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -DTOD_K4X_COUNT_WALKS -c match.hip -o /tmp/match_walks.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtodhip_walks.so capi.o /tmp/match_walks.o verify.o orb.o train.o l2.o pnp.o lsh.o
ls -la ../libtodhip_walks.so
