#!/bin/bash
# rebuild the library objects + the encoder GEMM microbenchmark
set -e
cd "$(dirname "$0")/.."
python -m whisper_amd.build > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/mb_enc.hip -o /tmp/mb_enc.o
hipcc --offload-arch=gfx950 /tmp/mb_enc.o whisper_amd/csrc/build/gemm.o -o tools/mb_enc
