#!/bin/bash
# rebuild the library objects + the microbenchmark
set -e
cd "$(dirname "$0")/.."
python -m whisper_amd.build > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/microbench.hip -o /tmp/microbench.o
hipcc --offload-arch=gfx950 /tmp/microbench.o whisper_amd/csrc/build/gemm.o whisper_amd/csrc/build/elementwise.o whisper_amd/csrc/build/attention.o -o tools/microbench
