#!/bin/bash
# rebuild the library objects + the cross-attention microbenchmark
set -e
cd "$(dirname "$0")/.."
python -m whisper_amd.build > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/mb_cross.hip -o /tmp/mb_cross.o
hipcc --offload-arch=gfx950 /tmp/mb_cross.o whisper_amd/csrc/build/attention.o -o tools/mb_cross
