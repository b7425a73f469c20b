#!/bin/bash
# rebuild xattn.o + the E-stream cross-attention microbenchmark; usage: tools/mb_xattn.sh [skip mask]  (XA_SKIP of xattn.hip, default 0)
set -e
cd "$(dirname "$0")/.."
v=${1:-0}
mkdir -p whisper_amd/csrc/build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DXA_SKIP=$v -c whisper_amd/csrc/xattn.hip -o /tmp/xattn_v$v.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/mb_xattn.hip -o /tmp/mb_xattn.o
hipcc --offload-arch=gfx950 /tmp/mb_xattn.o /tmp/xattn_v$v.o -o tools/mb_xattn$([ "$v" != 0 ] && echo _v$v)
