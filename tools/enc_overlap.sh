#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/eo -o t -- python3 bench.py --steps 8 --warmup 4 --max-new-tokens 64 --no-cpu-baseline --no-single-batch --no-profile-pass > gpurun_out/eo.log 2>&1
python3 tools/enc_overlap_trace.py gpurun_out/eo | tee gpurun_out/enc_overlap.txt
rm -rf gpurun_out/eo
