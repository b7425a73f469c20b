#!/bin/bash
# rocprofv3 kernel-trace + stats of a short bench run; usage: tools/prof.sh <name> [bench args...]
# writes gpurun_out/<name>/ and prints the per-kernel summary (copy it to profiles/ to keep it)
set -e
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$name -o vm -- python3 bench.py "$@" > gpurun_out/$name.log 2>&1
grep '^{"metric"' gpurun_out/$name.log > gpurun_out/$name.json || true
python3 tools/prof_summary.py gpurun_out/$name 30 | tee gpurun_out/$name.summary.txt
# the raw kernel trace of a full-length run is ~100 MB: keep it only when it is small enough to travel back
find gpurun_out/$name -name '*kernel_trace.csv' -size +8M -delete
