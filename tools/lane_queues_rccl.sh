#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mode in 0 1; do
  export NO_RCCL=$mode
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lqr$mode -o t -- python3 tools/lane_queues_rccl.py > gpurun_out/lqr$mode.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/lqr$mode/**/*kernel_trace.csv", recursive=True)[0]
per = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    per[r["Thread_Id"]][r["Queue_Id"]] += 1
print("NO_RCCL=$mode: thread -> {queue: launches}")
for t, c in sorted(per.items(), key=lambda kv: -sum(kv[1].values())):
    print("  thread", t, dict(c))
PY
  rm -rf gpurun_out/lqr$mode
done
