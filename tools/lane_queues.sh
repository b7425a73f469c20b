#!/bin/bash
# which hardware queue each lane's kernels run on (rocprofv3 kernel trace, Queue_Id per launching thread); usage: tools/lane_queues.sh <GPU_MAX_HW_QUEUES>
set -e
q=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=$q
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lq$q -o t -- python3 bench.py --steps 4 --warmup 0 --max-new-tokens 8 --no-cpu-baseline --no-single-batch --no-profile-pass > gpurun_out/lq$q.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/lq$q/**/*kernel_trace.csv", recursive=True)[0]
per = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    per[r["Thread_Id"]][r["Queue_Id"]] += 1
print("GPU_MAX_HW_QUEUES=$q: thread -> {queue: launches}")
for t, c in sorted(per.items(), key=lambda kv: -sum(kv[1].values())):
    print("  thread", t, dict(c))
PY
rm -rf gpurun_out/lq$q
