#!/bin/bash
# hardware queues per process (GPU_MAX_HW_QUEUES, the HIP runtime's default is 4) x lanes: does a lane stream of its own queue help?
# every GPU step under its own time limit; the script stops at the first failure
set -e
cd "$(dirname "$0")/.."
out=gpurun_out/sweep_hw_queues.txt
: > $out
for cfg in ${SWEEP:-"8 4" "8 6" "8 8" "16 8"}; do
    set -- ${cfg//:/ }
    echo "== GPU_MAX_HW_QUEUES=$1 lanes=$2 steps=${3:-8} warmup=${4:-4}" | tee -a $out
    GPU_MAX_HW_QUEUES=$1 timeout -k 10 300 python bench.py --lanes $2 --steps ${3:-8} --warmup ${4:-4} --no-single-batch --no-cpu-baseline --no-profile-pass > gpurun_out/_q.json 2> gpurun_out/_q.err
    python - <<'PY' | tee -a $out
import json
d = json.loads(open("gpurun_out/_q.json").read().strip().splitlines()[-1])
print("   %.1f RTFx  %.1f ms per step  lanes %s" % (d["value"], d["ms_per_step"], d["config"].get("lanes")))
PY
done
