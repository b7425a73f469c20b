#!/bin/bash
# RTFx / ms per step / stage times of the other BASELINE configurations (profiles/rNN_other_configs.txt); usage: tools/other_configs.sh <out file>
out=${1:-gpurun_out/other_configs.txt}
: > "$out"
run() {
    timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-profile-pass --no-single-batch > gpurun_out/oc.log 2>&1 || { echo "$*: FAILED" >> "$out"; tail -3 gpurun_out/oc.log >> "$out"; return 1; }
    echo "done: $*"
    python - "$*" >> "$out" <<'PY'
import json, sys
l = json.loads([x for x in open("gpurun_out/oc.log") if x.startswith('{"metric"')][-1])
print(f"{sys.argv[1]}: {l['value']:.1f} RTFx, {l['ms_per_step']:.1f} ms per step,", {k: round(v['ms'], 1) for k, v in l['stages'].items() if isinstance(v, dict)})
PY
}
run --model turbo --batch 64 && run --model turbo --batch 80 && run --batch 80 && run --batch 1 && run --batch 4 && run --batch 16 && run --batch 32 && \
run --batch 15 --beam 5 && run --model base --batch 16 && run --model small --batch 16 && run --model medium --batch 64
cat "$out"
