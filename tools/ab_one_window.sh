#!/bin/bash
# A/B of a decode-chain change at few rows: bench.py --batch 1 / 4 --lanes 1 and the 5-minute sequential file, with and without ENVVAR=1
# usage: tools/ab_one_window.sh ENVVAR
set -e
cd "$(dirname "$0")/.."
v=$1
for b in ${ROWS:-1 4}; do
  for off in 0 1; do
    if [ $off = 1 ]; then export $v=1; else unset $v; fi
    timeout -k 10 200 python bench.py --batch $b --lanes 1 --steps 6 --warmup 2 --no-cpu-baseline --no-single-batch --no-profile-pass > gpurun_out/_ab.json 2>/dev/null
    python - <<PY
import json
d = json.loads(open("gpurun_out/_ab.json").read().strip().splitlines()[-1])
print("batch $b  $v=%s: %.1f RTFx, %.1f ms per step, decode %.3f ms per position" % ("$off", d["value"], d["ms_per_step"], d["stages"]["decode"]["ms_per_token_step"]))
PY
  done
done
unset $v
timeout -k 10 200 python tools/bench_sequential.py large-v3 5 1 | grep "prompt pass on" | tail -1
export $v=1
timeout -k 10 200 python tools/bench_sequential.py large-v3 5 1 | grep "prompt pass on" | tail -1 | sed "s/^/$v=1: /"
