#!/bin/bash
# decode GEMM K-splits with 4 batches in flight (fewer workgroups per launch = less CU-time); usage: tools/sweep_split_lanes.sh <out file>
out=${1:-gpurun_out/sweep_split_lanes.txt}
: > "$out"
run() { echo -n "$*: " | tee -a "$out"; env "$@" timeout -k 10 250 python bench.py --no-cpu-baseline --no-profile-pass --no-single-batch 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'RTFx', round(d['ms_per_step'],1), 'ms')" | tee -a "$out"; }
run X=1
run WX_SPLIT_DD=2
run WX_SPLIT_DD=1
run WX_SPLIT_QKV=2
run WX_SPLIT_FC2=5
run WX_SPLIT_DD=2 WX_SPLIT_QKV=2 WX_SPLIT_FC2=5
run X=2
