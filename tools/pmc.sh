#!/bin/bash
# one rocprofv3 PMC pass (own run, no tracing domains) over a very short bench, one batch at a time (--lanes 1: per-kernel counters); usage: tools/pmc.sh <name> <counters...>
set -e
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$name -o pmc -- python3 bench.py --lanes ${PMC_LANES:-1} --steps ${PMC_LANES:-1} --warmup 0 --max-new-tokens 4 --no-cpu-baseline --no-profile-pass > gpurun_out/$name.log 2>&1
ls gpurun_out/$name
