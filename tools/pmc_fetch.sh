#!/bin/bash
# FETCH_SIZE only (one rocprofv3 --pmc pass over one forward): tools/pmc_fetch.sh <tag> [bench args] -> gpurun_out/pmc_fetch_<tag>/
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc_fetch_$tag
mkdir -p $out
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out -o FETCH_SIZE -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-inflight2 "$@" > $out/FETCH_SIZE.log 2>&1
echo "rc=$?"
