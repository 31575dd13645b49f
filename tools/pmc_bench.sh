#!/bin/bash
# HBM traffic of the bench's kernels: separate rocprofv3 --pmc passes (FETCH_SIZE needs 3 of the 4 TCC slots,
# WRITE_SIZE 2: never together), one forward each. Run on the GPU box: tools/pmc_bench.sh <tag> [bench args]
# writes gpurun_out/pmc_bench_<tag>/*.csv; summarise with tools/pmc_summarize.py.
tag=${1:-vitl}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc_bench_$tag
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o $c -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-inflight2 "$@" > $out/$c.log 2>&1
  rc=$?
  echo "$c rc=$rc"
  [ $rc -eq 0 ] || exit $rc
done
