#!/bin/bash
# HBM traffic of the bench's kernels: separate rocprofv3 --pmc passes (FETCH_SIZE needs 3 of the 4 TCC slots,
# WRITE_SIZE 2: never together), one forward each. Run on the GPU box; writes gpurun_out/pmc_bench/*.csv.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/pmc_bench
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_bench -o $c -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_bench/$c.log 2>&1
  echo "$c rc=$?"
done
