#!/bin/bash
# MFMA / LDS utilisation of the bench's kernels: one rocprofv3 --pmc pass (SQ counters only), one forward.
# tools/pmc_mfma.sh <tag> [bench args] -> gpurun_out/pmc_mfma_<tag>/; summarise with tools/pmc_mfma_summarize.py.
tag=${1:-vitl}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc_mfma_$tag
mkdir -p $out
timeout -k 5 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o m -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-inflight2 "$@" > $out/m.log 2>&1
echo "rc=$?"
