#!/bin/bash
# MFMA / LDS utilisation of the bench's kernels: one rocprofv3 --pmc pass (SQ counters only), one forward.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/pmc_mfma
timeout -k 5 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -o m -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_mfma/m.log 2>&1
echo "rc=$?"
