#!/bin/bash
# Where one kernel's cycles go: three rocprofv3 --pmc passes over a small script -> gpurun_out/pmc_kernel/; tools/pmc_tail_summarize.py
# usage: tools/pmc_kernel.sh tools/conv_up_one.py [args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc_kernel
rm -rf $out; mkdir -p $out
timeout -k 5 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o a -- python3 "$@" > $out/a.log 2>&1
echo "pass a rc=$?"
timeout -k 5 300 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o b -- python3 "$@" > $out/b.log 2>&1
echo "pass b rc=$?"
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $out -o c -- python3 "$@" > $out/c.log 2>&1
echo "pass c rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o $c -- python3 "$@" > $out/$c.log 2>&1
  echo "$c rc=$?"
done
