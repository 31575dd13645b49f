#!/bin/bash
# Where the attention kernel's wave-cycles go: two rocprofv3 --pmc passes (SQ counters only) over tools/attn_one.py.
# tools/pmc_attn.sh <variants, e.g. 1,9> [heads] -> gpurun_out/pmc_attn/; summarise with tools/pmc_attn_summarize.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc_attn
mkdir -p $out
export ATTN_VARIANTS=${1:-1,9} ATTN_REPS=1 ATTN_SCALES=1.0
timeout -k 5 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o a -- python3 tools/attn_one.py ${2:-16} > $out/a.log 2>&1
echo "pass a rc=$?"
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out -o b -- python3 tools/attn_one.py ${2:-16} > $out/b.log 2>&1
echo "pass b rc=$?"
