#!/bin/bash
# PMC passes of the select+attend launch for one kernel form (run on the GPU box):
#   tools/pmc_attn.sh <tag> <SEL_ROWS> <SEL_BLOCKS> <S> <B>
set -e
TAG=$1; export NSA_HIP_SEL_ROWS=$2; export NSA_HIP_SEL_BLOCKS=$3; S=$4; B=$5
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" \
           "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr" \
           "TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/prof_hot.py prefill $S $B 3 attn > $OUT/log$i 2>&1 || echo "pass $i failed: $set"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT sel_attn > $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.txt
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.txt
