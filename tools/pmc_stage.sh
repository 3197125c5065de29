#!/bin/bash
# PMC passes of one hot-path stage (run on the GPU box):  tools/pmc_stage.sh <tag> <stage: attn|scores|select> <kernel name filter> <S> <B>
set -e
TAG=$1; STAGE=$2; FILT=$3; S=$4; B=$5
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" \
           "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/prof_hot.py prefill $S $B 3 $STAGE > $OUT/log$i 2>&1 || echo "pass $i failed: $set"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT $FILT > $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.txt
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.txt
