#!/bin/bash
# One profiling batch of the round (run on the GPU box from the repo root): kernel-trace stats of bench.py, PMC traffic of the headline
# attention launches and of the decode shapes, PMC instruction mix of the one-launch decode step.  Everything lands under gpurun_out/
# (merged back by gpurun); copy what is to be judged into profiles/<round>/.
set -x
R=${NSA_PROFILE_ROUND:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
export NSA_PROFILE_ROUND=$R
bash tools/pmc_traffic.sh decode_B64_S16384 decode_step_kernel decode 64 16384 10
bash tools/pmc_traffic.sh decode_B256_S16384 decode_step_kernel decode 256 16384 10
bash tools/pmc_traffic.sh decode_B64_S65536 decode_step_kernel decode 64 65536 10
bash tools/pmc_traffic.sh decode_B1_S65536 decode_step_kernel decode 1 65536 10
bash tools/pmc_traffic.sh S65536_B16 sel_attn_blocks_mfma_kernel,sel_attn_ksplit_combine_kernel prefill 65536 16 3 attn
bash tools/pmc_traffic.sh scores_select_S65536_B16 scores_mfma,select_topn_kernel prefill 65536 16 3 all
bash tools/pmc_run.sh $OUT/pmc_decode_B64_S16384 tools/prof_hot.py decode 64 16384 10
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-extra > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err)
f=$(find $OUT/bench_stats -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats_bench_S65536_B16.csv
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/decode_stats -- python3 $GRAFT_REPO_ROOT/tools/prof_hot.py decode 64 16384 50 > $OUT/decode_stats.log 2>&1)
f=$(find $OUT/decode_stats -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats_decode_B64_S16384.csv
python3 tools/pmc_summary.py $OUT/pmc_decode_B64_S16384 decode_step_kernel > $OUT/pmc_decode_B64_S16384.txt 2>&1
ls $OUT
