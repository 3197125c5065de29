#!/bin/bash
# One profiling batch of the round (run on the GPU box from the repo root): kernel-trace stats of bench.py, PMC traffic of the headline
# attention launches, of scores + select, of the selection backward and of the COLD decode shapes.  Everything lands under gpurun_out/
# (merged back by gpurun); copy what is to be judged into profiles/<round>/.
set -x
R=${NSA_PROFILE_ROUND:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
export NSA_PROFILE_ROUND=$R
for bs in "64 4096" "256 4096" "64 16384" "128 16384" "256 16384" "64 65536" "128 65536" "256 65536" "512 65536"; do
  set -- $bs
  bash tools/pmc_traffic.sh decode_cold_B$1_S$2 decode_step decode_cold $1 $2 12 > /dev/null
done
bash tools/pmc_traffic.sh S65536_B16 sel_attn_blocks_mfma_kernel,sel_attn_ksplit_combine_kernel prefill 65536 16 3 attn > /dev/null
bash tools/pmc_traffic.sh scores_select_S65536_B16 scores_mfma,select_topn_kernel prefill 65536 16 3 all > /dev/null
bash tools/pmc_traffic.sh scores_select_one_launch_S65536_B16 scores_mfma,select_topn_kernel prefill 65536 16 3 scsel > /dev/null
bash tools/pmc_traffic.sh sel_bwd_S4096_B8 bwd_ bwd 4096 8 4 > /dev/null
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-extra > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err)
f=$(find $OUT/bench_stats -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats_bench_S65536_B16.csv
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/decode_stats -- python3 $GRAFT_REPO_ROOT/tools/prof_hot.py decode_cold 256 65536 30 > $OUT/decode_stats.log 2>&1)
f=$(find $OUT/decode_stats -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats_decode_cold_B256_S65536.csv
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
ls $OUT
