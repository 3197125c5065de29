#!/bin/bash
# Ablation builds of the 32x32x16 scorer (sel_scores_mfma32.hip: SC32_* switches) next to the product library, timed on one box.
# In the container:  tools/ablate_scorer.sh build      (writes ab/<name>.so; ab/ is git-ignored and travels with gpurun)
# On the GPU box:    tools/ablate_scorer.sh run <out-prefix>   (bench_stages.py 65536x16 under every build)
set -eu
cd "$(dirname "$0")/.."
CS=nsa_vibe_amd/csrc
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize"
declare -A V=([full]="" [k1]="-DSC32_KSTEPS=1" [noexp]="-DSC32_NOEXP" [noexp_k1]="-DSC32_NOEXP -DSC32_KSTEPS=1" [nosync]="-DSC32_NOSYNC" [nosync_k1]="-DSC32_NOSYNC -DSC32_KSTEPS=1"
              [s1]="-DSC32_NOSYNC -DSC32_SWEEP1" [s1_k1]="-DSC32_NOSYNC -DSC32_SWEEP1 -DSC32_KSTEPS=1" [s1_noexp]="-DSC32_NOSYNC -DSC32_SWEEP1 -DSC32_NOEXP"
              [s1_noexp_k1]="-DSC32_NOSYNC -DSC32_SWEEP1 -DSC32_NOEXP -DSC32_KSTEPS=1")
ORDER="full k1 noexp noexp_k1 nosync nosync_k1 s1 s1_k1 s1_noexp s1_noexp_k1"
if [ "$1" = build ]; then
    make -C $CS -j8 > /dev/null
    mkdir -p ab
    for n in $ORDER; do
        /opt/rocm/bin/hipcc $F ${V[$n]} -c $CS/sel_scores_mfma32.hip -o /tmp/sc32_$n.o
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/$n.so $(ls $CS/build/*.o | grep -v sel_scores_mfma32) /tmp/sc32_$n.o
    done
    cp $CS/../libnsa_sel_hip.so ab/new.so
else
    AB_SET="$ORDER" tools/ab_libs.sh "$2" -- python tools/bench_stages.py 65536x16
    for n in $ORDER; do printf "%-12s %s\n" $n "$(grep 'S=' $2.$n.log | sed 's/  select.*//')"; done | tee $2.txt
fi
