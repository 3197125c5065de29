#!/bin/bash
# A/B of two builds of the native library on the same box: ab/old.so and ab/new.so (both built in the container, git-ignored; e.g.
# `make -C nsa_vibe_amd/csrc BUILD=build_new OUT=../../ab/new.so`) are loaded one after the other THROUGH NSA_HIP_LIB (nsa_vibe_amd/_lib.py) --
# the product library nsa_vibe_amd/libnsa_sel_hip.so is never overwritten -- and the same timing scripts run against each.
# Usage: tools/ab_libs.sh <out-prefix> -- cmd1 ';;' cmd2 ...
set -u
out=$1; shift; shift
mkdir -p "$(dirname "$out")"
IFS=$'\n' read -r -d '' -a cmds < <(printf '%s ' "$@" | sed 's/ ;; /\n/g' && printf '\0')
for which in ${AB_SET:-old new}; do
    export NSA_HIP_LIB="$PWD/ab/$which.so"
    for c in "${cmds[@]}"; do
        echo "== [$which] $c" >> "$out.$which.log"
        timeout -k 10 400 bash -c "$c" >> "$out.$which.log" 2>&1 || { echo "FAILED: $c" >> "$out.$which.log"; exit 1; }
    done
done
