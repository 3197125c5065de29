#!/bin/bash
# The GPU suite under the non-default kernel forms (run on the GPU box): every A/B switch of nsa_hip_set_tuning keeps an older or alternative
# form of a kernel alive, and only the defaults run in a plain `pytest -m gpu`.  Each set must pass like the defaults do.
#   tools/run_alt_tunings.sh            (about a minute per set)
set -e
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -1; }
run NSA_HIP_SEL_FLAT=1 NSA_HIP_SEL_FUSE=1
run NSA_HIP_SEL_FLAT=0 NSA_HIP_SCORES_FORM=0 NSA_HIP_DECODE_STENCIL=0 NSA_HIP_SEL_ROWSUM=0
run NSA_HIP_SEL_ROWS=1 NSA_HIP_DECODE_UNFUSED=1 NSA_HIP_DECODE_WG=0 NSA_HIP_SCORES_FORM=1
run NSA_HIP_SEL_ROWS=0 NSA_HIP_ATTN_STAGE=0 NSA_HIP_BAND_STAGE=0 NSA_HIP_ATTN_MAP=0
run NSA_HIP_SEL_BLOCKS=2 NSA_HIP_DECODE_UNFUSED=0 NSA_HIP_ATTN_MAP=1
run NSA_HIP_SEL_KSPLIT=1 NSA_HIP_SEL_FLAT=0
# round 4: the zoned key split on every attention shape of the suite (rows whole / in two / in four key classes by position), the selection
# inside the scorer launch at every length, and the exact one-workgroup form of the decode step wherever a row fits it
run NSA_HIP_SEL_KSPLIT=1 NSA_HIP_SEL_FLAT=0 NSA_HIP_SEL_KSPLIT_T1=200 NSA_HIP_SEL_KSPLIT_T2=900
run NSA_HIP_SEL_KSPLIT=1 NSA_HIP_SEL_FLAT=0 NSA_HIP_SEL_KSPLIT_T1=0 NSA_HIP_SEL_KSPLIT_T2=0
run NSA_HIP_SCORES_SELECT=1 NSA_HIP_DECODE_WIDE=1
# the layer decode step with the band branches as their own launch / on the step's launch with the finish kernel / with the mix in the
# projection at every batch, the last two with the step forced into a team of workgroups and into the one-pass form
run NSA_HIP_DECODE_BAND=0
run NSA_HIP_DECODE_BAND=1 NSA_HIP_DECODE_SPLIT=2
run NSA_HIP_DECODE_BAND=3 NSA_HIP_DECODE_WIDE=2
