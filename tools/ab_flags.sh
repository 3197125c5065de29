#!/bin/bash
# same-box A/B of compiler flags: tools/ab_flags.sh "<extra flags>"   (rebuilds the library twice on the GPU box)
BASE="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -ffp-contract=off -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1"
for v in base extra base extra; do
  if [ $v = base ]; then F="$BASE"; else F="$BASE $1"; fi
  make -C nsa_vibe_amd/csrc -B -j16 FLAGS="$F" > /dev/null 2>&1
  echo "VARIANT $v"
  timeout -k 10 120 python tools/bench_band.py 1 65536
  timeout -k 10 120 python tools/bench_band.py 8 4096
  timeout -k 10 200 python bench.py --no-extra --steps 20 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hot', d['ms_per_step'], d['stages_ms'])"
  timeout -k 10 120 python tools/bench_bwd.py 8 4096 | head -1
done
