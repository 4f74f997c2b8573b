#!/bin/bash
# A/B a bashF kernel variant against the in-tree library on the GPU box.
#   build here:   make -C bee2_amd/csrc EXTRA=-DBASHF_STAGED OUT=$PWD/tools/ubench/variant
#   run on GPU:   bash tools/ab_bashf.sh tools/ubench/variant/libbee2hip.so
# Prints wall-clock and event-timed G perm/s for both, alternating, then runs the bashF parity
# tests against the variant (a variant that is fast and wrong is worth nothing).
VAR=${1:?path to the variant libbee2hip.so}
run() { python bench.py --steps 200 --warmup 20 --no-cpu --only bashF 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['value']/1e9,3), 'Gperm/s wall;', round(d['roofline']['achieved']/384*1e0,3), 'event-timed')"; }
for i in 1 2 3; do
  run base
  BEE2HIP_LIB=$VAR run variant
done
echo "parity, variant:"; BEE2HIP_LIB=$VAR python -m pytest tests -m gpu -q -x -k "bash" 2>&1 | tail -1
