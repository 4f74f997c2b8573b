#!/bin/bash
# A/B of the ragged hashing leg of bench.py between tools/ubench/base/libbee2hip.so (the build before a change) and the in-tree library,
# alternating inside ONE gpurun call
for i in 1 2; do
for L in base new; do
  if [ $L = base ]; then export BEE2HIP_LIB=$PWD/tools/ubench/base/libbee2hip.so; else unset BEE2HIP_LIB; fi
  python bench.py --no-cpu --only ragged --steps 5 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
def find(o):
    if isinstance(o,dict):
        if 'bash256_uniform_1000B' in o: return o
        for v in o.values():
            r=find(v)
            if r: return r
print('$L', {k:round(v,1) for k,v in find(d).items() if isinstance(v,float)})"
done; done
