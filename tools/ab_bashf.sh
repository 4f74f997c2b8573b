#!/bin/bash
run() { python bench.py --steps 200 --warmup 20 --no-cpu --only bashF 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['value']/1e9,3), 'Gperm/s wall;', round(d['roofline']['frac']*8000/384,3), 'event-timed')"; }
for i in 1 2; do
  run base
  for v in 2n 3n 2p 3p 4p; do BEE2HIP_BASHF_PERSIST=$v run persist_$v; done
done
python -m pytest tests -m gpu -q -x -k "bashF" 2>&1 | tail -1
BEE2HIP_BASHF_PERSIST=3p python -m pytest tests -m gpu -q -x -k "bashF" 2>&1 | tail -1
