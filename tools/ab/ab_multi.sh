# usage: ab_multi.sh <workload> <lib1> <lib2> ...   alternating bench runs (value + kernel ms)
W=$1; shift
for i in 1 2 3; do for L in "$@"; do
  BEE2HIP_LIB=$L python bench.py --no-cpu --only $W 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
r=d.get('roofline') or {}
print('$L'.split('/')[-2], '%.4g %s' % (d['value'], d['unit']), ('kernel %.4f ms' % r['avg_launch_ms']) if 'avg_launch_ms' in r else '')"
done; done
