"""200 single drop-in calls each of bign128Sign2, bign128Verify, bignPubkeyCalc -- to be run under rocprofv3 --kernel-trace --stats
(cd /tmp && export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/single -o s -- python $R/tools/ab/single_call_prof.py)
so that the kernel durations behind bench.py's single_call_latency_us can be read per kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bee2_amd, goldenlib
from bee2_amd import engine as E
eng = bee2_amd.load(); eng.set_device(0)
G = goldenlib.Golden()
l = 128
P = eng.bignParamsStd(E.CURVE_NAME[l])
oid = E.LEVEL_OID[l]
h, s, p = G.bign_base[0]
priv = bytes(range(1, 33))
for _ in range(200):
    assert eng.bignSign2(P, oid, h, priv, None)[0] == 0
for _ in range(200):
    assert eng.lib.bign128Verify(h, s, p) == 0
for _ in range(200):
    assert eng.bignPubkeyCalc(P, priv)[0] == 0
