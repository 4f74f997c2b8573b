import sys, os
sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/tools']
import numpy as np, torch
import bee2_amd, orclib
from bee2_amd import engine as E
eng=bee2_amd.load(); eng.set_device(0); orc=orclib.load()
fail=0
for l in (128,192,256):
    no=l//4
    params=eng.bignParamsStd(E.CURVE_NAME[l]); oid=E.LEVEL_OID[l]
    n=70
    privs=bytearray(orc.fill(no*n, 77+l)); hashes=orc.fill(no*n, 99+l)
    # a few special keys: 0, 1, q, q-1, q-2
    q=int.from_bytes(bytes(params.q)[:no],'little')
    for i,v in enumerate((0,1,q,q-1,q-2,q+5)): privs[no*i:no*(i+1)]=(v%(1<<(8*no))).to_bytes(no,'little')
    privs=bytes(privs)
    code,pubs,codes=eng.bignPubkeyCalc_batch(params,privs)
    assert code==0
    for i in range(n):
        c,p=orc.pubkey_calc(l,privs[no*i:no*(i+1)])
        if c!=codes[i] or (c==0 and p!=pubs[2*no*i:2*no*(i+1)]): fail+=1; print("pubkeycalc",l,i,c,codes[i])
    for t in (None,b"abc",bytes(range(64)),bytes(range(100))):
        code,sigs,codes=eng.bignSign2_batch(params,oid,hashes,privs,t)
        assert code==0,code
        for i in range(n):
            c,s=orc.sign2(l,oid,hashes[no*i:no*(i+1)],privs[no*i:no*(i+1)],t)
            sg=3*l//8
            if c!=codes[i] or (c==0 and s!=sigs[sg*i:sg*(i+1)]): fail+=1; print("sign2",l,i,c,codes[i], t and len(t))
    print("level",l,"done, failures so far",fail, flush=True)
print("FAIL" if fail else "ALL OK")
