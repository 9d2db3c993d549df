#!/usr/bin/env python3
"""Differential fuzz of the one-pass row parser against the general parser (csrc/dsp_text.cpp): golden rows with random byte replacements / deletions / insertions; arrays, sampleinfo addressing and error texts must agree.  No GPU.
usage: fuzz_parser.py SEED SECONDS   (round 3: 4 seeds x 40 s + 2 x 300 s = 4.9 M cases, then 3 more seeds x 300 s at the round's last HEAD = 6.2 M cases: no mismatch)"""
import sys, os, numpy as np, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsignal_plant_amd import textio, _native as nat
L=nat.lib()
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
rows=open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'f2_rows.tsv')).read().splitlines()
pool=b"\t,;.-+eE0123456789 \nACGTNX\r:_"
def both(data):
    out=[]
    for fast in (1,0):
        L.dsp_text_set_fast_rows_(fast)
        try:
            r=textio.parse_rows(data,13,16,nthreads=1)
            out.append(("ok",r.n,r.kmer.tobytes(),r.means.tobytes(),r.stds.tobytes(),r.lens.tobytes(),r.signals.tobytes(),r.labels.tobytes(),r.row_off.tobytes(),r.info_len.tobytes(),r.read_off.tobytes(),r.read_len.tobytes()))
        except (ValueError, KeyError, IndexError) as e:
            out.append(("err",type(e).__name__+str(e)))
    return out
t0=time.time(); n=0; nerr=0
while time.time()-t0 < float(sys.argv[2]) if len(sys.argv)>2 else 60:
    k=int(rng.integers(1,4))
    base=("\n".join(rows[int(i)] for i in rng.integers(0,len(rows),k))+"\n").encode()
    bad=bytearray(base)
    m=int(rng.integers(0,4))
    for _ in range(m):
        op=int(rng.integers(0,3)); pos=int(rng.integers(0,len(bad)))
        if op==0: bad[pos]=pool[int(rng.integers(0,len(pool)))]
        elif op==1: del bad[pos]
        else: bad.insert(pos,pool[int(rng.integers(0,len(pool)))])
    a,b=both(bytes(bad))
    if a!=b:
        print("MISMATCH", bytes(bad)[:300]); print(a[:2], b[:2]); sys.exit(1)
    n+=1; nerr+= a[0]=="err"
L.dsp_text_set_fast_rows_(1)
print("cases",n,"errors",nerr,"no mismatch")
