import sys; sys.path.insert(0,"tests"); sys.path.insert(0,".")
import numpy as np, oracle_lib as ol
from minimap2_chaindp_amd import anchorgen as ag, chaindp, params as P
par=P.preset("ava-ont")
off,a=ag.generate("ava-ont", n_reads=2, seed=11)
with chaindp.Device(0, max_anchors=int(off[-1])+1, max_reads=4) as d:
    f,p,v=d.chain_batch(par,off,a)
of,op,ov,_=ol.oracle_batch(par,off,a,threads=2)
x=a[:,0]
# unit starts: read start or gap > max_dist_x
start=np.zeros(len(x),bool)
for r in range(len(off)-1):
    lo,hi=int(off[r]),int(off[r+1]); start[lo]=True
    start[lo+1:hi]=(x[lo+1:hi]-x[lo:hi-1])>np.uint64(par.max_dist_x)
us=np.maximum.accumulate(np.where(start,np.arange(len(x)),0))
rel=np.arange(len(x))-us
bad=np.flatnonzero((f!=of)|(p!=op))
print("n bad",bad.size)
seen=set()
for b in bad:
    u0=int(us[b])
    if u0 in seen: continue
    seen.add(u0)
    ulen=int(np.flatnonzero(start[u0+1:])[0])+1 if start[u0+1:].any() else len(x)-u0
    r=int(np.searchsorted(off,b,side="right")-1)
    print("first bad in unit", u0, "len", ulen, "rel", int(rel[b]), "rel%64", int(rel[b])%64, "got f,p", int(f[b]), int(p[b]), "exp", int(of[b]), int(op[b]), "exp pred rel", int(op[b])-(u0-int(off[r])) if op[b]>=0 else -1, "dist", int(rel[b])-(int(op[b])-(u0-int(off[r]))) if op[b]>=0 else None)
    if len(seen)>=12: break
