import sys, numpy as np, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from fiesta_amd.sharded import ShardedESDFMap
P=(0.70,0.35,0.12,0.97,0.80)
gs,res=(72,64,80),0.1
sm=ShardedESDFMap((0,0,0),res,gs,2)
sm.SetParameters(*P); sm.SetOriginalRange()
print({r:i for r,i in sm.infos.items()}, flush=True)
allv=np.stack(np.meshgrid(*[np.arange(n) for n in gs],indexing="ij"),-1).reshape(-1,3).astype(np.int32)
sm.SetOccupancy(allv,0); print("obs",flush=True)
print(sm.UpdateOccupancy(True), sm.last_insert, sm.last_delete, flush=True)
t=time.time()
# manual update with prints
for r,sh in sm.shards.items(): print("seed", r, {k:v for k,v in sh.esdf_seed().items() if k in("inserted","deleted")}, flush=True)
print("sweep0", sm._sweep(), flush=True)
for it in range(6):
    for r,sh in sm.shards.items():
        n,st=sh.relax_pending(); print(" relax",r,n,st["rounds"],st["tile_visits"],flush=True)
    c=sm._sweep(); print("sweep",it,c,flush=True)
    if c==0: break
rng=np.random.RandomState(3)
S=(rng.rand(500,3)*gs).astype(np.int32)
S[:60,0]=rng.randint(34,38,60)
for c in range(3):
    sm.SetOccupancy(S,1); print("occ cycle",c,sm.UpdateOccupancy(True), sm.last_insert, sm.last_delete, flush=True)
for r,sh in sm.shards.items(): print("seed", r, {k:v for k,v in sh.esdf_seed().items() if k in("inserted","deleted")}, flush=True)
print("sweep0", sm._sweep(), flush=True)
for it in range(12):
    for r,sh in sm.shards.items():
        t=time.time(); n,st=sh.relax_pending(); print(" relax",r,n,st["rounds"],st["tile_visits"],"%.3fs"%(time.time()-t),flush=True)
    c=sm._sweep(); print("sweep",it,c,flush=True)
    if c==0: break
