import sys, numpy as np
sys.path.insert(0,'.')
import fiesta_amd
from bench import Workload
G=256; res=0.1
m=fiesta_amd.ESDFMap((0,0,0),res,(G*res,)*3)
m.SetParameters(0.70,0.35,0.12,0.97,0.80); m.SetOriginalRange()
m.SetOccupancyBox((0,0,0),(G-1,)*3,0); m.UpdateOccupancy(True); print(m.UpdateESDF())
w=Workload(G,6250)
for _ in range(3):
    m.SetOccupancy(w.initial(),1,want_ret=False); m.UpdateOccupancy(True)
st=m.UpdateESDF(); print({k:st[k] for k in ('inserted','deleted','bulk','rounds','observed_voxels','occupied_voxels','relax_ms','ft_rows_ms','ft_plane_ms','ft_x_ms','ft_overflow')})
for s in range(3):
    new,old=w.next_step()
    for c in range(3):
        m.SetOccupancy(new,1,want_ret=False)
        if c==2: m.SetOccupancy(old,0,want_ret=False)
        m.UpdateOccupancy(True)
    st=m.UpdateESDF(); print({k:st[k] for k in ('inserted','deleted','bulk','rounds','observed_voxels','occupied_voxels','relax_ms','ft_rows_ms','ft_plane_ms','ft_x_ms','ft_overflow')})
