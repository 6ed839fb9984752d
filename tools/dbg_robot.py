import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
torch.cuda.init()
import oracle_py as O
from voxblox_amd import capi, scenes
VOXEL=0.05; TRUNC=0.2
frames=[scenes.room_frame(4*k,100,f=80.0,width=160,height=120) for k in range(3)]
sph=dict(clear_sphere_radius=0.6, occupied_sphere_radius=3.8)
gm=capi.Map(VOXEL,16,max_blocks=4096)
gt=capi.tsdf_cfg(default_truncation_distance=TRUNC,max_ray_length_m=3.2)
ge=capi.esdf_cfg(min_distance_m=TRUNC/2,min_diff_m=0.0,**sph)
for pose,pts,col in frames:
    gm.integrate(capi.TSDF_SIMPLE,gt,pose[0],pose[1],pts,col)
gm.esdf_update(ge,batch=True,clear_updated_flag=False)
def stats(tag):
    nz=0; tot=0; hall=0
    for i in gm.block_indices(capi.LAYER_ESDF):
        v,u,_=gm.block_download(i,capi.LAYER_ESDF)
        nz+=int((np.abs(v["parent"]).sum(1)>0).sum()); tot+=int(v["observed"].sum()); hall+=int(v["hallucinated"].sum())
    print(tag,"blocks",gm.num_blocks(capi.LAYER_ESDF),"observed",tot,"halluc",hall,"nonzero parents",nz,gm.counters())
stats("batch")
p0=frames[0][0][0]
for p in (p0,p0,p0+np.array([0.1,-0.05,0.05],np.float32)):
    gm.esdf_add_new_robot_position(ge,p)
    stats("after add")
    gm.esdf_update(ge,batch=False,clear_updated_flag=False)
    stats("after update")
