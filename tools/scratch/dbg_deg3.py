import sys, copy
sys.path[:0]=['/root/repo','/root/repo/3d-gaussian-splatting_amd','/root/repo/tests']
import torch, numpy as np
from gs_frame import FrameRenderer
from gs_scene import make_camera, make_scene
from gs_testutil import to_torch
gpu=torch.device('cuda:0')
W,H=320,208
s3=make_scene(30000,W,H,seed=21,use_sh=True,sh_degree=3); cam=make_camera(W,H,yaw_deg=2.0)
c=s3.rgb.reshape(-1,3,16); c[:,:,9:]=0
s2=copy.deepcopy(s3); s2.rgb=np.ascontiguousarray(c[:,:,:9]).reshape(-1,27)
gimg=torch.from_numpy(np.random.default_rng(3).normal(size=(H,W,3)).astype(np.float32)).to(gpu)
outs=[]
for sc in (s3,s2):
    params=to_torch(sc,gpu,requires_grad=True)
    r=FrameRenderer(gpu,max_pairs=400_000,training=True)
    img=r.render(*params,cam); img.backward(gimg)
    outs.append((img.detach().clone(),[t.grad.clone() for t in params]))
(i3,g3),(i2,g2)=outs
d=(i3-i2).abs(); print("img maxdiff",float(d.max()),"n diff",int((d>0).sum()),"of",d.numel(),"nan",int(torch.isnan(i3).sum()))
idx=torch.nonzero(d>0)[:5]; print(idx.tolist())
for a,b,n in zip(g3[:4],g2[:4],("pos","quat","scale","opa")):
    print(n,float((a-b).abs().max()),float(a.abs().max()))
a=g3[4].view(-1,3,16)[:,:,:9]; b=g2[4].view(-1,3,9); print("rgb",float((a-b).abs().max()),float(a.abs().max()))
