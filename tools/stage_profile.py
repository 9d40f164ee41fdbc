import sys
sys.path[:0]=['/root/repo','/root/repo/3d-gaussian-splatting_amd']
import torch, numpy as np
from gs_frame import FrameRenderer
from gs_scene import CONFIGS, make_camera, make_scene
dev=torch.device('cuda:0')
for cfg in (sys.argv[1:] or ('cfg2','cfg4','cfg5')):
    deg=3 if cfg.endswith('_deg3') else 2   # e.g. cfg4_deg3: BASELINE config 4 with true degree-3 SH (48 coefficients)
    n,W,H,use_sh=CONFIGS[cfg.replace('_deg3','')]
    scene=make_scene(n,W,H,seed=2023,use_sh=use_sh,sh_degree=deg); cam=make_camera(W,H)
    params=[torch.from_numpy(a).to(dev) for a in (scene.pos,scene.quat,scene.scale,scene.opa,scene.rgb)]
    r=FrameRenderer(dev,max_pairs=1<<20,training=True,auto_grow=True)
    r.forward(*params,cam); st=r.stats(); r.max_pairs=int(st.pairs*1.1)+4096; r.auto_grow=False
    img,_=r.forward(*params,cam)
    prof=[r.profile_forward(*params,cam) for _ in range(8)][3:]
    fw={k:round(float(np.median([p[k] for p in prof])),4) for k in prof[0]}
    g=torch.sign(img-0.5)/img.numel()
    pb=[r.profile_backward(g) for _ in range(6)][2:]
    bw={k:round(float(np.median([p[k] for p in pb])),4) for k in pb[0]}
    print(cfg, "V",st.visible,"M",st.pairs, "fwd",fw,"bwd",bw, flush=True)
