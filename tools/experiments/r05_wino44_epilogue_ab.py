import os, sys
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd'))
import torch
from hipvsr import lib as L
if os.environ.get('STAMPS_LIB'):
    L.LIB_PATH = os.path.join(ROOT,'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd','hipvsr',os.environ['STAMPS_LIB'])
from hipvsr.hip_ops import HipOps
from hipvsr.plans import NetPlans, Src
from hipvsr.spec import state_dict_spec
from oracle import refinenet_oracle as orc
def timed(fn, reps=200, warm=50):
    for _ in range(warm): fn()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/reps*1e3
B,H,W=8,128,128
dev=torch.device('cuda:0'); cfg=orc.exp1_x4_config(); P,ops=NetPlans(cfg),HipOps(dev); spec=state_dict_spec(cfg)
plan=P.lstm[('forward',1)]['full']
g=torch.Generator('cpu').manual_seed(1)
w,b=(torch.randn(*spec[plan.wkey],generator=g)*0.03).to(dev),(torch.randn(*spec[plan.bkey],generator=g)*0.1).to(dev)
ops.pack(plan,w,b)
x,h,c=(torch.randn(B,H,W,64,generator=g).to(dev) for _ in range(3))
ho,co,go=torch.empty_like(x),torch.empty_like(x),torch.empty(B,H,W,256,device=dev)
vx,vh=ops.wino44_v(B,H,W,64)[0],ops.wino44_v(B,H,W,64)[0]
ops.wino44_transform(Src(x),B,H,W,vx); ops.wino44_transform(Src(h),B,H,W,vh)
for name,lstm in (('gates+c+h',dict(hd=64,c_prev=c,h_out=ho,c_out=co,gates_out=go)),('no gates',dict(hd=64,c_prev=c,h_out=ho,c_out=co,gates_out=None)),('no state',dict(hd=64,c_prev=None,h_out=ho,c_out=co,gates_out=go))):
    print(os.environ.get('STAMPS_LIB','product'), name, round(timed(lambda: ops.wino44_cell(plan,[vx,vh],B,H,W,lstm)),1),'us')
