import sys, time, json, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import __graft_entry__ as ge
ge.load_package()
from empanada_napari_amd import synth
from empanada_napari_amd.engines import HipPanopticDeepLab
from test_regnet import regnet_model
res={}
for tag, prec, batches in (('x','fp32',(1,4)),('y','fp32',(1,4)),('x','fp16',(1,4,16)),('y','fp16',(1,4,16))):
    cfg,P=regnet_model(tag)
    m=HipPanopticDeepLab(P,cfg,folded=True,precision=prec)
    for B in batches:
        x=torch.from_numpy(synth.em_tiles(B,1024,seed=3))[:,None].cuda()
        sub,mul=0.57571*255,1/(0.12765*255)
        m(x,2,False,sub=sub,mul=mul); torch.cuda.synchronize()
        t=time.perf_counter(); R=3
        for _ in range(R): m(x,2,False,sub=sub,mul=mul)
        torch.cuda.synchronize(); dt=(time.perf_counter()-t)/R
        res[f'{cfg["arch"]}/{cfg["encoder"]} {prec} batch {B} 1024^2']={'ms':round(dt*1e3,1),'tiles_per_s':round(B/dt,1),'TFLOPs':round(m.last_flops()/dt/1e12,1)}
        print(list(res.items())[-1], flush=True)
json.dump(res,open('/root/repo/gpurun_out/regnet_rate.json','w'),indent=1)
