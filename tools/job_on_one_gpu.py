import sys,time,json
sys.path.insert(0,'/root/repo')
import torch
from world_amd import synth, distributed as wd
from world_amd.api import WorldHip
FS=48000
n_job=int(sys.argv[1]) if len(sys.argv)>1 else 512
sb=int(sys.argv[2]) if len(sys.argv)>2 else 32
nl=int(sys.argv[3]) if len(sys.argv)>3 else 2
dev=torch.device('cuda',0)
lengths=[240000]*n_job
xs={i: synth.utterance(i, FS, 5.0, device=dev) for i in range(n_job)}
wh=WorldHip(device=0)
lanes=None
if nl==1: lanes=[]
ph={}
def step(): return wd.analyze_sharded(xs, FS, lengths=lengths, packer=wh, sub_batch=sb, timings=ph, **({'lanes':[(torch.cuda.current_stream(), wh.analyze_packed)]} if nl==1 else {}))
res=step(); torch.cuda.synchronize(); ph.clear()
t=time.perf_counter(); n=0
while n<3: res=step(); n+=1
torch.cuda.synchronize(); dt=time.perf_counter()-t
fr=sum(res.n_frames)
# check one utterance vs lone
i=n_job//2+1
tp1,f01,sp1,ap1,nf1=wh.analyze(xs[i].unsqueeze(0),FS); tp,f0,sp,ap=res.utterance(i); k=int(nf1[0])
print(json.dumps({"job":n_job,"sub_batch":sb,"lanes":nl,"frames_per_s":fr*n/dt,"ms_per_step":dt/n*1e3,"same":bool(torch.equal(sp,sp1[0,:k]) and torch.equal(ap,ap1[0,:k]) and torch.equal(f0,f01[0,:k]))}))
