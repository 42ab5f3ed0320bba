import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv=["bench.py","--graph","off","--no-op-timing","--no-cpu-baseline"]
import bench, torch
from torch.profiler import profile, ProfilerActivity
args=bench.parse()
dev=torch.device("cuda",0)
import pointnet2_utils, synth
net=bench.build_model(0).to(dev); net.train()
pool=[synth.make_clouds(100+i,args.batch,args.points,kind="room").to(dev) for i in range(3)]
step,_=bench.make_step(net,net,pool,args,torch.bfloat16,1)
for i in range(4): step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(2): step(i)
    torch.cuda.synchronize()
# attribute aten ops (CPU side events with device time) to the innermost frame inside the repo
agg=collections.defaultdict(lambda:[0.0,0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU: continue
    dt = getattr(ev, "self_device_time_total", 0) or 0
    if dt <= 0: continue
    if not ev.name.startswith("aten::"): continue
    frame="?"
    for fr in (ev.stack or []):
        if "/root/repo" in fr or "omni-pq_amd" in fr or "bench.py" in fr:
            frame=fr.split("/")[-1][:70]; break
    k=(ev.name, frame)
    agg[k][0]+=dt; agg[k][1]+=1
tot=sum(v[0] for v in agg.values())
print("aten self device time per step (us):", tot/2)
for (name,frame),(t,c) in sorted(agg.items(), key=lambda kv:-kv[1][0])[:70]:
    print(f"{t/2:8.1f} us {c/2:6.1f}x  {name:34s} {frame}")
