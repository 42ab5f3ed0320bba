#!/usr/bin/env python
"""Host-side profile (cProfile) of the EAGER training step: where the ~25 ms of a kernel-by-kernel launched step go on
the host (the captured step replays the same kernels in 10.6 ms).  python tools/host_profile.py"""
import sys, os, time, cProfile, pstats, argparse
sys.argv=["bench.py","--graph","off","--no-op-timing","--no-cpu-baseline"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
args=bench.parse()
dev=torch.device("cuda",0)
import pointnet2_utils, synth
net=bench.build_model(0).to(dev); net.train()
pool=[synth.make_clouds(100+i,args.batch,args.points,kind="room").to(dev) for i in range(3)]
step,_=bench.make_step(net,net,pool,args,torch.bfloat16,1)
for i in range(5): step(i)
torch.cuda.synchronize()
t=time.perf_counter()
for i in range(10): step(i)
t1=time.perf_counter()-t
torch.cuda.synchronize(); t2=time.perf_counter()-t
print("host ms/step",t1*100,"total",t2*100)
pr=cProfile.Profile(); pr.enable()
for i in range(5): step(i)
pr.disable(); torch.cuda.synchronize()
st=pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
