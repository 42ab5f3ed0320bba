import sys, os, time, cProfile, pstats
sys.argv=["bench.py","--graph","off","--no-op-timing","--no-cpu-baseline"]
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
import bench, torch
args=bench.parse()
dev=torch.device("cuda",0)
import pointnet2_utils, synth
net=bench.build_model(0).to(dev); net.train()
pool=[synth.make_clouds(100+i,args.batch,args.points,kind="room").to(dev) for i in range(3)]
step,_=bench.make_step(net,net,pool,args,torch.bfloat16,1)
for i in range(5): step(i)
torch.cuda.synchronize()
pr=cProfile.Profile(); pr.enable()
for i in range(5): step(i)
pr.disable(); torch.cuda.synchronize()
st=pstats.Stats(pr); st.sort_stats("cumtime").print_stats(40)
st.print_callers("named_modules")
st.print_callers("_named_members")
