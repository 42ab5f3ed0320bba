import os, sys, time
REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO,"tests")); sys.path.insert(0, REPO)
import torch, conftest, bench, synth
dev=torch.device("cuda",0)
torch.manual_seed(0)
net=bench.build_model(0).to(dev)
pc=synth.make_clouds(3, 8, 40000, kind="room").to(dev)
net.train()
with torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(2): ep=net({"point_clouds":pc}); 
net.eval()
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    ep=net({"point_clouds":pc})
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(5): ep=net({"point_clouds":pc})
    torch.cuda.synchronize(); print("eval bf16 fwd ms", (time.perf_counter()-t)/5*1e3, len(ep))
with torch.no_grad():
    ep32=net({"point_clouds":pc})
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(3): ep32=net({"point_clouds":pc})
    torch.cuda.synchronize(); print("eval f32 fwd ms", (time.perf_counter()-t)/3*1e3)
bad=[k for k in ep if torch.is_tensor(ep[k]) and ep[k].is_floating_point() and not torch.isfinite(ep[k].float()).all()]
print("non-finite:", bad)
for k in ("last_center","last_objectness_scores","seed_features"):
    a,b=ep[k].float(),ep32[k].float(); print(k, float((a-b).norm()/(b.norm()+1e-30)))
# long run: memory stable?
net.train()
torch.cuda.reset_peak_memory_stats()
m0=None
for i in range(30):
    for p in net.parameters(): p.grad=None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ep=net({"point_clouds":pc}); loss=bench.loss_of(ep)
    loss.backward()
    if i==5: torch.cuda.synchronize(); m0=torch.cuda.memory_allocated()
torch.cuda.synchronize(); print("allocated after 6 / 30 steps (MB):", m0/1e6, torch.cuda.memory_allocated()/1e6, "peak", torch.cuda.max_memory_allocated()/1e6)
