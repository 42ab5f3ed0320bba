import os, sys
REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO,"tests")); sys.path.insert(0, REPO)
import torch, conftest, bench
dev=torch.device("cuda",0)
mode=sys.argv[1] if len(sys.argv)>1 else "full"
def run():
    torch.manual_seed(3)
    net=bench.build_model(0).to(dev).train()
    xyz=(torch.rand(4, 8192, 3, generator=torch.Generator().manual_seed(5))*3).to(dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ep=net.backbone(xyz, {})
    f=ep["sa4_features"] if mode=="sa_only" else ep["fp2_features"]
    w=torch.randn(f.shape, generator=torch.Generator().manual_seed(9)).to(dev)
    (f.float()*w).mean().backward()
    return f.detach().float(), {n:p.grad.clone() for n,p in net.named_parameters() if p.grad is not None}
if mode=="nofprows":
    import pointnet2_modules
    pointnet2_modules.PointnetFPModule._forward_rows = lambda self,*a: None
f0,g0=run(); f1,g1=run()
rel=lambda a,b: float((a-b).norm()/(b.norm()+1e-30))
print(mode, "same run twice: fwd", rel(f1,f0))
for n in sorted(g0, key=lambda n:-rel(g1[n],g0[n]))[:6]:
    print(f"  {rel(g1[n],g0[n]):.3e} |g| {float(g0[n].norm()):.2e} {n}")
