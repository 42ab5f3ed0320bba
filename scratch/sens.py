import os, sys
REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO,"tests")); sys.path.insert(0, REPO)
import torch, conftest, bench
dev=torch.device("cuda",0)
def run(eps_scale):
    torch.manual_seed(3)
    net=bench.build_model(0).to(dev).train()
    with torch.no_grad():
        # a 1e-6 relative nudge of ONE BatchNorm weight vector in sa4: the kind of difference another summation order makes
        net.backbone.sa4.mlp_module.layer0.bn.bn.weight.mul_(1.0+eps_scale)
    xyz=(torch.rand(4, 8192, 3, generator=torch.Generator().manual_seed(5))*3).to(dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ep=net.backbone(xyz, {})
    f=ep["fp2_features"]
    w=torch.randn(f.shape, generator=torch.Generator().manual_seed(9)).to(dev)
    (f.float()*w).mean().backward()
    return f.detach().float(), {n:p.grad.clone() for n,p in net.named_parameters() if p.grad is not None}
f0,g0=run(0.0); f1,g1=run(0.0); f2,g2=run(1e-6)
rel=lambda a,b: float((a-b).norm()/(b.norm()+1e-30))
print("same run twice: fwd", rel(f1,f0), "max grad rel", max(rel(g1[n],g0[n]) for n in g0))
print("1e-6 nudge   : fwd", rel(f2,f0))
for n in sorted(g0, key=lambda n:-rel(g2[n],g0[n]))[:8]:
    print(f"  {rel(g2[n],g0[n]):.3e} {n}")
