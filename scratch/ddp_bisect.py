import os, sys, socket
REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO,"tests")); sys.path.insert(0, REPO)
import torch, torch.distributed as dist, torch.multiprocessing as mp

def build(which, dev):
    import conftest, pointnet2_modules, rows_mlp
    from procedural import load_procedural
    torch.manual_seed(3)
    if which == "sa":
        m = pointnet2_modules.PointnetSAModuleVotes(npoint=256, radius=0.5, nsample=32, mlp=[0, 64, 64, 128], use_xyz=True, normalize_xyz=True)
    elif which == "sa2":
        class M(torch.nn.Module):
            def __init__(s):
                super().__init__()
                s.a = pointnet2_modules.PointnetSAModuleVotes(npoint=256, radius=0.5, nsample=32, mlp=[0, 64, 64, 128], use_xyz=True, normalize_xyz=True)
                s.b = pointnet2_modules.PointnetSAModuleVotes(npoint=64, radius=1.0, nsample=16, mlp=[128, 128, 128, 256], use_xyz=True, normalize_xyz=True)
            def forward(s, xyz, f=None):
                x1, f1, _ = s.a(xyz, None); x2, f2, _ = s.b(x1, f1); return x2, f2, None
        m = M()
    elif which == "fp":
        class M(torch.nn.Module):
            def __init__(s):
                super().__init__()
                s.a = pointnet2_modules.PointnetSAModuleVotes(npoint=256, radius=0.5, nsample=32, mlp=[0, 64, 64, 128], use_xyz=True, normalize_xyz=True)
                s.b = pointnet2_modules.PointnetSAModuleVotes(npoint=64, radius=1.0, nsample=16, mlp=[128, 128, 128, 256], use_xyz=True, normalize_xyz=True)
                s.fp = pointnet2_modules.PointnetFPModule(mlp=[256 + 128, 256, 96])
            def forward(s, xyz, f=None):
                x1, f1, _ = s.a(xyz, None); x2, f2, _ = s.b(x1, f1); return None, s.fp(x1, x2, f1, f2), None
        m = M()
    elif which == "rows":
        class M(torch.nn.Module):
            def __init__(s):
                super().__init__()
                s.c1 = torch.nn.Conv1d(3, 288, 1); s.b1 = torch.nn.BatchNorm1d(288); s.c2 = torch.nn.Conv1d(288, 288, 1); s.b2 = torch.nn.BatchNorm1d(288); s.c3 = torch.nn.Conv1d(288, 64, 1)
            def forward(s, xyz, f=None):
                x = xyz.reshape(-1, 3)
                y = rows_mlp.run(x, [rows_mlp.Layer(s.c1.weight, s.c1.bias, s.b1), rows_mlp.Layer(s.c2.weight, s.c2.bias, s.b2), rows_mlp.Layer(s.c3.weight, s.c3.bias)], True)
                return None, y, None
        m = M()
    elif which in ("backbone", "bbvote", "bbdec"):
        sys.path.insert(0, REPO)
        import bench
        net = bench.build_model(0)
        for mm in net.modules():
            if isinstance(mm, torch.nn.Dropout): mm.p = 0.0
            if hasattr(mm, "dropout") and isinstance(getattr(mm, "dropout"), float): mm.dropout = 0.0
        class M(torch.nn.Module):
            def __init__(s):
                super().__init__(); s.net = net
            def forward(s, xyz, f=None):
                ep = s.net.backbone(xyz, {})
                feat = ep["fp2_features"]
                if which == "backbone": return None, feat, None
                if which == "bbvote":
                    vx, vf = s.net.vote(ep["fp2_xyz"], feat); return None, vf, None
                from pq_transformer import conv1x1
                q = conv1x1(feat[:, :, :512].contiguous(), s.net.decoder_query_proj); k = conv1x1(feat, s.net.decoder_key_proj)
                out = s.net.decoder[0](q, k, ep["fp2_xyz"][:, :512].contiguous(), ep["fp2_xyz"]); return None, out, None
        m = M()
        torch.manual_seed(3)
        return m.to(dev).train()
    return load_procedural(m, 3).to(dev).train()

def flat(net): return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in net.parameters()]).cpu()

def worker(rank, world, port, which):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dev=torch.device("cuda",0); torch.cuda.set_device(dev)
    npts = 8192 if which.startswith('b') else 2048
    xyz=(torch.rand(world*2, npts, 3, generator=torch.Generator().manual_seed(5))*3).to(dev)
    ref=None
    if rank==0:
        net=build(which, dev)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            _, f, _ = net(xyz, None)
        w=torch.randn(f.shape, generator=torch.Generator().manual_seed(9)).to(dev)
        fref=f.detach().float().clone()
        (f.float()*w).mean().backward(); ref=flat(net); names=[n for n,_ in net.named_parameters()]; sizes=[p.numel() for p in net.parameters()]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net=build(which, dev)
    mine=xyz[rank*2:(rank+1)*2].contiguous()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        _, f, _ = net(mine, None)
    wfull=torch.randn((f.shape[0]*world,)+tuple(f.shape[1:]), generator=torch.Generator().manual_seed(9)).to(dev) if f.dim()>2 else None
    if f.dim()==2:
        wfull=torch.randn((f.shape[0]*world, f.shape[1]), generator=torch.Generator().manual_seed(9)).to(dev)
        w=wfull[rank*f.shape[0]:(rank+1)*f.shape[0]]
    else:
        w=wfull[rank*2:(rank+1)*2]
    if rank==0:
        d=f.detach().float()-fref[:f.shape[0]]
        print("forward diff rel", float(d.norm()/fref[:f.shape[0]].norm()), "max", float(d.abs().max()), flush=True)
    (f.float()*w).mean().backward()
    g=torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in net.parameters()])
    dist.all_reduce(g); g=(g/world).cpu()
    if rank==0:
        off=0
        for n,k in zip(names,sizes):
            a,b=g[off:off+k],ref[off:off+k]; off+=k
            if float(b.norm()) > 0 or float(a.norm()) > 0:
                print(f"{float((a-b).norm()/(b.norm()+1e-30)):.3e} ratio {float((a*b).sum()/((b*b).sum()+1e-30)):.4f} |ref| {float(b.norm()):.3e} {n}")
    dist.barrier(); dist.destroy_process_group()

if __name__=="__main__":
    which=sys.argv[1]
    s=socket.socket(); s.bind(("127.0.0.1",0)); port=s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port, which), nprocs=2, join=True)
