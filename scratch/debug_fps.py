import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import conftest, torch, capi, synth
from oracle import oracle_ext
d = torch.device('cuda:0')
def run(kind, b, n, m, seed=5):
    if kind == 'adv': xyz = synth.adversarial_cloud(seed, b, max(n,32))[:, :n].contiguous()
    elif kind == 'rand': xyz = torch.rand((b, n, 3), generator=torch.Generator().manual_seed(seed))*2+0.1
    else: xyz = synth.make_clouds(seed, b, n, kind=kind)
    want = oracle_ext.furthest_point_sampling(xyz, m)
    got, tmp = capi.fps(xyz.to(d), m)
    got = got.cpu()
    bad = (got != want).nonzero()
    print(kind, b, n, m, 'mismatches', len(bad), flush=True)
    if len(bad):
        bi, j = bad[0].tolist()
        print('  first at', bi, j, 'got', int(got[bi,j]), 'want', int(want[bi,j]), 'prev', want[bi, max(j-3,0):j].tolist(), got[bi, max(j-3,0):j].tolist())
        # recompute temp up to round j on CPU (double check) 
        p = xyz[bi]
        t = torch.full((n,), 1e10)
        for jj in range(j):
            c = p[want[bi, jj].long()]
            dd = ((p - c)**2).sum(-1)
            t = torch.minimum(t, dd)
        kg, kw = int(got[bi,j]), int(want[bi,j])
        print('  approx temp got/want', float(t[kg]), float(t[kw]), 'k mod 512', kg % 512, kw % 512, 'norm2', float((p[kg]**2).sum()), float((p[kw]**2).sum()))
for case in [('rand',1,7,7),('rand',1,64,16),('rand',2,100,37),('rand',1,256,64),('rand',1,300,100),('rand',1,512,128),('rand',1,1024,128),('rand',1,2048,128),('room',2,4096,512),('room',1,8192,64),('room',1,8193,64),('room',2,40000,256)]:
    run(*case)
