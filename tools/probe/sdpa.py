import torch, time, torch.nn.functional as F
dev='cuda'
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
for (L,S) in [(512,512),(512,1024)]:
    for dt in (torch.bfloat16, torch.float32):
        q=torch.randn(8,8,L,36,device=dev,dtype=dt,requires_grad=True); k=torch.randn(8,8,S,36,device=dev,dtype=dt,requires_grad=True); v=torch.randn(8,8,S,36,device=dev,dtype=dt,requires_grad=True)
        def manual():
            a=torch.matmul(q*(36**-0.5),k.transpose(-1,-2)); p=F.dropout(F.softmax(a.float(),-1),0.1).to(dt); o=torch.matmul(p,v); o.sum().backward()
        def sdpa():
            o=F.scaled_dot_product_attention(q,k,v,dropout_p=0.1); o.sum().backward()
        try:
            print(L,S,dt,'manual %.3f ms'%bench(manual),'sdpa %.3f ms'%bench(sdpa))
        except Exception as e:
            print(L,S,dt,'ERR',repr(e)[:200])
        with torch.no_grad():
            o1=F.scaled_dot_product_attention(q,k,v); a=torch.softmax(torch.matmul(q.float()*(36**-0.5),k.float().transpose(-1,-2)),-1); o2=torch.matmul(a,v.float())
            print('   max diff', float((o1.float()-o2).abs().max()))
from torch.nn.attention import SDPBackend, sdpa_kernel
for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION, SDPBackend.MATH):
    q=torch.randn(8,8,512,36,device=dev,dtype=torch.bfloat16,requires_grad=True); k=torch.randn(8,8,1024,36,device=dev,dtype=torch.bfloat16,requires_grad=True); v=torch.randn_like(k,requires_grad=True)
    try:
        with sdpa_kernel(be):
            def f():
                o=F.scaled_dot_product_attention(q,k,v,dropout_p=0.1); o.sum().backward()
            print(be, '%.3f ms'%bench(f))
    except Exception as e:
        print(be,'ERR',repr(e)[:150])
