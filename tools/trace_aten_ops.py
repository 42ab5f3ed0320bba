#!/usr/bin/env python
"""Which source lines of the model still issue PyTorch (aten) GPU ops in one eager training step?  A TorchDispatchMode
logs every op with the innermost repo frame; used to hunt stray casts / copies / zero-fills.  Run on a GPU box."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CAPTURE=bool(os.environ.get("TRACE_CAPTURE"))     # log the ops recorded INTO the captured step instead of an eager step's
sys.argv=["bench.py","--graph","on" if CAPTURE else "off","--no-op-timing","--no-cpu-baseline"]
import bench, torch
from torch.utils._python_dispatch import TorchDispatchMode
args=bench.parse()
dev=torch.device("cuda",0)
import pointnet2_utils, synth
net=bench.build_model(0).to(dev); net.train()
pool=[synth.make_clouds(100+i,args.batch,args.points,kind="room").to(dev) for i in range(3)]
agg=collections.defaultdict(lambda:[0,0])
shapes=collections.Counter()
seq=[]
WATCH=("copy_","_to_copy","clone","cat","fill_","zeros","add","div","mul","zero_")
class M(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out=func(*args, **(kwargs or {}))
        if CAPTURE and not torch.cuda.is_current_stream_capturing(): return out
        name=func.__name__ if hasattr(func,"__name__") else str(func)
        base=str(func).split(".")[1] if "." in str(func) else str(func)
        if torch.is_tensor(out) and out.is_cuda and base not in ("view","_unsafe_view","transpose","t","slice","select","unsqueeze","squeeze","expand","as_strided","detach","alias","permute","reshape","empty","empty_like","empty_strided","split","split_with_sizes","unbind","narrow","_reshape_alias","view_as","lift_fresh","chunk","record_stream","is_same_size","contiguous"):
            fr="?"
            for f in reversed(traceback.extract_stack()):
                if ("omni-pq_amd" in f.filename or f.filename.endswith("bench.py")):
                    fr=f"{os.path.basename(f.filename)}:{f.lineno}"; break
            n=0
            if torch.is_tensor(out): n=out.numel()*out.element_size()
            agg[(base,fr)][0]+=1; agg[(base,fr)][1]+=n
            if os.environ.get("TRACE_ALL"):
                node=None
                try: node=torch._C._current_autograd_node()
                except Exception: pass
                seq.append((base, fr, tuple(out.shape), str(out.dtype).replace("torch.",""), node.name() if node is not None else "fwd"))
            if fr=="?" and os.environ.get("TRACE_SHAPES"):
                node=None
                try: node=torch._C._current_autograd_node()
                except Exception: pass
                # gradient accumulation runs between nodes: the node named is the one whose backward produced the addend
                shapes[(base,tuple(out.shape),str(out.dtype).replace("torch.","")+" after "+(node.name() if node is not None else "-"))]+=1
        return out
if CAPTURE:
    with M():
        step,_=bench.make_step(net,net,pool,args,torch.bfloat16,1)
else:
    step,_=bench.make_step(net,net,pool,args,torch.bfloat16,1)
    for i in range(3): step(i)
    torch.cuda.synchronize()
    with M():
        step(0)
torch.cuda.synchronize()
print("ops watched:", sum(v[0] for v in agg.values()))
byfile=collections.defaultdict(int)
for (b,fr),(c,n) in agg.items(): byfile[fr.split(":")[0]]+=c
print(sorted(byfile.items(), key=lambda kv:-kv[1]))
for (b,fr),(c,n) in sorted(agg.items(), key=lambda kv:-kv[1][0])[:70]:
    print(f"{c:5d}x {n/1e6:9.2f} MB  {b:10s} {fr}")

if shapes:
    print("ops issued from autograd's backward (no repo frame), by shape:")
    for (b,sh,dt),c in shapes.most_common(40):
        print(f"{c:5d}x  {b:10s} {dt} {sh}")

if seq:
    print("every watched op in issue order (op, site, shape, dtype, autograd node):")
    for i, r in enumerate(seq):
        print(f"{i:4d} {r[0]:14s} {r[1]:28s} {str(r[2]):22s} {r[3]:9s} {r[4]}")
