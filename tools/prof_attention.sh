cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pa && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o pa -- python $GRAFT_REPO_ROOT/tools/bench_attention.py > /dev/null 2>&1; python - <<EOP
import csv,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open("/tmp/pa/pa_kernel_trace.csv")):
    n=r["Kernel_Name"]
    if "attn" in n:
        d[(n.split("(")[0], r["Grid_Size_X"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
cfg=["S=512 p=0","S=1024 p=0","S=512 p=.1","S=1024 p=.1"]
for k,v in d.items():
    n=len(v)//4 if "dkdv" not in k[0] else len(v)//2
    for c in range(len(v)//n):
        w=sorted(v[c*n:(c+1)*n]); print(k, "chunk",c, len(w), "median %.1f us"%w[len(w)//2], "min %.1f"%w[0])
EOP
