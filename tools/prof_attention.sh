cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pa && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o pa -- python $GRAFT_REPO_ROOT/tools/bench_attention.py > /dev/null 2>&1; python - <<EOP
import csv,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open("/tmp/pa/pa_kernel_trace.csv")):
    n=r["Kernel_Name"]
    if "attn" in n:
        d[(n.split("(")[0], r["Grid_Size_X"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items():
    v=sorted(v); print(k, len(v), "median %.1f us"%v[len(v)//2], "min %.1f"%v[0], "p25 %.1f p75 %.1f"%(v[len(v)//4], v[3*len(v)//4]))
EOP
