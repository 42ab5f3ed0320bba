import sqlite3, sys
from collections import defaultdict
c=sqlite3.connect(sys.argv[1])
rows=c.execute("select name,start,end from kernels order by start").fetchall()
fps=[r for r in rows if 'fps_kernel<1024, 4, true' in r[0]]
nsteps=int(sys.argv[2]) if len(sys.argv)>2 else 2
t0=fps[-1-nsteps][1]; t1=fps[-1][1]
sel=[r for r in rows if t0<=r[1]<t1]
agg=defaultdict(lambda:[0,0.0])
for n,s,e in sel:
    agg[n][0]+=1; agg[n][1]+=(e-s)/1e3
tot=sum(v[1] for v in agg.values())
print('steady: %d steps, kernels/step %.0f, busy us/step %.1f, wall us/step %.1f'%(nsteps,len(sel)/nsteps,tot/nsteps,(t1-t0)/nsteps/1e3))
top=int(sys.argv[3]) if len(sys.argv)>3 else 40
for n,(cnt,us) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:top]:
    print(f"{us/nsteps:10.1f} us/step x{cnt/nsteps:6.1f}  {n[:130]}")
