import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'attn_' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
by=collections.defaultdict(list)
for r in rows:
    k=r['Kernel_Name'].split('(')[0].replace('omnipq::','')
    g=(r.get('Grid_Size') or r.get('Grid_Size_X'), )
    by[(k,g)].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in by.items():
    n=len(v); half=n//2
    import statistics
    print(k, n, 'p=0 median %.1f us'%statistics.median(v[:half]), 'p=0.1 median %.1f us'%statistics.median(v[half:]))
