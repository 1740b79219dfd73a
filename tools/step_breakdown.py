import csv, collections, sys
path=sys.argv[1]; nsteps=int(sys.argv[2]) if len(sys.argv)>2 else 5
rows=list(csv.DictReader(open(path)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
ema=[i for i,r in enumerate(rows) if 'ema_flat' in r['Kernel_Name']]
a,b=ema[-nsteps-1],ema[-1]
seg=rows[a:b]
t0=int(seg[0]['Start_Timestamp']); t1=int(rows[b]['Start_Timestamp'])
wall=(t1-t0)/1e6/nsteps
agg=collections.defaultdict(lambda:[0,0])
busy=0
for r in seg:
    d=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    n=r['Kernel_Name']
    agg[n][0]+=d; agg[n][1]+=1; busy+=d
print(f"wall/step {wall:.2f} ms, kernel-busy/step {busy/1e6/nsteps:.2f} ms, launches/step {len(seg)/nsteps:.0f}")
top=sorted(agg.items(), key=lambda kv:-kv[1][0])
for n,(d,c) in top[:int(sys.argv[3]) if len(sys.argv)>3 else 45]:
    print(f"{d/1e6/nsteps:8.3f} ms/step {c/nsteps:7.1f} calls  avg {d/c/1e3:8.1f} us  {n[:130]}")
