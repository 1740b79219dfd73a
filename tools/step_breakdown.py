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

# category roll-up (ms per step)
def cat(n):
    if n.startswith(('bn_', 'void bn_')): return 'fused BN kernels (csrc/bn.hip)'
    if 'wgrad' in n: return 'convolution weight gradients, 1x1 and k x k (csrc/wgrad.hip)'
    if 'maxpool3s2' in n: return 'stem max-pool (csrc/pool.hip)'
    if any(k in n for k in ('ema_flat', 'sgd_flat', 'bf16_image')): return 'EMA + optimizer kernels (csrc/ema.hip, sgd.hip)'
    if any(k in n for k in ('rowkey', 'dense_', 'quantile', 'feat_', 'pool_', 'compose', 'strided_gather', 'gather_rows',
                            'corr_iou', 'enqueue', 'keys_split', 'mean_kernel', 'step_scalars', 'step_tail', 'step_post', 'loss_post', 'densecl_match')): return 'loss-section kernels (csrc/*.hip)'
    if any(k in n for k in ('igemm', 'ck::', '_ZN2ck', 'SubTensorOp', 'Cijk', 'miopen', 'MIOpen', 'naive_conv', 'gemm')): return 'MIOpen / CK / hipBLASLt (convolutions, GEMMs)'
    if 'rocclr' in n: return 'runtime fills / copies'
    return 'ATen elementwise / pooling / reductions'
cats = collections.defaultdict(lambda: [0, 0])
for n, (d, c) in agg.items():
    cats[cat(n)][0] += d; cats[cat(n)][1] += c
print("--- by category")
for n, (d, c) in sorted(cats.items(), key=lambda kv: -kv[1][0]):
    print(f"{d/1e6/nsteps:8.3f} ms/step {c/nsteps:7.1f} launches  {n}")
