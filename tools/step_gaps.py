"""Where a step's wall time goes when no kernel runs: idle gaps of the GPU (union over every queue) in a rocprofv3 kernel trace.
usage: step_gaps.py <kernel_trace.csv> [steps=5] [top=25]
Steps are delimited by the EMA kernel (first launch of every step).  Prints wall / busy / idle per step, the idle time
grouped by the kernel that FOLLOWS the gap (what the GPU was waiting to be given), and the largest single gaps."""
import collections
import csv
import sys

path = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in csv.DictReader(open(path))]
rows.sort()
ema = [i for i, r in enumerate(rows) if "ema_flat" in r[2]]
a, b = ema[-nsteps - 1], ema[-1]
seg = rows[a:b]
t0, t1 = seg[0][0], rows[b][0]
busy, idle_by, gaps, cur_end, prev = 0, collections.defaultdict(lambda: [0, 0]), [], seg[0][0], "(step start)"
for s, e, n, q in seg:
    if s > cur_end:
        g = s - cur_end
        idle_by[n[:90]][0] += g
        idle_by[n[:90]][1] += 1
        gaps.append((g, prev[:70], n[:70], q))
        busy_from = s
    else:
        busy_from = cur_end
    if e > cur_end:
        busy += e - max(busy_from, s) if s > cur_end else e - cur_end
        cur_end, prev = e, n
wall = (t1 - t0) / 1e6 / nsteps
queues = collections.Counter(q for _, _, _, q in seg)
print(f"wall/step {wall:.3f} ms   GPU busy (any queue) {busy / 1e6 / nsteps:.3f} ms   idle {wall - busy / 1e6 / nsteps:.3f} ms   launches/step {len(seg) / nsteps:.0f}   queues {dict(queues)}")
print("--- idle time by the kernel that ends the gap (ms/step, gaps/step)")
for n, (d, c) in sorted(idle_by.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{d / 1e6 / nsteps:8.4f} ms  {c / nsteps:6.1f}  {n}")
print("--- largest gaps (us): before -> after [queue of the kernel after]")
for g, p, n, q in sorted(gaps, reverse=True)[:top]:
    print(f"{g / 1e3:8.1f}  {p}  ->  {n}  [{q}]")
