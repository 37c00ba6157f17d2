"""Timeline of the steady-state steps in a rocprofv3 --kernel-trace of scripts/profile_step.py: how much of a step has 0, 1,
2+ kernels in flight, busy time per queue, and the largest holes (with the kernels on either side).
usage: step_timeline.py <trace dir> <steps in the trace> [holes]"""
import csv, re, glob, sys, collections
path = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
n = int(sys.argv[2]); nholes = int(sys.argv[3]) if len(sys.argv) > 3 else 25
def short(nm):
    nm = re.sub(r'\(anonymous namespace\)::', '', nm); nm = re.sub(r'^void ', '', nm); nm = re.sub(r'\(.*$', '', nm)
    nm = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', nm)
    return nm[:60]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '0'), short(r['Kernel_Name'])) for r in csv.DictReader(open(path))]
rows.sort()
# steady state: the last n steps end with an adamw launch each; cut at the adamw ends
ad = [e for s, e, q, k in rows if k.startswith('adamw')]
per = max(1, round(len(ad) / (len(ad) // max(1, len(ad) // n)))) if ad else 1
# adamw launches per step may be > 1: take the last launch of each step = the one followed by a non-adamw kernel
cuts = [rows[i][1] for i in range(len(rows) - 1) if rows[i][3].startswith('adamw') and not rows[i + 1][3].startswith('adamw')]
# behind a hold (profile_step.py HOLD_MS=..): the host's lead shrinks under the tracer, so take the FIRST n steps after the hold
hold_end = max([e for s, e, q, k in rows if 'OnSelf_add<float>' in k and e - s > 200_000] or [0])
if hold_end:
    first = [c for c in cuts if c > hold_end]
    before = [c for c in cuts if c <= hold_end]
    cuts = ([hold_end] + first)[: n + 1]
else:
    cuts = cuts[-(n + 1):] if len(cuts) > n else cuts
t0, t1 = cuts[0], cuts[-1]; nst = len(cuts) - 1
sel = [r for r in rows if r[0] >= t0 and r[1] <= t1 + 1]
print(f"{nst} steps, {(t1 - t0) / 1e6 / nst:.3f} ms per step, {len(sel) / nst:.0f} launches per step")
ev = []
for s, e, q, k in sel: ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth = 0; last = t0; hist = collections.Counter()
for t, d in ev:
    hist[min(depth, 3)] += t - last; last = t; depth += d
hist[0] += t1 - last
for d in range(4):
    print(f"  {d}{'+' if d == 3 else ' '} kernels in flight: {hist[d] / 1e6 / nst:7.3f} ms per step ({100 * hist[d] / (t1 - t0):5.1f} %)")
busy = collections.Counter()
for s, e, q, k in sel: busy[q] += e - s
for q, v in busy.most_common(): print(f"  queue {q}: {v / 1e6 / nst:.3f} ms of kernels per step")
# holes: intervals with nothing in flight
holes = []; cur_end = t0; prev = None
for s, e, q, k in sel:
    if s > cur_end: holes.append((s - cur_end, cur_end, prev, k))
    if e > cur_end: cur_end = e; prev = k
holes.sort(reverse=True)
agg = collections.Counter(); cnt = collections.Counter()
for d, at, a, b in holes: agg[(a, b)] += d; cnt[(a, b)] += 1
print(f"holes: {len(holes) / nst:.0f} per step, {sum(h[0] for h in holes) / 1e6 / nst:.3f} ms per step; by (kernel before -> kernel after):")
for (a, b), v in agg.most_common(nholes):
    print(f"  {v / 1e3 / nst:7.1f} us/step  {cnt[(a, b)] / nst:5.1f} x {v / cnt[(a, b)] / 1e3:5.1f} us   {a} -> {b}")
