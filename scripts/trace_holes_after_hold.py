"""Gaps > G us with NO kernel in flight in the first step behind the hold of a traced `HOLD_MS=.. scripts/profile_step.py`
(everything was enqueued before the hold ended, so these are not host latency): python scripts/trace_holes_after_hold.py <csv> [G=8]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1]))); G = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r['Kernel_Name']) for r in rows)
def short(n):
    n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n); n = re.sub(r'void |\(anonymous namespace\)::', '', n); return n[:56]
hold_end = max(e for s, e, q, n in ks if 'add' in n.lower() and e - s > 200_000)
after = [k for k in ks if k[0] >= hold_end]
ad = [i for i, k in enumerate(after) if 'adamw' in k[3]]
first_end = next(i for i in ad if i + 1 < len(after) and 'adamw' not in after[i + 1][3])
step = after[:first_end + 1]
print(f"first step behind the hold: {(step[-1][1] - hold_end) / 1e6:.3f} ms, {len(step)} launches")
cur_end = hold_end; tot = 0.0
for i, (s, e, q, n) in enumerate(step):
    if s - cur_end > G * 1e3:
        prev = max(step[:i], key=lambda k: k[1]) if i else None
        print(f"t={(s - hold_end) / 1e3:9.1f} us  hole {(s - cur_end) / 1e3:6.1f} us  after [{short(prev[3]) if prev else 'hold'} q{prev[2] if prev else ''}] before [{short(n)} q{q}]")
    if s > cur_end: tot += s - cur_end
    cur_end = max(cur_end, e)
print(f"idle (no kernel in flight): {tot / 1e3:.1f} us")
