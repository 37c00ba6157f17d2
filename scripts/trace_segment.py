"""Kernel sequence (all queues) of the last full step of a scripts/profile_step.py trace between t0 and t1 microseconds after the
step's first kernel:  python scripts/trace_segment.py <kernel_trace.csv> <t0_us> <t1_us>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
t0u, t1u = float(sys.argv[2]), float(sys.argv[3])
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r['Kernel_Name'], int(r.get('Grid_Size', 0) or 0), int(r.get('Workgroup_Size', 0) or 1)) for r in rows))
def short(n):
    n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n); n = re.sub(r'void |\(anonymous namespace\)::', '', n)
    return n[:84]
cuts = [i for i in range(len(ks) - 1) if 'adamw' in ks[i][3] and 'adamw' not in ks[i + 1][3]]
a, b = cuts[-2] + 1, cuts[-1] + 1
step = ks[a:b]
t0 = step[0][0]
prev = {}
busy = 0.0
for s, e, q, n, g, w in step:
    ts = (s - t0) / 1e3
    if t0u <= ts <= t1u:
        gap = (s - prev[q]) / 1e3 if q in prev else float('nan')
        print(f"t={ts:8.1f} q={q} dur={(e - s) / 1e3:6.1f} gap={gap:6.1f} blocks={g // max(w, 1):6d}  {short(n)}")
    prev[q] = e
