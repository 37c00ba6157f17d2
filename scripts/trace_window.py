"""Print the kernel sequence (all queues) around the n-th launch of a kernel whose name contains <pattern>:
python scripts/trace_window.py <kernel_trace.csv> <pattern> [n=-2] [before=12] [after=14]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]; n = int(sys.argv[3]) if len(sys.argv) > 3 else -2
before = int(sys.argv[4]) if len(sys.argv) > 4 else 12; after = int(sys.argv[5]) if len(sys.argv) > 5 else 14
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r['Kernel_Name']) for r in rows))
idx = [i for i, k in enumerate(ks) if pat in k[3]]
c = idx[n]
t0 = ks[c][0]
prev_end = {}
for i in range(max(0, c - before), min(len(ks), c + after)):
    s, e, q, name = ks[i]
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else float('nan')
    prev_end[q] = e
    name = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', name)[:60]
    print(f"{'>>' if i == c else '  '} t={(s - t0) / 1e3:9.1f} us  dur={(e - s) / 1e3:7.1f}  q={q}  gap_same_queue={gap:7.1f}  {name}")
