"""Kernel launches per step in the steady state of a rocprofv3 kernel trace: counts kernels whose start lies in the last
`frac` of the trace span and divides by the number of adamw launches there / 4.  python scripts/trace_tail_counts.py <csv>"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r['Start_Timestamp']) for r in rows); t1 = max(int(r['End_Timestamp']) for r in rows)
# steps are delimited by adamw launches: take the window between the first adamw of step -6 and the last adamw
ad = sorted(int(r['Start_Timestamp']) for r in rows if 'adamw' in r['Kernel_Name'])
per = 4
nst = 5
lo, hi = ad[-per * nst - 1], ad[-1]
cnt = collections.Counter(); dur = collections.Counter()
for r in rows:
    s = int(r['Start_Timestamp'])
    if lo < s <= hi:
        n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', r['Kernel_Name'])[:50]
        cnt[n] += 1; dur[n] += (int(r['End_Timestamp']) - s) / 1e3
print(f"window {(hi - lo) / 1e6:.2f} ms = {nst} steps -> {(hi - lo) / 1e6 / nst:.2f} ms/step, {sum(cnt.values()) / nst:.0f} launches/step")
for n, c in cnt.most_common(40):
    print(f"{c / nst:7.1f}/step {dur[n] / nst / 1e3:7.3f} ms/step  {n}")
