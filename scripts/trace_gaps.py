"""Per-queue occupancy of a rocprofv3 kernel_trace.csv: busy time, idle gaps between consecutive kernels of a queue,
over the last N steps' worth of time.  python scripts/trace_gaps.py <kernel_trace.csv> <nsteps>"""
import csv, sys, collections
path, nsteps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
byq = collections.defaultdict(list)
for r in rows:
    byq[r.get('Queue_Id', '0')].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    ks.sort()
    # keep the steady-state tail: the last 70 % of the launches
    ks = ks[int(len(ks) * 0.3):]
    busy = sum(e - s for s, e, _ in ks) / 1e3
    span = (ks[-1][1] - ks[0][0]) / 1e3
    gaps = [(ks[i][0] - ks[i - 1][1]) / 1e3 for i in range(1, len(ks))]
    pos = [g for g in gaps if g > 0]
    small = [g for g in pos if g < 50]
    hist = collections.Counter(min(int(g), 20) for g in small)
    print(f"queue {q}: {len(ks)} launches, span {span / 1e3:.2f} ms, busy {busy / 1e3:.2f} ms ({busy / span:.1%}), "
          f"gaps<50us: n={len(small)} sum={sum(small) / 1e3:.2f} ms mean={sum(small) / max(len(small), 1):.2f} us; "
          f"overlapping launches: {sum(1 for g in gaps if g <= 0)}")
    print("   gap histogram (us: count):", " ".join(f"{k}:{hist[k]}" for k in sorted(hist)))
    import re
    short = lambda n: re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n)[:44]
    pairs = collections.Counter()
    pg = collections.defaultdict(float)
    for i in range(1, len(ks)):
        g = (ks[i][0] - ks[i - 1][1]) / 1e3
        if 3 < g < 50:
            k = (short(ks[i - 1][2]), short(ks[i][2]))
            pairs[k] += 1; pg[k] += g
    for k, n in pairs.most_common(14):
        print(f"   {n:4d} gaps, mean {pg[k] / n:5.1f} us: {k[0]} -> {k[1]}")
