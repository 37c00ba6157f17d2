"""For every launch of kernels matching PATTERN in a rocprofv3 kernel trace: the kernel before and after it on the same queue.
    python scripts/trace_neighbours.py <kernel_trace.csv> PATTERN [skip_fraction]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
rows = rows[int(len(rows) * skip):]
byq = collections.defaultdict(list)
for r in rows: byq[r["Queue_Id"]].append(r)
def short(n): return re.sub(r"\(anonymous namespace\)::|void |_ZN12_GLOBAL__N_1\d+", "", n)[:60]
cnt = collections.Counter()
for q, lst in byq.items():
    for i, r in enumerate(lst):
        if pat.search(r["Kernel_Name"]):
            prev = short(lst[i - 1]["Kernel_Name"]) if i else "-"
            nxt = short(lst[i + 1]["Kernel_Name"]) if i + 1 < len(lst) else "-"
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            cnt[(q, prev, nxt, r.get("Grid_Size", ""))] += 1
for (q, prev, nxt, g), n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(f"{n:5d}  queue {q} grid {g}:  {prev}  ->  [match]  ->  {nxt}")
