"""In-situ duration of a kernel class split by WHAT RAN BESIDE IT (round 6).

    python scripts/in_situ_overlap.py <kernel_trace.csv> <pattern> [skip_first_ms=0]

For every launch whose name contains <pattern>: its duration and the kernels of OTHER queues in flight during it (share of
the launch's duration they overlap).  Launches are grouped by (kernel, grid) and by the dominant neighbour (none / the
neighbour's name) and the mean / median duration is printed per group: a class that runs at stand-alone speed when alone and
several times slower beside one particular neighbour is losing to co-residency, not to its own code.
(`rocprofv3 --pmc` serialises dispatches, so counters cannot show this; a plain `--kernel-trace` keeps the concurrency.)"""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]
skip = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 0.0
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"],
             int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 0) or 0)) for r in rows)
t_first = ks[0][0]
short = lambda n: re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)[:58]
groups = collections.defaultdict(list)
import bisect
starts = [k[0] for k in ks]
for i, (s, e, q, name, grid, wg) in enumerate(ks):
    if pat not in name or s - t_first < skip:
        continue
    d = e - s
    # neighbours: any kernel of another queue overlapping [s, e)
    best, share = "alone", 0.0
    j = bisect.bisect_left(starts, s - 400_000)          # nothing on this path runs longer than 0.4 ms
    tot = 0.0
    while j < len(ks) and ks[j][0] < e:
        s2, e2, q2, n2 = ks[j][0], ks[j][1], ks[j][2], ks[j][3]
        if q2 != q and e2 > s:
            ov = (min(e, e2) - max(s, s2)) / max(d, 1)
            tot += ov
            if ov > share:
                best, share = short(n2), ov
        j += 1
    key = (short(name), grid // max(wg, 1))
    groups[key].append((d / 1e3, best if share >= 0.3 else "alone", tot))
for key in sorted(groups):
    v = groups[key]
    print(f"== {key[0]} blocks={key[1]}: {len(v)} launches, mean {sum(x[0] for x in v) / len(v):.1f} us")
    by = collections.defaultdict(list)
    for d, nb, tot in v:
        by[nb].append(d)
    for nb, ds in sorted(by.items(), key=lambda kv: -len(kv[1])):
        ds.sort()
        print(f"     beside {nb:60s} n={len(ds):4d}  mean {sum(ds) / len(ds):7.1f}  median {ds[len(ds) // 2]:7.1f}  min {ds[0]:6.1f}  max {ds[-1]:7.1f} us")
