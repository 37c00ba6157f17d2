"""Summarise a rocprofv3 kernel_trace.csv: per (kernel, grid) time per step."""
import csv, re, sys, collections
path, nsteps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = list(csv.DictReader(open(path)))
agg = collections.defaultdict(lambda: [0.0, 0])
tot = 0.0
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    name = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', r['Kernel_Name'])
    name = re.sub(r'\(anonymous namespace\)::', '', name)[:60]
    grid = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    agg[(name, grid)][0] += d; agg[(name, grid)][1] += 1
    tot += d
print(f"total {tot/1e3/nsteps:.2f} ms/step over {len(rows)/nsteps:.0f} launches/step")
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
for (name, grid), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{t/1e3/nsteps:7.3f} ms/step  n/step={n/nsteps:6.1f}  avg={t/n:8.1f}us  grid={grid}  {name}")
