"""bench.py's live `roofline.avg_launch_us` against the kernel trace of the SAME run:
    rocprofv3 --kernel-trace -d <dir> -o b --output-format csv -- python3 bench.py --steps 20 --warmup 10 > bench.json
    python scripts/roofline_vs_trace.py <dir>/b_kernel_trace.csv bench.json 10 20
The timed region of the headline variant is found from the optimizer launches (steps warmup+1 .. warmup+steps)."""
import csv, json, re, sys
trace, line, warm, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "adamw_kernel" in r["Kernel_Name"]]
groups, cur = [], [ad[0]]
for a in ad[1:]:
    if a[0] - cur[-1][1] < 2_000_000: cur.append(a)
    else: groups.append(cur); cur = [a]
groups.append(cur)
t0, t1 = groups[warm - 1][-1][1], groups[warm + steps - 1][-1][1]
d = json.loads(open(line).read().strip().splitlines()[-1])
FAM = r"gemm_kernel|gemm_glds_kernel|rowstream_kernel|gemm8_kernelILi.ELi.ELb0|gemm8_kernel<\d, \d, false"
for key, pat in (("roofline", FAM), ("roofline_wgrad_group", r"gemm_group_kernel|gemm_group_glds_kernel")):
    k = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows
         if t0 <= int(r["Start_Timestamp"]) < t1 and re.search(pat, r["Kernel_Name"])]
    tr = sum(k) / len(k)
    live = d[key]["avg_launch_us"]
    print(f"{key}: trace {tr:.2f} us over {len(k) / steps:.0f} launches/step; bench.py live {live:.2f} us ({(live / tr - 1) * 100:+.1f} %); "
          f"achieved {d[key]['achieved']} {d[key]['unit']}, frac {d[key]['frac']}")
print(f"timed region: {(t1 - t0) / steps / 1e6:.3f} ms/step in the trace, bench.py {d['ms_per_step']} ms/step (under the profiler)")
