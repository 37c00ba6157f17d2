"""Where do the runtime's copy / fill kernels sit in a step?  python scripts/trace_copies.py <kernel_trace.csv> -- for the last full
step of a rocprofv3 --kernel-trace of scripts/profile_step.py: every __amd_rocclr_* / at::native launch with the kernel before
and after it on the same queue."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r['Kernel_Name']) for r in rows))
def short(n):
    n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n); n = re.sub(r'void |\(anonymous namespace\)::', '', n)
    return n[:70]
cuts = [i for i in range(len(ks) - 1) if 'adamw' in ks[i][3] and 'adamw' not in ks[i + 1][3]]
a, b = cuts[-2] + 1, cuts[-1] + 1
step = ks[a:b]
t0 = step[0][0]
byq = {}
for k in step: byq.setdefault(k[2], []).append(k)
for q, lst in byq.items():
    for i, (s, e, _, n) in enumerate(lst):
        if 'rocclr' in n or 'at::native' in n:
            pv = short(lst[i - 1][3]) if i else '-'; nx = short(lst[i + 1][3]) if i + 1 < len(lst) else '-'
            print(f"t={(s - t0) / 1e3:8.1f} us q={q} dur={(e - s) / 1e3:5.1f}  {short(n)[:44]:44s} after [{pv[:40]}] before [{nx[:40]}]")
