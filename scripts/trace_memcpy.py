"""Memory copies (SDMA / blit) inside a traced step next to the kernels around them:
python scripts/trace_memcpy.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv> -- prints every copy of the last full step
(between the last two optimizer sweeps) with its direction, size, duration, and the kernel that ended before / started after it."""
import csv, glob, sys, re, bisect
d = sys.argv[1]
kt = list(csv.DictReader(open(glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0])))
mc = list(csv.DictReader(open(glob.glob(d + '/**/*_memory_copy_trace.csv', recursive=True)[0])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in kt)
def short(n):
    n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n); n = re.sub(r'void |\(anonymous namespace\)::', '', n); return n[:48]
cuts = [i for i in range(len(ks) - 1) if 'adamw' in ks[i][2] and 'adamw' not in ks[i + 1][2]]
t0, t1 = ks[cuts[-2]][1], ks[cuts[-1]][1]
starts = [k[0] for k in ks]
print(f"step {(t1 - t0) / 1e6:.3f} ms; columns: t (us from the end of the previous sweep), direction, bytes, duration, kernel before -> kernel after")
for r in sorted(mc, key=lambda r: int(r['Start_Timestamp'])):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s < t0 or s > t1: continue
    i = bisect.bisect_right(starts, s)
    before = max((k for k in ks[max(0, i - 40):i] if k[1] <= s), key=lambda k: k[1], default=None)
    after = ks[i] if i < len(ks) else None
    print(f"t={(s - t0) / 1e3:9.1f} {r.get('Direction', '?'):24s} {r.get('Bytes', r.get('Size', '?')):>10s} B dur={(e - s) / 1e3:6.1f} us  "
          f"[{short(before[2]) if before else '-'} ended {((s - before[1]) / 1e3) if before else 0:.1f} us earlier] -> [{short(after[2]) if after else '-'} starts {((after[0] - e) / 1e3) if after else 0:.1f} us later]")
