"""Per-step kernel table from a rocprofv3 --kernel-trace of scripts/profile_step.py: name, launches per step, mean us,
ms per step, share.  usage: step_kernels.py <trace dir> <steps in the trace> [top]"""
import csv, re, collections, glob, sys
path = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
n = int(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
tot = collections.Counter(); cnt = collections.Counter()
def short(nm):
    nm = re.sub(r'\(anonymous namespace\)::', '', nm)
    nm = re.sub(r'^void ', '', nm)
    nm = re.sub(r'\(.*$', '', nm)
    return nm[:86]
for r in csv.DictReader(open(path)):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    k = short(r['Kernel_Name']); tot[k] += d; cnt[k] += 1
T = sum(tot.values())
print(f"{'kernel':86s} {'n/step':>7s} {'us':>7s} {'ms/step':>8s} {'%':>5s}")
for k, v in tot.most_common(top):
    print(f"{k:86s} {cnt[k]/n:7.1f} {v/cnt[k]:7.1f} {v/1e3/n:8.3f} {100*v/T:5.1f}")
print(f"total kernel time {T/1e3/n:.2f} ms/step, {sum(cnt.values())/n:.0f} launches/step, {len(tot)} kernels")
