"""Per-kernel time per step of two rocprofv3 kernel traces of scripts/profile_step.py, side by side (A/B of a switch):
    python scripts/cmp_trace.py a_kernel_trace.csv b_kernel_trace.csv
Steps are cut at the optimizer launches; steps 6..14 of each trace are averaged."""
import csv, re, sys, collections
def load(path):
    rows=list(csv.DictReader(open(path)))
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    ad=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"])) for r in rows if "adamw_kernel" in r["Kernel_Name"]]
    groups=[]; cur=[ad[0]]
    for a in ad[1:]:
        if a[0]-cur[-1][1] < 2_000_000: cur.append(a)
        else: groups.append(cur); cur=[a]
    groups.append(cur)
    t0=groups[5][-1][1]; t1=groups[14][-1][1]; n=9
    acc=collections.defaultdict(lambda:[0,0.0])
    for r in rows:
        if t0<=int(r["Start_Timestamp"])<t1:
            nm=re.sub(r"_ZN12_GLOBAL__N_1\d+|void |\(anonymous namespace\)::","",r["Kernel_Name"])
            nm=re.sub(r"IDF16b","<bf16>",nm)[:46]
            a=acc[nm]; a[0]+=1; a[1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    return {k:(v[0]/n, v[1]/n) for k,v in acc.items()}, (t1-t0)/n/1e6
a,wa=load(sys.argv[1]); b,wb=load(sys.argv[2])
print("wall ms/step", wa, wb)
keys=sorted(set(a)|set(b), key=lambda k:-(a.get(k,(0,0))[1]+b.get(k,(0,0))[1]))
ta=tb=0
for k in keys[:40]:
    ca,ua=a.get(k,(0,0)); cb,ub=b.get(k,(0,0))
    print(f"{k:46s} {ca:6.1f} {ua:8.1f} us | {cb:6.1f} {ub:8.1f} us | {ub-ua:+8.1f}")
print("total", sum(v[1] for v in a.values()), sum(v[1] for v in b.values()))
