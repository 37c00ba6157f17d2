import csv,re,collections,glob,sys
path=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]; n=int(sys.argv[2])
rows=list(csv.DictReader(open(path)))
cat=collections.Counter(); cnt=collections.Counter()
def c(nm):
    if 'gemm_kernel' in nm:
        return 'gemm wgrad' if 'Lb1ELb1' in nm else 'gemm dgrad' if 'Lb0ELb1' in nm else 'gemm fwd'
    for k in ('splitk_reduce','attn_bwd','attn_fwd','ln_bwd','ln_fwd','ln_param','adamw','colsum','rows_transform','embed','ce_','cast','im2col','unary'):
        if k in nm: return k
    return 'torch/other'
for r in rows:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    k=c(r['Kernel_Name']); cat[k]+=d; cnt[k]+=1
tot=sum(cat.values())
for k,v in cat.most_common(): print(f"{k:16s} {v/1e3/n:7.2f} ms/step  {100*v/tot:5.1f}%  launches/step={cnt[k]/n:6.1f}")
print("total", round(tot/1e3/n,2), "launches/step", len(rows)/n)
