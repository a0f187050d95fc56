import csv, collections, sys
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:90]) for r in csv.DictReader(open(sys.argv[1]))]
t_end=max(r[1] for r in rows)
def win(a,b):
    c=collections.Counter(); d=collections.Counter()
    for s,e,n in rows:
        if t_end-b*1e9 <= s < t_end-a*1e9: c[n]+=1; d[n]+=e-s
    return c,d
gA,gAd=win(0.7,1.5)   # graph replay region (0.8 s)
eA,eAd=win(0.0,0.21)  # eager tail (3 steps at ~70 ms)
# normalise by a kernel that runs exactly 3x per train step (the fused warp of the three netG levels)
key=[n for n in set(gA)|set(eA) if "warp_fwd_kernel<float, true>" in n][0]
gn=gA[key]/3.0; en=eA[key]/3.0
print("graph steps %.1f eager steps %.1f"%(gn,en))
names=set(gA)|set(eA)
tab=[(n,gA[n]/gn,eA[n]/en,gAd[n]/gn/1e6,eAd[n]/en/1e6) for n in names]
tab.sort(key=lambda t:-(t[3]-t[4]))
print("%-70s %8s %8s %8s %8s"%("kernel","g/step","e/step","g ms","e ms"))
for t in tab[:22]: print("%-70s %8.1f %8.1f %8.2f %8.2f"%t)
print("total per step: graph %d calls %.1f ms ; eager %d calls %.1f ms"%(sum(gA.values())/gn,sum(gAd.values())/gn/1e6,sum(eA.values())/en,sum(eAd.values())/en/1e6))
