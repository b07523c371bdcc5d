import csv, glob, collections, sys
path=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True)[0]
vols=float(sys.argv[2]); skip=float(sys.argv[3])
rows=list(csv.DictReader(open(path)))
t0=min(int(r["Start_Timestamp"]) for r in rows); t1=max(int(r["End_Timestamp"]) for r in rows)
lo=t0+(t1-t0)*skip
acc=collections.Counter(); cnt=collections.Counter()
import re
for r in rows:
    if int(r["Start_Timestamp"])>=lo:
        n=re.sub(r"\(.*","",r["Kernel_Name"])[:70]; acc[n]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"]); cnt[n]+=1
tot=sum(acc.values())
print("window %.3f s, sum of kernel durations %.3f s; per volume (over %.1f volumes in the window): %.1f ms" % ((t1-lo)/1e9, tot/1e9, vols*(1-skip), tot/1e6/(vols*(1-skip))))
for n,v in acc.most_common(16): print("%-72s %6d calls %8.2f ms per volume  avg %.1f us" % (n, cnt[n], v/1e6/(vols*(1-skip)), v/1e3/cnt[n]))
