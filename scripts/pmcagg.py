import csv, collections, sys
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    k=r["Kernel_Name"].split("(")[0]
    agg[(k,r["Counter_Name"])][0]+=1; agg[(k,r["Counter_Name"])][1]+=float(r["Counter_Value"])
for (k,c),(n,v) in sorted(agg.items()):
    if k.startswith(sys.argv[2]): print(k,c,n,v/n)
