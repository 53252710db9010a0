import csv, collections, sys
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.OrderedDict()
for r in rows:
    if (int(r['cls'])>2) != (len(sys.argv)>2): continue
    k=(int(r['cls']),int(r['M']),int(r['N']),int(r['K']),int(r['ksize']))
    a=agg.setdefault(k,[0,0.0,0.0]); a[0]+=1; a[1]+=float(r['us']); a[2]+=float(r['flops'])
tot=sum(v[1] for v in agg.values())
print("total igemm us", tot)
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:40]:
    print("cls %d M=%6d N=%5d K=%6d ks=%d  n=%3d  tot_us %8.1f (%4.1f%%) avg_us %7.1f  %6.1f TF" % (*k, v[0], v[1], 100*v[1]/tot, v[1]/v[0], v[2]/v[1]/1e6))
