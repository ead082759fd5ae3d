import csv,glob,re,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    n=r["Name"]; m=re.search(r'(k_\w+(<\d>)?)',n); print("  %-20s calls %4s avg %9.1f us %s%%"%(m.group(1) if m else n[:20],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"][:5]))
