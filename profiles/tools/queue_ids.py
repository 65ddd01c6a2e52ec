import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
c=collections.Counter((r['Queue_Id'], r['Stream_Id'], 'rccl' if 'rccl' in r['Kernel_Name'] else ('fcpt' if 'fcpt' in r['Kernel_Name'] else 'other')) for r in rows)
print(sorted(c.items()))
