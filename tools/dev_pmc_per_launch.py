# per-launch counter values of the fused matcher from one rocprofv3 --pmc run (counter_collection.csv)
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "k_knn_bvh_post" not in r["Kernel_Name"]: continue
    per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(per)[-50:]
names = sorted(per[ids[0]])
print("launch", *names)
for n, i in enumerate(ids):
    print(n, *[int(per[i][c]) for c in names])
