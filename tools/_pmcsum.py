import csv, glob, sys, collections
tag, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(f"gpurun_out/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
    disp = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            disp[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in disp.items():
        acc[c][0] += v; acc[c][1] += 1
for c in sorted(acc):
    print(f"  {c:32s} mean_per_dispatch={acc[c][0]/acc[c][1]:.4g}  (n={acc[c][1]})")
