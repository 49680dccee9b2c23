"""Summarise rocprofv3 --pmc output (counter_collection.csv files under a directory) for the
propagate kernel: mean per dispatch of every counter.  Usage: pmc_pass.py DIR [DIR...]"""
import csv, glob, json, sys, collections
out = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for row in csv.DictReader(open(f)):
            if "propagate_kernel" not in row["Kernel_Name"]:
                continue
            acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for name, per in acc.items():
            v = list(per.values())
            out[name] = {"dispatches": len(v), "mean_per_dispatch": sum(v) / len(v)}
print(json.dumps(out, indent=1))
