"""Per-kernel averages of the counters of a rocprofv3 --pmc output tree (csv): last 8 dispatches of each kernel."""
import csv, glob, sys, collections
root = sys.argv[1]
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.replace(root, ""))
    for k, cs in rows.items():
        print("  ", k)
        for c, v in cs.items():
            tail = v[-8:]
            print("      %-24s %14.0f  (n=%d)" % (c, sum(tail) / len(tail), len(v)))
