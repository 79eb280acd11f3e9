"""Mean per launch of every PMC counter for the hta:: kernels in rocprofv3 counter_collection.csv files."""
import csv, sys, collections
for path in sys.argv[1:]:
    acc = collections.defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "")
            if "hta::" not in k:
                continue
            acc[(k.split("(")[0].replace("void ", ""), row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print("%-60s %-32s launches=%d mean=%.1f" % (k[:60], c, len(v), sum(v) / len(v)))
