# Sum rocprofv3 counter_collection.csv files per kernel and counter (mean per dispatch).
# usage: python tools/pmc_sum.py <dir>...
import csv, glob, os, re, sys, collections
def kname(n):
    m = re.search(r"(k_\w+(<[^>]*>)?)", n)
    return m.group(1) if m else n[:40]
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            per[(row["Dispatch_Id"], kname(row["Kernel_Name"]), row["Counter_Name"])] += float(row["Counter_Value"])
        for (disp, k, c), v in per.items():
            acc[k][c].append(v)
    print("==", d)
    for k in sorted(acc):
        print("  ", k, "dispatches", max(len(v) for v in acc[k].values()))
        print("     " + "  ".join("%s %.4gM" % (c.replace("SQ_INSTS_", ""), sum(v) / len(v) / 1e6) for c, v in sorted(acc[k].items())))
