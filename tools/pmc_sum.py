"""Sum rocprofv3 counter_collection.csv files per kernel: python3 tools/pmc_sum.py <dir> [name substrings ...]
Prints one line per (pass, kernel): launches and counter totals (kernel names shortened to their template head)."""
import csv, glob, re, sys
root = sys.argv[1]
want = sys.argv[2:] or ["mm_"]
csv.field_size_limit(1 << 30)
for f in sorted(glob.glob(root + "/*/*counter_collection.csv")):
    acc = {}
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if not any(w in name for w in want):
            continue
        mt = re.search(r"(mm_\w+(<[^>(]*>)?)", name)
        k = mt.group(1) if mt else name[:60]
        a = acc.setdefault(k, {"_disp": set()})
        a["_disp"].add(r["Dispatch_Id"])
        a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for k, a in acc.items():
        n = len(a.pop("_disp"))
        print(f.split("/")[-2], k, "launches", n, {c: f"{v:.5g}" for c, v in a.items()})
