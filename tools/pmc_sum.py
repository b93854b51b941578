"""Average per-launch value of a PMC counter for the kernels whose name contains a substring, from the
*_counter_collection.csv of `rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv`.
usage: python tools/pmc_sum.py dir_or_csv COUNTER substring [substring...]"""
import csv
import glob
import os
import sys


def main():
    src, counter, subs = sys.argv[1], sys.argv[2], sys.argv[3:]
    files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
    acc = {s: [0.0, set()] for s in subs}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            for s in subs:
                if s in r["Kernel_Name"]:
                    acc[s][0] += float(r["Counter_Value"])
                    acc[s][1].add(r["Dispatch_Id"])
    for s in subs:
        tot, ids = acc[s]
        print("%s %s launches %d per_launch %.3f" % (counter, s, len(ids), tot / max(len(ids), 1)))


if __name__ == "__main__":
    main()
