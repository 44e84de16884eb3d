"""Per-kernel mean of one rocprofv3 --pmc counter from the csv output (counter_collection.csv) as JSON."""
import collections
import csv
import glob
import json
import re
import sys


def summarise(directory):
    out = {}
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            m = re.search(r"((?:lmpc|nlmpc)_\w+)", r["Kernel_Name"])
            if m:
                acc[r["Counter_Name"]][m.group(1)].append(float(r["Counter_Value"]))
        for cname, per in acc.items():
            out[cname] = {k: {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in per.items()}
    return out


if __name__ == "__main__":
    res = {}
    for d in sys.argv[1:]:
        res.update(summarise(d))
    print(json.dumps(res, indent=1))
