"""Per-kernel mean of rocprofv3 --pmc counters from the csv output (counter_collection.csv) as JSON, stamped with the digest
of the kernel sources the numbers were measured on (bench.py refuses a summary whose stamp does not match the tree)."""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


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
    from bench import kernel_source_hash
    res = {"kernel_source_hash": kernel_source_hash(), "unit": "KB per launch (FETCH_SIZE as reported: double it on gfx950)"}
    for d in sys.argv[1:]:
        res.update(summarise(d))
    print(json.dumps(res, indent=1))
