#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc csv output.
   python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [kernel-substring]"""
import collections
import csv
import json
import re
import sys


def load(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        m = re.search(r"\b(k_[a-z_0-9]+)", r["Kernel_Name"])
        name = m.group(1) if m else r["Kernel_Name"][:40]
        d[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


def main():
    f, w = load(sys.argv[1]), load(sys.argv[2])
    out = {}
    print("%-22s %6s %16s %16s %16s" % ("kernel", "calls", "FETCH_SIZE_KB/call", "WRITE_SIZE_KB/call", "EA0_ATOMIC/call"))
    for k in sorted(f):
        fv, wv, av = f[k].get("FETCH_SIZE", []), w[k].get("WRITE_SIZE", []), w[k].get("TCC_EA0_ATOMIC_sum", [])
        avg = lambda v: sum(v) / len(v) if v else 0.0
        print("%-22s %6d %16.1f %16.1f %16.0f" % (k, len(fv), avg(fv), avg(wv), avg(av)))
        out[k] = {"calls": len(fv), "fetch_kb": avg(fv), "write_kb": avg(wv), "atomics": avg(av)}
    return out


if __name__ == "__main__":
    main()
