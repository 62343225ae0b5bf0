#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 --pmc csv: python tools/pmc_any.py <counter_collection.csv> [substr]"""
import collections, csv, re, sys

d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"\b(k_[a-z_0-9]+(?:<[a-z ]+>)?)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:40]
    d[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
ctrs = sorted({c for k in d for c in d[k]})
print("%-28s %6s " % ("kernel", "calls") + " ".join("%18s" % c[:18] for c in ctrs))
for k in sorted(d):
    if sub not in k:
        continue
    n = max(len(v) for v in d[k].values())
    print("%-28s %6d " % (k[:28], n) + " ".join("%18.0f" % (sum(d[k][c]) / max(1, len(d[k][c]))) for c in ctrs))
