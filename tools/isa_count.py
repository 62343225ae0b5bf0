#!/usr/bin/env python3
"""Static instruction mix per kernel of a device assembly listing (hipcc -S --cuda-device-only):
   python tools/isa_count.py nkml.s [substring]   ->  vector ALU, scalar ALU, LDS, vector memory, branch counts and register use"""
import re, sys, collections
sub = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
cnt = collections.OrderedDict()
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1); cnt[cur] = collections.Counter(); continue
    if cur is None: continue
    if line.startswith(".Lfunc_end"): cur = None; continue
    t = line.strip().split()
    if not t or t[0].startswith((".", ";")) or t[0].endswith(":"): continue
    op = t[0]
    c = cnt[cur]
    if op.startswith("ds_"): c["lds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1
    elif op.startswith("v_"): c["valu"] += 1
    elif op.startswith(("s_cbranch", "s_branch")): c["branch"] += 1
    elif op.startswith(("s_waitcnt", "s_barrier", "s_nop")): c["wait"] += 1
    elif op.startswith("s_"): c["salu"] += 1
    m = re.search(r"\.(vgpr_count|sgpr_count|vgpr_spill_count|lds_size)", line)
print("%-70s %6s %6s %6s %6s %6s" % ("kernel", "valu", "salu", "lds", "vmem", "branch"))
for k, c in cnt.items():
    if sub in k and sum(c.values()) > 0:
        print("%-70s %6d %6d %6d %6d %6d" % (k[:70], c["valu"], c["salu"], c["lds"], c["vmem"], c["branch"]))
