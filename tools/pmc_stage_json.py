#!/usr/bin/env python3
"""HBM bytes of the counting stage per batch from two rocprofv3 --pmc passes of bench.py (tools/round_profiles.sh), as a
record for profiles/pmc_counting_stage.json:

    python tools/pmc_stage_json.py <tag>_fetch_pmc.txt <tag>_write_pmc.txt <tag>_bench.json [existing.json] > new.json

Bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes) summed over the stage's kernels (k_pt_* / k_hp_*), each kernel's
per-call average times its calls per batch.  FETCH_SIZE is doubled because gfx950 tallies the 128-byte requests of wide
coalesced reads at 64 bytes; the stage's big readers are such reads (k_pt_hist: 1.09 GB reported for 2.18 GB of keys).
"""
import json
import re
import sys


def table(path):
    out = {}
    for line in open(path):
        m = re.match(r"(\S.*?)\s+(\d+)\s+(\d+)\s*$", line)
        if m and not line.startswith("kernel"):
            out[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return out


def main():
    fetch, write = table(sys.argv[1]), table(sys.argv[2])
    bench = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    stage = lambda k: k.startswith("k_pt_") or k.startswith("k_hp_")
    steps = max(c for k, (c, _) in fetch.items() if "scatter_words" in k)
    f = sum(c * v for k, (c, v) in fetch.items() if stage(k)) / steps * 1024
    w = sum(c * v for k, (c, v) in write.items() if stage(k)) / steps * 1024
    cfg, roof = bench["config"], bench["roofline"]
    occ = roof["occurrences_per_launch"]
    k = cfg.get("ksize") or int(re.search(r"k=(\d+)", bench["metric"]).group(1))
    table_bytes = bench.get("table", {}).get("table_bytes")
    if not table_bytes and k > 16:  # older bench lines: from the bucket kernel's design bytes (2 x table + 9 per occurrence)
        table_bytes = (roof["bucket_kernel"]["design_bytes_per_launch"] - 9.0 * occ) / 2
    if k <= 15:
        # per k-mer position: packed words + start bits in twice (0.75), pass A 4 out + 1.5 (12-bit sorted positions), histogram
        # B 4 in, pass B 4 in + 2 out + 1.5, bucket 2 in + 1 out, reverse passes 2 x (1 + 1.5 in) + 1 + 0.125 out; table twice
        design = 26.9 * occ + 2.0 * 4 ** k
    else:
        # the same pipeline with 8-byte hashes: 0.75 + 8 + 1.5 + 8 + 8 + 8 + 1.5 + 8 + 1 + 6.1; 8-byte slots in and out
        design = 50.9 * occ + 2.0 * (table_bytes or 0)
    rec = {"ksize": k, "pairs_per_step": cfg["pairs_per_step"], "partitioned": bool(cfg.get("partitioned", True)),
           "canonical": bool(cfg.get("canonical", k > 16)), "hbm_bytes_per_launch": 2 * f + w, "fetch_size_bytes_reported": f,
           "write_size_bytes": w, "design_bytes_per_launch": design, "steps_profiled": steps,
           "source": [sys.argv[1].split("/")[-1], sys.argv[2].split("/")[-1]]}
    recs = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else []
    recs = [r for r in recs if (r["ksize"], r["pairs_per_step"], r["partitioned"], r["canonical"]) !=
            (rec["ksize"], rec["pairs_per_step"], rec["partitioned"], rec["canonical"])] + [rec]
    json.dump(recs, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
