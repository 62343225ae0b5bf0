#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the COMPILED REFERENCE program.

Runs only in the build container (needs /root/reference and oracle/_ref/nkml_ref, which
`make -C oracle ref` builds from the reference's own source file).  The outputs are data only:
per-record (kept, high_fwd, high_rev, total_fwd, total_rev) parsed from the reference's
`--debug 2` lines (normalise_kmers_multi_large.c:1682-1692), its final report numbers and the
md5 of the files it wrote.  No reference source text is stored.

usage: python tests/golden/make_golden.py [case ...]
"""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "nkml_ref")
REF_TEST = "/root/reference/test"
GOLD = os.path.join(ROOT, "tests", "golden")
EDGE = os.path.join(GOLD, "inputs")

T = REF_TEST + "/"
E = EDGE + "/"

# name -> argv after the binary.  The ids G1..G8 follow the table of reference runs this build
# was specified against; E* are edge-case inputs written by tools/make_edge_inputs.py.
CASES = {
    "G1_2seq_single_d2": ["-f", T + "2seq.fastq", "--single", "--depth", "2"],
    "G2_a1b1_k15_d100": ["-f", T + "a1.fastq", "-r", T + "b1.fastq", "-k", "15", "-d", "100", "-g", "0.9"],
    "G3_a2b2_default": ["-f", T + "a2.fastq", "-r", T + "b2.fastq"],
    "G4_a1a2_b1b2_default": ["-f", T + "a1.fastq", T + "a2.fastq", "-r", T + "b1.fastq", T + "b2.fastq"],
    "G5_a1a2_b1b2_d4": ["-f", T + "a1.fastq", T + "a2.fastq", "-r", T + "b1.fastq", T + "b2.fastq", "--depth", "4"],
    "G5x_a1a2_b1b2_d4_m16": ["-f", T + "a1.fastq", T + "a2.fastq", "-r", T + "b1.fastq", T + "b2.fastq",
                             "--depth", "4", "-m", "16"],
    "G6_a1a2_b1b2_d4_k25_c": ["-f", T + "a1.fastq", T + "a2.fastq", "-r", T + "b1.fastq", T + "b2.fastq",
                              "--depth", "4", "-k", "25", "-c"],
    "G7_mixed_k31_d50_fa": ["-f", T + "a1.fastq", T + "a2.fastq", "-r", T + "b1.fastq", "--single", "-k", "31",
                            "--depth", "50", "--coverage", "0.96", "--outformat", "fa"],
    "G8_a1a2_b1b2_k31_d50": ["-f", T + "a1.fastq", T + "a2.fastq", "-r", T + "b1.fastq", T + "b2.fastq", "-k", "31",
                             "-d", "50", "-g", "0.96"],
    "G9_a1b1_k11_d3_c": ["-f", T + "a1.fastq", "-r", T + "b1.fastq", "-k", "11", "-d", "3", "-c", "-g", "0.5"],
    "G10_a2_single_k13_d5": ["-f", T + "a2.fastq", "--single", "-k", "13", "-d", "5", "-g", "0.75"],
    "E1_edge_pe_k15_d3": ["-f", E + "edge_1.fastq", "-r", E + "edge_2.fastq", "-k", "15", "-d", "3", "-g", "0.9"],
    "E2_edge_pe_k15_d3_c": ["-f", E + "edge_1.fastq", "-r", E + "edge_2.fastq", "-k", "15", "-d", "3", "-g", "0.9", "-c"],
    "E3_edge_se_k9_d2_fa": ["-f", E + "edge_1.fastq", "--single", "-k", "9", "-d", "2", "-g", "0.5", "--outformat", "fa"],
    "E4_edge_fasta_k21_d2_c": ["-f", E + "edge_1.fasta", "-r", E + "edge_2.fasta", "-t", "fa", "-o", "fa", "-k", "21",
                               "-d", "2", "-c"],
    "E5_edge_pe_k15_d3_fa": ["-f", E + "edge_1.fastq", "-r", E + "edge_2.fastq", "-k", "15", "-d", "3", "-g", "0.9",
                             "--outformat", "fa"],
}

LINE = re.compile(
    r"^Thread 0 - Sequence pair ([\d,]+) (PRINTED|SKIPPED): High \((\d+)\) count kmers: F:(\d+)(?:;R:(\d+))?, "
    r"Total kmers: F:(\d+)(?:;R:(\d+))?")
FILESTAT = re.compile(r"^Cumulative file statistics: Processed ([\d,]+), Printed ([\d,]+), Skipped ([\d,]+), "
                      r"Cumulative Max Unique Kmers in a thread: ([\d,]+)")


def num(s):
    return int(s.replace(",", ""))


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def run_case(name, args):
    tmp = tempfile.mkdtemp(prefix="nkml_gold_")
    try:
        cmd = [REF_BIN] + args + ["--debug", "2"]
        res = subprocess.run(cmd, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=False)
        out = res.stdout.decode("latin-1").splitlines()
        kept, hf, hr, tf, tr = [], [], [], [], []
        per_file = []
        final = {}
        for ln in out:
            m = LINE.match(ln)
            if m:
                assert num(m.group(1)) == len(kept) + 1, (name, ln)
                kept.append(1 if m.group(2) == "PRINTED" else 0)
                hf.append(int(m.group(4)))
                hr.append(int(m.group(5)) if m.group(5) is not None else -1)
                tf.append(int(m.group(6)))
                tr.append(int(m.group(7)) if m.group(7) is not None else -1)
                continue
            m = FILESTAT.match(ln)
            if m:
                per_file.append([num(m.group(i)) for i in range(1, 5)])
                continue
            for key, tag in (("processed", "Processed Records: "), ("printed", "Printed Records: "),
                             ("skipped", "Skipped Records: "),
                             ("unique", "Cumulative Max unique kmers in any thread: ")):
                if ln.startswith(tag):
                    final[key] = num(ln[len(tag):])
        files = {}
        for fn in sorted(os.listdir(tmp)):
            p = os.path.join(tmp, fn)
            files[fn] = {"md5": md5(p), "bytes": os.path.getsize(p)}
        np.savez_compressed(os.path.join(GOLD, name + ".npz"),
                            kept=np.array(kept, dtype=np.uint8),
                            high_fwd=np.array(hf, dtype=np.int32), high_rev=np.array(hr, dtype=np.int32),
                            total_fwd=np.array(tf, dtype=np.int32), total_rev=np.array(tr, dtype=np.int32))
        return {"args": [a.replace(T, "REF_TEST/").replace(E, "EDGE/") for a in args],
                "returncode": res.returncode, "final": final, "per_file": per_file, "files": files,
                "records": len(kept)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    want = sys.argv[1:] or list(CASES)
    meta_path = os.path.join(GOLD, "golden.json")
    meta = json.load(open(meta_path)) if os.path.exists(meta_path) else {}
    for name in want:
        print("running", name, flush=True)
        meta[name] = run_case(name, CASES[name])
        print("   ", meta[name]["final"], meta[name]["records"], flush=True)
    with open(meta_path, "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
