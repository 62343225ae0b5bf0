#!/usr/bin/env python3
"""Write the small edge-case inputs under tests/golden/inputs/ (deterministic, seed 7).

These are this build's own synthetic records, shaped to hit the corners of the reference's
per-read path: pairs dropped because a mate is shorter than k, reads of exactly k bases, runs of N
(masked to A, giving zero-code windows that are neither counted nor totalled), poly-A / poly-T /
poly-G reads (the same k-mer many times inside one read, so the depth threshold is crossed in the
middle of a read), reverse-complement mates (canonical mode), tandem repeats, clones that push
k-mers over the depth, and headers that already end in /1 or /2 (FASTQ->FASTA header rule).
tests/golden/make_golden.py then runs the compiled reference on them to get the expected outputs.
"""
import os
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "inputs")

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def rc(s):
    return "".join(COMP[c] for c in reversed(s))


def main():
    rnd = random.Random(7)

    def rseq(n):
        return "".join(rnd.choice("ACGT") for _ in range(n))

    def mutate(s, pos, to=None):
        c = s[pos]
        alt = to or next(x for x in "ACGT" if x != c)
        return s[:pos] + alt + s[pos + 1:]

    base_f, base_r = rseq(60), rseq(72)
    long_f, long_r = rseq(300), rseq(251)
    pal = "ACGTACGTACGTACGTACGTACGTACGTACGT"  # reverse-complement palindrome
    recs = []  # (header_fwd, fwd, header_rev, rev)

    def add(f, r, hf=None, hr=None):
        i = len(recs) + 1
        recs.append((hf or "@EDGE:%d 1:N:0:ACGT" % i, f, hr or "@EDGE:%d 2:N:0:ACGT" % i, r))

    add(base_f, base_r)
    add(base_f, base_r)
    add(base_f, base_r)                                   # third copy: depth-3 k-mers become high here
    add(mutate(base_f, 30), mutate(base_r, 5))
    add(rseq(10), rseq(50))                               # fwd shorter than 15
    add(rseq(50), rseq(14))                               # rev shorter than 15
    add(rseq(15), rseq(16))                               # exactly k / k+1 at k=15
    add("A" * 40, rseq(44))                               # no countable window in fwd
    add("ACGTTGCA" + "N" * 22 + "GATTACAGATTACAGATTACA", "N" * 15 + rseq(30))
    add("T" * 40, "G" * 40)                               # poly-T is code 0 only in canonical mode
    add("G" * 33, "G" * 40)
    add("G" * 21, "C" * 40)                               # poly-C is the rc of poly-G
    add(rc(base_f), rc(base_r))                           # rc mates: same canonical k-mers as record 1
    add(pal, pal[2:] + "AC")
    add("AC" * 35, "ACG" * 24)                            # tandem repeats: few distinct k-mers, many times
    add("AC" * 35, "ACG" * 24)
    add(long_f, long_r)
    add(long_f, long_r)
    add(mutate(long_f, 150), long_r)
    add(long_f[:120] + base_f, long_r)
    add(base_f, base_r, hf="@EDGE:hdr/1", hr="@EDGE:hdr/2")
    add(base_f[5:], base_r[7:], hf="@EDGE:swapped/2", hr="@EDGE:swapped/1")
    add(base_f[:40] + "N" + base_f[41:], "N" + base_r[1:])
    for _ in range(12):
        add(rseq(rnd.randint(31, 140)), rseq(rnd.randint(31, 140)))
    add(long_f[40:200], long_r[10:170])
    add(base_f, base_r)                                   # last record must be valid for every k used

    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "edge_1.fastq"), "w") as f1, open(os.path.join(OUT, "edge_2.fastq"), "w") as f2, \
            open(os.path.join(OUT, "edge_1.fasta"), "w") as a1, open(os.path.join(OUT, "edge_2.fasta"), "w") as a2:
        for hf, f, hr, r in recs:
            f1.write("%s\n%s\n+\n%s\n" % (hf, f, "F" * len(f)))
            f2.write("%s\n%s\n+\n%s\n" % (hr, r, "F" * len(r)))
            a1.write(">%s\n%s\n" % (hf[1:], f))
            a2.write(">%s\n%s\n" % (hr[1:], r))
    print("wrote", len(recs), "records to", OUT)


if __name__ == "__main__":
    main()
