"""Randomised soak of the drop-in command line against the compiled reference program (a checker run): random small
FASTQ/FASTA inputs - paired and single, several files, mates shorter than k, N runs, shared and unrelated reads - through
oracle/_ref/nkml_ref --cpu 1 and through nkml_amd with the same options, k <= 13 (where the reference's table is
collision-free by construction); the output files must be the same byte for byte, and the report numbers equal.
python tests/soak_cli.py [cases] [seed]   (not collected by pytest: a long run for the GPU box)"""
import hashlib, os, re, shutil, subprocess, sys, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "nkml_ref")
BIN = os.path.join(ROOT, "nomalise_kmers_multi_large_amd", "nkml_amd")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def read(pool, lo, hi):
    n = int(rng.integers(lo, hi))
    if rng.random() < 0.7:
        at = int(rng.integers(0, len(pool) - n))
        r = pool[at:at + n].copy()
    else:
        r = ACGT[rng.integers(0, 4, n)]
    if rng.random() < 0.1:
        a = int(rng.integers(0, n)); r[a:a + int(rng.integers(1, 12))] = ord("N")
    return r.tobytes()


def write(path, reads, fasta, tag):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            if fasta:
                f.write(b">r%d%s\n%s\n" % (i, tag, r))
            else:
                f.write(b"@r%d%s\n%s\n+\n%s\n" % (i, tag, r, b"F" * len(r)))


def outputs(d):
    res = {}
    for fn in sorted(os.listdir(d)):
        if fn.startswith("output_"):
            res[fn] = hashlib.md5(open(os.path.join(d, fn), "rb").read()).hexdigest()
    return res


def report(out):
    keep = {}
    for ln in out.decode("latin-1").splitlines():
        for tag in ("Processed Records: ", "Printed Records: ", "Skipped Records: "):
            if ln.startswith(tag):
                keep[tag] = ln[len(tag):].replace(",", "").strip()
    return keep


bad = crashed = 0
for case in range(cases):
    k = int(rng.integers(5, 14))
    depth = int(rng.choice([2, 3, 5, 20]))
    cov = str(rng.choice([0.5, 0.75, 0.9, 0.96]))
    fasta = bool(rng.random() < 0.25)
    single = bool(rng.random() < 0.3)
    n_files = int(rng.choice([1, 1, 2]))
    pool = ACGT[rng.integers(0, 4, int(rng.choice([600, 5000])))]
    top = tempfile.mkdtemp(prefix="nkml_soak_")
    try:
        ext = "fasta" if fasta else "fastq"
        fwd, rev = [], []
        for i in range(n_files):
            n = int(rng.integers(1, 1500))
            lo = int(rng.choice([k, max(5, k - 4)]))   # sometimes mates shorter than k
            a = [read(pool, lo, 260) for _ in range(n)]
            write(os.path.join(top, "f%d.%s" % (i, ext)), a, fasta, b"/1")
            fwd.append("../f%d.%s" % (i, ext))
            if not single:
                b = [read(pool, lo, 260) for _ in range(n)]
                write(os.path.join(top, "r%d.%s" % (i, ext)), b, fasta, b"/2")
                rev.append("../r%d.%s" % (i, ext))
        args = ["-f"] + fwd + (["--single"] if single else ["-r"] + rev) + ["-k", str(k), "-d", str(depth), "-g", cov]
        if fasta:
            args += ["-t", "fa", "-o", "fa"]
        elif rng.random() < 0.3:
            args += ["--outformat", "fa"]
        if rng.random() < 0.4:
            args += ["-c"]
        res = []
        for name, exe in (("ref", REF), ("amd", BIN)):
            d = os.path.join(top, name)
            os.mkdir(d)
            extra = ["--cpu", "1"] if name == "ref" else []
            r = subprocess.run([exe] + args + extra, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            res.append((r.returncode, outputs(d), report(r.stdout)))
        if res[0][0] < 0:   # the reference itself died of a signal (it does on some multi-file inputs): nothing to compare with
            crashed += 1
            print("reference killed by signal %d in case %d: %s" % (-res[0][0], case, " ".join(args)), flush=True)
        elif res[0] != res[1]:
            bad += 1
            print("MISMATCH case %d: %s\n  ref %s\n  amd %s" % (case, " ".join(args), res[0], res[1]), flush=True)
    finally:
        shutil.rmtree(top, ignore_errors=True)
    if case % 10 == 9:
        print("%d cases done, %d mismatches" % (case + 1, bad), flush=True)
print("cli soak: %d cases, %d mismatches, %d without a reference result (reference crashed)" % (cases, bad, crashed))
sys.exit(1 if bad else 0)
