"""Synthetic paired-end RNA-seq-like reads for benchmarks and parity tests (no dataset can be downloaded).

Generator spec (the one the reference numbers of this build were measured on): T "transcripts" of length
U[600,3000) of uniform random ACGT; abundance ~ lognormal(0, 2) * (len - 300); fragment length U[200,500)
clipped to the transcript; read 1 = first 150 bases of the fragment, read 2 = reverse complement of its last
150; 0.5 % substitution errors per base; 0.01 % of bases become N.  numpy's default_rng(seed) is consumed in a
fixed order, so (n, seed, transcripts) names one exact dataset: n=2_000_000, seed=1, transcripts=2000 gives,
at k=15 depth=100 coverage=0.9, printed 1 011 077 / skipped 988 923 / 28 393 181 distinct 15-mers with the
reference run collision-free (-m 16)."""
import numpy as np

READ_LEN = 150
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def pair_blocks(n, seed=1, transcripts=2000, block=100000):
    """Yields (r1, r2): uint8 ASCII arrays of shape [m, 150], m <= block, n pairs in total."""
    rng = np.random.default_rng(seed)
    T = transcripts
    lens = rng.integers(600, 3000, T)
    offs = np.concatenate([[0], np.cumsum(lens)])
    G = rng.integers(0, 4, offs[-1]).astype(np.uint8)
    ab = rng.lognormal(0, 2.0, T) * (lens - 300)
    ab /= ab.sum()
    L = READ_LEN
    ar = np.arange(L)
    for s in range(0, n, block):
        m = min(block, n - s)
        t = rng.choice(T, m, p=ab)
        fl = np.minimum(rng.integers(200, 500, m), lens[t])
        st = offs[t] + (rng.random(m) * (lens[t] - fl + 1)).astype(np.int64)
        r1 = G[st[:, None] + ar[None, :]]
        r2 = 3 - G[(st + fl - 1)[:, None] - ar[None, :]]
        for r in (r1, r2):
            e = rng.random(r.shape) < 0.005
            r[e] = (r[e] + rng.integers(1, 4, e.sum())) % 4
        c1 = _ACGT[r1]
        c2 = _ACGT[r2]
        nm = rng.random(c1.shape) < 0.0001
        c1[nm] = ord("N")
        nm = rng.random(c2.shape) < 0.0001
        c2[nm] = ord("N")
        yield c1, c2


def interleave(c1, c2):
    """[m,150] x2 -> (seq uint8[m*300], offsets uint64[2m+1]) with mates adjacent, forward first."""
    m, L = c1.shape
    seq = np.empty((m, 2, L), dtype=np.uint8)
    seq[:, 0, :] = c1
    seq[:, 1, :] = c2
    offsets = np.arange(2 * m + 1, dtype=np.uint64) * np.uint64(L)
    return seq.reshape(-1), offsets


def write_fastq(prefix, n, seed=1, transcripts=2000):
    """PREFIX_1.fastq / PREFIX_2.fastq with headers @SYN:<seed>:<i> 1:N:0:ACGT, '+' line bare, quality F*150."""
    qual = b"F" * READ_LEN
    i = 0
    with open(prefix + "_1.fastq", "wb") as f1, open(prefix + "_2.fastq", "wb") as f2:
        for c1, c2 in pair_blocks(n, seed, transcripts):
            b1, b2 = [], []
            for j in range(c1.shape[0]):
                h = b"@SYN:%d:%d " % (seed, i)
                b1.append(h + b"1:N:0:ACGT\n" + c1[j].tobytes() + b"\n+\n" + qual + b"\n")
                b2.append(h + b"2:N:0:ACGT\n" + c2[j].tobytes() + b"\n+\n" + qual + b"\n")
                i += 1
            f1.write(b"".join(b1))
            f2.write(b"".join(b2))
