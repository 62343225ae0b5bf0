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


# ---- the same kinds of data made on the GPU (torch), for runs whose input would take minutes to make with numpy ----
# Test and bench plumbing: torch is only the random number generator and the gather here.

def gpu_pair_steps(n_pairs, pairs_per_step, seed=1, transcripts=None, device="cuda"):
    """The generator of pair_blocks() on the device: same distribution (transcript lengths, lognormal abundances,
    fragment lengths, 0.5 % substitutions, 0.01 % N), torch's generator instead of numpy's, so the reads differ from
    pair_blocks' for the same seed.  Yields (seq uint8[2*m*150] device tensor, m) per step, mates interleaved, forward
    first; all steps come from one transcriptome, so k-mer counts build up across steps as in one file."""
    import torch
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    T = int(transcripts) if transcripts else max(50, n_pairs // 1000)
    lens = torch.randint(600, 3000, (T,), generator=g, device=dev)
    offs = torch.cumsum(lens, 0) - lens
    G = torch.randint(0, 4, (int(lens.sum().item()),), generator=g, device=dev, dtype=torch.uint8)
    ab = torch.exp(2.0 * torch.randn(T, generator=g, device=dev, dtype=torch.float64)) * (lens - 300).double()
    ab = ab / ab.sum()
    ascii_t = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    ar = torch.arange(READ_LEN, device=dev)
    piece = 500000  # pairs made at a time (bounds the temporaries: about 150 B x 30 per pair)
    for s in range(0, n_pairs, pairs_per_step):
        m = min(pairs_per_step, n_pairs - s)
        out = torch.empty((m, 2, READ_LEN), dtype=torch.uint8, device=dev)
        for p0 in range(0, m, piece):
            q = min(piece, m - p0)
            t = torch.multinomial(ab, q, replacement=True, generator=g)
            lt = lens[t]
            fl = torch.minimum(torch.randint(200, 500, (q,), generator=g, device=dev), lt)
            st = offs[t] + (torch.rand(q, generator=g, device=dev, dtype=torch.float64) * (lt - fl + 1).double()).long()
            r1 = G[st[:, None] + ar[None, :]]
            r2 = 3 - G[(st + fl - 1)[:, None] - ar[None, :]]
            both = torch.stack([r1, r2], dim=1)
            u = torch.rand(both.shape, generator=g, device=dev)
            shift = torch.randint(1, 4, both.shape, generator=g, device=dev, dtype=torch.uint8)
            both = torch.where(u < 0.005, (both + shift) % 4, both)
            chars = ascii_t[both.long()]
            chars[u > 0.9999] = ord("N")
            out[p0:p0 + q] = chars
        yield out.reshape(-1), m


class GpuGenomeReads:
    """Paired reads that cover a random genome evenly (what makes every k-mer of a bucket reach the depth in the same
    few batches, and a k > 16 table grow to hundreds of millions of k-mers): fragments of 300-399 bases at uniform
    positions of a `genome_len`-base genome, 0.1 % substitutions, 0.05 % N.  block(i, n) is a pure function of
    (seed, i), so two runs over the same block numbers see the same stream."""

    def __init__(self, genome_len=50000000, seed=12345, device="cuda"):
        import torch
        self.torch = torch
        self.dev = torch.device(device)
        self.g = torch.Generator(device=self.dev)
        self.g.manual_seed(int(seed))
        self.genome = torch.randint(0, 4, (genome_len,), generator=self.g, device=self.dev, dtype=torch.uint8)
        self.ascii = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=self.dev)
        self.ar = torch.arange(READ_LEN, device=self.dev)

    def block(self, i, n):
        """(seq uint8[n*300]) of block i: n pairs, mates interleaved, forward first."""
        torch, g, dev = self.torch, self.g, self.dev
        g.manual_seed(1000 + int(i))
        pos = torch.randint(0, self.genome.numel() - 400, (n,), generator=g, device=dev)
        frag = torch.randint(300, 400, (n,), generator=g, device=dev)
        fwd = self.genome[pos[:, None] + self.ar]
        rev = 3 - self.genome[(pos + frag - 1)[:, None] - self.ar]
        both = torch.stack([fwd, rev], dim=1).reshape(2 * n, READ_LEN)
        r = torch.rand(both.shape, generator=g, device=dev)
        sub = torch.randint(0, 4, both.shape, generator=g, device=dev, dtype=torch.uint8)
        both = torch.where(r < 0.001, sub, both)
        seq = self.ascii[both.long()]
        seq[r > 0.9995] = ord("N")
        return seq.reshape(-1)
