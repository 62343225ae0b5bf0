"""Pins the CPU oracle (oracle/nkml_oracle.c) to the reference: the known-answer comment in the reference
source (normalise_kmers_multi_large.c:56-70) and the vectors under tests/golden/ that tests/golden/make_golden.py
captured from the compiled reference on the reference's own test files and on this build's edge inputs."""
import numpy as np
import pytest

from tests.helpers import Oracle, drive_oracle, golden_arrays, golden_meta, parse_args, records, input_path

META = golden_meta()

# Cases in which the reference's table is collision-free (4^k <= capacity, or so few distinct k-mers that
# no two share a start slot), so its counts are exact stream-order counts.
COUNT_EXACT = {"G1_2seq_single_d2", "G5x_a1a2_b1b2_d4_m16", "G9_a1b1_k11_d3_c", "G10_a2_single_k13_d5",
               "E1_edge_pe_k15_d3", "E2_edge_pe_k15_d3_c", "E3_edge_se_k9_d2_fa", "E4_edge_fasta_k21_d2_c",
               "E5_edge_pe_k15_d3_fa"}
FIELDS = ("high_fwd", "high_rev", "total_fwd", "total_rev")


def test_known_answer_comment(oracle_lib):
    # normalise_kmers_multi_large.c:56-70: four 87 bp reads at --depth 2, k=15
    recs = records(input_path("REF_TEST/2seq.fastq"))
    o = oracle_lib(15, 2)
    got = [o.read(r[1])[1:] for r in recs]
    assert got == [(0, 73), (73, 73), (70, 73), (58, 73)]
    assert o.unique == 91
    # float32 ratios printed there: 0.00 1.00 0.96 0.79 ; with the default coverage 0.9 -> keep, skip, skip, keep
    assert [oracle_lib.keep_single(h, t, np.float32(0.9)) for h, t in got] == [True, False, False, True]


def test_encode_matches_hand_values(oracle_lib):
    assert oracle_lib.encode(b"A" * 15) == 0
    assert oracle_lib.encode(b"ACGT") == 0b00011011
    assert oracle_lib.encode(b"T" * 31) == (1 << 62) - 1
    assert oracle_lib.encode(b"CA") == 4  # first base in the high bits
    # canonical = min(code, code of reverse complement)
    assert oracle_lib.encode(b"TTTT", canonical=True) == 0
    assert oracle_lib.encode(b"ACGT", canonical=True) == oracle_lib.encode(b"ACGT")  # palindrome
    assert oracle_lib.encode(b"GGGA", canonical=True) == oracle_lib.encode(b"GGGA")  # rc = TCCC is larger
    assert oracle_lib.encode(b"TCCC", canonical=True) == oracle_lib.encode(b"GGGA")


def test_rolling_window_codes_equal_per_window_encoding(oracle_lib):
    rng = np.random.default_rng(3)
    for k in (5, 15, 16, 17, 25, 31):
        for canonical in (False, True):
            seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 200))
            o = oracle_lib(k, 2, canonical)
            rc, high, total = o.read(seq)
            codes = [oracle_lib.encode(seq[i:i + k], canonical) for i in range(len(seq) - k + 1)]
            nz = [c for c in codes if c]
            assert rc == 1 and total == len(nz)
            assert o.unique == len(set(nz))
            for c in set(nz):
                assert o.count_of(c) == nz.count(c)


def test_keep_skip_edges(oracle_lib):
    cov = np.float32(0.9)
    assert oracle_lib.keep_single(0, 0, cov)            # total 0 -> ratio 0 -> keep
    assert not oracle_lib.keep_single(9, 10, cov)       # (float)9/10 == 0.9f -> not < -> skip
    assert oracle_lib.keep_single(8, 10, cov)
    assert not oracle_lib.keep_pair(0, 10, 10, 10, cov)  # either mate over the coverage -> skip
    assert oracle_lib.keep_pair(8, 10, 0, 0, cov)
    assert not oracle_lib.keep_single(24, 25, np.float32(0.96))  # 0.96f/0.96f
    assert oracle_lib.keep_single(95, 100, np.float32(0.96))


def test_short_and_invalid_reads(oracle_lib):
    o = oracle_lib(15, 2)
    assert o.read(b"ACGTACGTACGTAC")[0] == 0            # 14 < k: dropped, nothing counted
    assert o.unique == 0
    assert o.read(b"ACGTACGTACGTACG")[0] == 1
    assert o.read(b"ACGTACGTACGTACGx")[0] == -1         # reference exits on a non-ACGT base
    assert o.read(b"acgtacgtacgtacgt")[0] == -1         # lower case is fatal too
    assert o.pair(b"ACGTACGTACGTACGT", b"ACGT")[0] == 0
    assert o.seed(b"ACGTACGTACGTACG") == 0              # seeding needs len > k
    rc, h, t = o.read(b"N" * 20 + b"C")                 # N -> A: 6 all-A windows are skipped, one ends in C
    assert (rc, h, t) == (1, 0, 1)


@pytest.mark.parametrize("name", sorted(META))
def test_exact_model_against_reference(oracle_lib, name):
    """Decisions of the exact-count model equal the reference's on every captured case; where the
    reference's table had no collisions its per-record counters and unique-k-mer total match as well."""
    cfg = parse_args(META[name]["args"])
    gold = golden_arrays(name)
    got = drive_oracle(cfg, model=Oracle.EXACT)
    assert len(got["kept"]) == META[name]["records"]
    np.testing.assert_array_equal(got["kept"], gold["kept"])
    np.testing.assert_array_equal(got["total_fwd"], gold["total_fwd"])
    np.testing.assert_array_equal(got["total_rev"], gold["total_rev"])
    final = META[name]["final"]
    assert int(got["kept"].sum()) == final["printed"]
    assert len(got["kept"]) - int(got["kept"].sum()) == final["skipped"]
    if name in COUNT_EXACT:
        for f in FIELDS:
            np.testing.assert_array_equal(got[f], gold[f], err_msg=f)
        assert got["unique"] == final["unique"]
        assert [p[:3] for p in got["per_file"]] == [p[:3] for p in META[name]["per_file"]]


@pytest.mark.parametrize("name", sorted(n for n in META if n not in ("G5x_a1a2_b1b2_d4_m16",)))
def test_quirk_model_against_reference(oracle_lib, name):
    """The literal table state machine (with the seeding pass) reproduces the default-table reference run
    counter for counter, including its 'unique kmers' figure."""
    cfg = parse_args(META[name]["args"])
    gold = golden_arrays(name)
    got = drive_oracle(cfg, model=Oracle.QUIRK)
    np.testing.assert_array_equal(got["kept"], gold["kept"])
    for f in FIELDS:
        np.testing.assert_array_equal(got[f], gold[f], err_msg=f)
    assert got["unique"] == META[name]["final"]["unique"]
    assert got["per_file"] == META[name]["per_file"]
