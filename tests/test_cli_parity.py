"""The command-line host (nkml_amd) against the compiled reference's recorded runs: byte-identical output
files (md5), identical per-record debug counters and identical report numbers."""
import gzip
import hashlib
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from tests.helpers import INPUTS, ROOT, Oracle, drive_oracle, golden_arrays, golden_meta, parse_args

META = golden_meta()
BIN = os.path.join(ROOT, "nomalise_kmers_multi_large_amd", "nkml_amd")
COUNT_EXACT = {"G1_2seq_single_d2", "G5x_a1a2_b1b2_d4_m16", "G9_a1b1_k11_d3_c", "G10_a2_single_k13_d5",
               "E1_edge_pe_k15_d3", "E2_edge_pe_k15_d3_c", "E3_edge_se_k9_d2_fa", "E4_edge_fasta_k21_d2_c",
               "E5_edge_pe_k15_d3_fa"}
LINE = re.compile(r"^Thread 0 - Sequence pair ([\d,]+) (PRINTED|SKIPPED): High \((\d+)\) count kmers: F:(\d+)(?:;R:(\d+))?, "
                  r"Total kmers: F:(\d+)(?:;R:(\d+))?")


@pytest.fixture(scope="session")
def inputs_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("inputs")
    for fn in os.listdir(INPUTS):
        src = os.path.join(INPUTS, fn)
        if fn.endswith(".gz"):
            with gzip.open(src, "rb") as f, open(os.path.join(d, fn[:-3]), "wb") as g:
                shutil.copyfileobj(f, g)
        else:
            shutil.copy(src, os.path.join(d, fn))
    return str(d)


def run_cli(args, cwd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([BIN] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=300)


def md5(path):
    with open(path, "rb") as f:
        return hashlib.md5(f.read()).hexdigest()


def num(s):
    return int(s.replace(",", ""))


# ---- argument handling runs before the GPU is touched: checked on any machine -------------------------------
def test_version_and_usage_without_gpu(tmp_path):
    assert os.path.exists(BIN), "build the host first: make -C nomalise_kmers_multi_large_amd/host"
    r = run_cli(["--version"], str(tmp_path))
    assert r.returncode == 0 and r.stdout.strip() == b"20240823"
    r = run_cli(["-h"], str(tmp_path))
    assert r.returncode == 0 and b"--forward|-f" in r.stderr


@pytest.mark.parametrize("args,needle", [
    ([], b"no fwd (0) or reverse (0) files provided"),
    (["-f", "X.fq", "-r", "X.fq", "-k", "4"], b"Only kmer sizes (4) of 5 to 31"),
    (["-f", "X.fq", "-r", "X.fq", "-k", "32"], b"Only kmer sizes (32) of 5 to 31"),
    (["-f", "X.fq", "-r", "X.fq", "-d", "1"], b"Depth (1)"),
    (["-f", "X.fq", "-r", "X.fq", "-g", "1.5"], b"Coverage"),
    (["-f", "X.fq", "-r", "X.fq", "-p", "0"], b"CPU count (0)"),
    (["-f", "X.fq", "X.fq", "-r", "X.fq"], b"Number of forward (2) and reverse files (1) must match"),
    (["-f", "X.fq", "-r", "X.fq", "-t", "fa", "-o", "fq"], b"cannot request an output format of FASTQ when input is FASTA"),
    (["-f", "missing.fq", "-r", "X.fq"], b"Warning: File 'missing.fq' does not exist"),
])
def test_argument_validation_without_gpu(tmp_path, args, needle):
    (tmp_path / "X.fq").write_text("@r\nACGT\n+\nFFFF\n")
    r = run_cli(args, str(tmp_path))
    assert r.returncode == 1
    assert needle in r.stderr + r.stdout


# ---- full runs ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(META))
def test_cli_against_reference_runs(name, inputs_dir, tmp_path, oracle_lib):
    check_cli_run(name, inputs_dir, tmp_path, {})


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(META))
def test_cli_on_two_table_shards_against_reference_runs(name, inputs_dir, tmp_path, oracle_lib):
    """The same runs with the k-mer table sharded over two devices from the C host (NKML_DEVICES): on a one-GPU machine the
    two shards share the GPU and device-to-device copies stand in for the links; outputs, report and per-record counters
    must not change."""
    check_cli_run(name, inputs_dir, tmp_path, {"NKML_DEVICES": "0,0", "NKML_TRANSPORT": "copy"})


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["G7_mixed_k31_d50_fa", "G5_a1a2_b1b2_d4"])
def test_cli_on_four_table_shards(name, inputs_dir, tmp_path, oracle_lib):
    """The mixed single-end + paired-end k=31 --outformat fa run (four devices in the headline configuration) and a k=15 run
    on four shards."""
    check_cli_run(name, inputs_dir, tmp_path, {"NKML_DEVICES": "0,0,0,0", "NKML_TRANSPORT": "copy"})


def check_cli_run(name, inputs_dir, tmp_path, extra_env):
    meta = META[name]
    args = [a.replace("REF_TEST/", inputs_dir + "/").replace("EDGE/", inputs_dir + "/") for a in meta["args"]]
    env = {"NKML_BATCH_READS": "6000", "NKML_BATCH_BASES": "600000"}
    env.update(extra_env)
    r = run_cli(args + ["--debug", "2"], str(tmp_path), env=env)
    assert r.returncode == 0, r.stderr.decode()
    out = r.stdout.decode("latin-1").splitlines()
    # output files: same names, same bytes
    assert sorted(os.listdir(tmp_path)) == sorted(meta["files"])
    for fn, want in meta["files"].items():
        p = os.path.join(tmp_path, fn)
        assert os.path.getsize(p) == want["bytes"], fn
        assert md5(p) == want["md5"], fn
    # report block
    final = {}
    for ln in out:
        for key, tag in (("processed", "Processed Records: "), ("printed", "Printed Records: "), ("skipped", "Skipped Records: "),
                         ("unique", "Cumulative Max unique kmers in any thread: ")):
            if ln.startswith(tag):
                final[key] = num(ln[len(tag):])
    for key in ("processed", "printed", "skipped"):
        assert final[key] == meta["final"][key]
    per_file = [[num(x) for x in re.findall(r"([\d,]+)", ln.split(":", 1)[1])] for ln in out if ln.startswith("Cumulative file statistics")]
    assert [p[:3] for p in per_file] == [p[:3] for p in meta["per_file"]]
    # per-record counters from the debug lines
    rows = [LINE.match(ln) for ln in out]
    rows = [m for m in rows if m]
    got = dict(kept=np.array([m.group(2) == "PRINTED" for m in rows], dtype=np.uint8),
               high_fwd=np.array([int(m.group(4)) for m in rows]), total_fwd=np.array([int(m.group(6)) for m in rows]),
               high_rev=np.array([int(m.group(5)) if m.group(5) else -1 for m in rows]),
               total_rev=np.array([int(m.group(7)) if m.group(7) else -1 for m in rows]))
    gold = golden_arrays(name)
    np.testing.assert_array_equal(got["kept"], gold["kept"])
    want = drive_oracle(parse_args(meta["args"]), model=Oracle.EXACT)
    for f in ("high_fwd", "high_rev", "total_fwd", "total_rev"):
        np.testing.assert_array_equal(got[f], want[f], err_msg=f)
        if name in COUNT_EXACT:
            np.testing.assert_array_equal(got[f], gold[f], err_msg=f)
    assert final["unique"] == want["unique"]
    if name in COUNT_EXACT:
        assert final["unique"] == meta["final"]["unique"]
        assert per_file == meta["per_file"]


@pytest.mark.gpu
def test_cli_fatal_on_invalid_base(inputs_dir, tmp_path):
    good = "@a\n" + "ACGT" * 10 + "\n+\n" + "F" * 40 + "\n"
    bad = "@b\n" + "ACGT" * 5 + "acgt" + "ACGT" * 4 + "\n+\n" + "F" * 40 + "\n"
    (tmp_path / "f.fq").write_text(good + bad + good)
    (tmp_path / "r.fq").write_text(good + good + good)
    # like the reference, the seeding pass (first 3 M records of every file) meets the bad read before any
    # output file exists
    r = run_cli(["-f", "f.fq", "-r", "r.fq", "-d", "2"], str(tmp_path))
    assert r.returncode == 1
    assert b"FATAL: FWD sequence does not appear to be a DNA sequence" in r.stderr
    assert not (tmp_path / "output_forward.k15_norm2_thread0.fastq").exists()
    # past the seeded prefix the record loop meets it: the records before it were already decided and written,
    # like the reference's stdio buffers flushed by exit()
    r = run_cli(["-f", "f.fq", "-r", "r.fq", "-d", "100"], str(tmp_path), env={"NKML_NO_SEED": "1"})
    assert r.returncode == 1
    assert b"FATAL: FWD sequence does not appear to be a DNA sequence" in r.stderr
    assert (tmp_path / "output_forward.k15_norm100_thread0.fastq").read_text() == good


@pytest.mark.gpu
def test_cli_print_table(inputs_dir, tmp_path, oracle_lib):
    r = run_cli(["-f", inputs_dir + "/2seq.fastq", "--single", "--depth", "2", "-P"], str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()
    rows = [ln.split("\t") for ln in (tmp_path / "output_kmer.k15_norm2_thread0.tsv").read_text().splitlines()]
    seeds = [ln.split("\t") for ln in (tmp_path / "output_kmer_seeds.k15_norm2.tsv").read_text().splitlines()]
    assert len(rows) == 91 and len(seeds) == 91 and all(c == "0" for _, c in seeds)
    o = oracle_lib(15, 2)
    from tests.helpers import records, input_path
    for rec in records(input_path("REF_TEST/2seq.fastq")):
        o.read(rec[1])
    assert all(o.count_of(Oracle.encode(k.encode())) == int(c) for k, c in rows)   # exact counts, past depth too
    assert max(int(c) for _, c in rows) == 4
