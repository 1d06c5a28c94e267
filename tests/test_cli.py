"""The drop-in command line (fem_amd/csrc/FEM): argument handling without a GPU, end-to-end `index` + `map` with one."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from oracle import fem_oracle as fo
from tests import util
from tests.test_host import expected_sam

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FEM = os.path.join(ROOT, "fem_amd", "csrc", "FEM")


def run(*args, env=None):
    full = dict(os.environ, **env) if env else None
    return subprocess.run([FEM] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=full)


def test_usage_and_argument_errors():
    import __graft_entry__ as g
    g.build()
    r = run()
    assert r.returncode == 1 and b"Usage:   FEM <command>" in r.stderr           # src/FEM.c:24-28
    r = run("frobnicate")
    assert r.returncode == 1 and b"unrecognized command" in r.stderr               # src/FEM.c:37-39
    r = run("index", "12", "3")
    assert r.returncode == 1 and b"Usage: FEM index <window_size> <step_size> <reference> <output>" in r.stderr
    r = run("map", "-h")
    assert r.returncode == 0 and b"--read1" in r.stderr                            # src/FEM_map.c:123-125
    r = run("map", "-e", "9", "--ref", "a", "--index", "b", "--read1", "c", "-o", "d")
    assert r.returncode == 1 and b"Wrong error threshold." in r.stderr             # src/FEM_map.c:30-33
    r = run("map", "-e", "3", "--index", "b", "--read1", "c", "-o", "d")
    assert r.returncode == 1 and b"Reference file path is required." in r.stderr
    r = run("map", "-a", "3", "--ref", "a", "--index", "b", "--read1", "c", "-o", "d")
    assert r.returncode == 1 and b"Wrong number of additional q-grams." in r.stderr
    r = run("map", "-f", "x")
    assert r.returncode == 1 and b"Wrong name of seeding algorithm!" in r.stderr   # src/FEM_map.c:113-117


def write_case(tmp_path, seed, e, L, n_reads, gz):
    rng = np.random.default_rng(seed)
    seqs = util.repeat_rich_reference(rng, n_seq=3, unit_len=300, n_units=4, copies=40, spacer=200)
    seqs.append(util.rand_seq(rng, 150_000))
    names = ["chr%s" % c for c in "ABCD"]
    reads = util.make_reads(rng, seqs, n_reads, L, e, n_rate=0.002)
    rnames = ["read_%d" % i for i in range(n_reads)]
    quals = ["".join(chr(33 + (7 * i + j) % 41) for j in range(L)) for i in range(n_reads)]
    fa = tmp_path / "ref.fa"
    with open(fa, "wb") as f:
        for n, s in zip(names, seqs):
            f.write(b">" + n.encode() + b" some description\n")
            for i in range(0, len(s), 70):
                f.write(s[i:i + 70] + b"\n")
    fq = tmp_path / ("reads.fq.gz" if gz else "reads.fq")
    op = gzip.open if gz else open
    with op(fq, "wb") as f:
        for n, r, q in zip(rnames, reads, quals):
            f.write(b"@" + n.encode() + b" 1:N:0\n" + r + b"\n+\n" + q.encode() + b"\n")
    return seqs, names, reads, rnames, quals, str(fa), str(fq)


@pytest.mark.gpu
@pytest.mark.parametrize("e,L,gz,batch", [(3, 100, False, 97), (7, 150, True, 1000000)])
def test_index_and_map_end_to_end(tmp_path, e, L, gz, batch):
    seqs, names, reads, rnames, quals, fa, fq = write_case(tmp_path, 77 + e, e, L, 600, gz)
    ref = fo.Reference(seqs)
    idx = fo.OracleIndex(ref)
    want = fo.map_reads(ref, idx, fo.ReadBatch(reads), e=e)
    index_path, sam_path, oracle_index = str(tmp_path / "ref.idx"), str(tmp_path / "out.sam"), str(tmp_path / "o.idx")
    r = run("index", "12", "3", fa, index_path)
    assert r.returncode == 0, r.stderr.decode()
    idx.save(oracle_index)
    assert open(index_path, "rb").read() == open(oracle_index, "rb").read()  # byte-identical index file
    r = run("map", "-e", str(e), "-t", "3", "--ref", fa, "--index", index_path, "--read1", fq, "-o", sam_path,
            "--batch", str(batch))
    assert r.returncode == 0, r.stderr.decode()
    text = open(sam_path).read()
    header = "".join("@SQ\tSN:%s\tLN:%d\n" % (n, len(s)) for n, s in zip(names, seqs))
    assert text.startswith(header)
    assert text[len(header):] == expected_sam(names, reads, rnames, quals, want)
    # the same run with the ordering / traceback / MD done by libfemhost instead of the device
    host_sam = str(tmp_path / "host.sam")
    r2 = run("map", "-e", str(e), "-t", "3", "--ref", fa, "--index", index_path, "--read1", fq, "-o", host_sam,
             "--batch", str(batch), env={"FEM_HOST_TAIL": "1"})
    assert r2.returncode == 0, r2.stderr.decode()
    assert open(host_sam).read() == text
    # ... with the records from the device and the text spliced by the host threads between the fields of the input file's
    # mapping (the default renders the text on the device), and with those fields copied by the parser instead of read in place
    # ... and with the device's text but the qualities kept on the host (the default from 24 threads on), packed and as characters
    for env in ({"FEM_HOST_FORMAT": "1"}, {"FEM_HOST_FORMAT": "1", "FEM_SPLICE": "0"}, {"FEM_PACK_BASES": "0"}, {"FEM_HOST_QUALS": "1"},
                {"FEM_HOST_QUALS": "1", "FEM_PACK_BASES": "0"}):
        r3 = run("map", "-e", str(e), "-t", "3", "--ref", fa, "--index", index_path, "--read1", fq, "-o", host_sam,
                 "--batch", str(batch), env=env)
        assert r3.returncode == 0, r3.stderr.decode()
        assert open(host_sam).read() == text, env
    err = r.stderr.decode()
    for label, v in zip(["The number of read", "The number of mapped read",
                         "The number of candidate before additional q-gram filter", "The number of candidate",
                         "The number of mapping"], want.stats):
        assert "%s: %d\n" % (label, int(v)) in err  # src/FEM_map.c:214-218


@pytest.mark.gpu
def test_map_fails_loudly_on_missing_or_truncated_reads(tmp_path):
    # the reference exits with EXIT_FAILURE in both cases (src/sequence_batch.c:33-35, 63-66): no statistics, no "Time:"
    seqs, names, reads, rnames, quals, fa, fq = write_case(tmp_path, 5, 3, 100, 300, False)
    index_path = str(tmp_path / "ref.idx")
    assert run("index", "12", "3", fa, index_path).returncode == 0
    r = run("map", "-e", "3", "-t", "2", "--ref", fa, "--index", index_path, "--read1", str(tmp_path / "nope.fq"), "-o",
            str(tmp_path / "a.sam"))
    assert r.returncode != 0 and b"Cannot find sequence file!" in r.stderr and b"Time:" not in r.stderr
    data = open(fq, "rb").read()
    cut = tmp_path / "cut.fq"
    cut.write_bytes(data[:len(data) - 37])  # ends inside a quality line
    r = run("map", "-e", "3", "-t", "2", "--ref", fa, "--index", index_path, "--read1", str(cut), "-o", str(tmp_path / "b.sam"),
            "--batch", "50")
    assert r.returncode != 0 and b"Didn't reach the end of sequence file" in r.stderr and b"Time:" not in r.stderr
    r = run("map", "-e", "3", "-t", "2", "--ref", fa, "--index", index_path, "--read1", fq, "-o", "/nonexistent_dir/x.sam")
    assert r.returncode != 0 and b"Cannot open output file" in r.stderr
    r = run("map", "-e", "3", "-t", "2", "--ref", fa, "--index", index_path, "--read1", fq, "-o", "/dev/full")
    assert r.returncode != 0 and b"write error" in r.stderr


def _n_gpus():
    import torch
    return torch.cuda.device_count()  # (does not initialise the GPU)


@pytest.mark.gpu
@pytest.mark.parametrize("shared", [True, False])
def test_map_on_several_gpus_gives_the_same_records_and_counters(tmp_path, shared):
    # src/FEM_map.c:200-212: the mapping threads' counters are summed at the end; here one RCCL all-reduce over the GPUs.
    # shared: three workers with a handle each on GPU 0 (FEM_TEST_SHARE_GPU=1) — the `--gpus N` pipeline (one thread per
    # handle, batches dealt to whichever has a free slot) on a one-GPU box; not shared: two real GPUs + RCCL
    if not shared and _n_gpus() < 2:
        pytest.skip("needs two GPUs")
    seqs, names, reads, rnames, quals, fa, fq = write_case(tmp_path, 21, 3, 100, 900, False)
    index_path = str(tmp_path / "ref.idx")
    assert run("index", "12", "3", fa, index_path).returncode == 0
    outs = []
    for gpus in ("1", "3" if shared else "2"):
        sam = str(tmp_path / ("g%s.sam" % gpus))
        r = run("map", "-e", "3", "-t", "4", "--gpus", gpus, "--ref", fa, "--index", index_path, "--read1", fq, "-o", sam,
                "--batch", "120", env={"FEM_TEST_SHARE_GPU": "1"} if shared else None)
        assert r.returncode == 0, r.stderr.decode()
        counters = [l for l in r.stderr.decode().splitlines() if l.startswith("The number of")]
        outs.append((sorted(open(sam).read().splitlines()), counters))
    assert outs[0] == outs[1]


@pytest.mark.gpu
def test_allreduce_stats_over_two_handles():
    if _n_gpus() < 2:
        pytest.skip("needs two GPUs")
    import ctypes as C
    from fem_amd import Device
    from fem_amd.device import load_hip
    a, b = Device(0), Device(1)
    L = load_hip()
    hs = (C.c_void_p * 2)(a._h, b._h)
    for rep in range(2):  # the second call reuses the communicator
        st = np.array([[1, 2, 3, 4, 5], [10, 20, 30, 40, 2 ** 40 + rep]], dtype=np.uint64)
        assert L.fem_dev_allreduce_stats(hs, 2, st.ctypes.data) == 0
        assert np.array_equal(st[0], st[1]) and st[0].tolist() == [11, 22, 33, 44, 2 ** 40 + rep + 5]
    a.close()
    b.close()


@pytest.mark.gpu
def test_allreduce_stats_over_one_handle_runs_rccl():
    # fem_dev_allreduce_stats with a single handle: the communicator is created (ncclCommInitAll over one device) and the
    # reduction runs over RCCL on a one-GPU box; the second and third call reuse the communicator
    import ctypes as C
    from fem_amd import Device
    from fem_amd.device import load_hip
    a = Device(0)
    L = load_hip()
    hs = (C.c_void_p * 1)(a._h)
    for rep in range(3):
        st = np.array([[7, 8, 9, 2 ** 41 + rep, 11]], dtype=np.uint64)
        assert L.fem_dev_allreduce_stats(hs, 1, st.ctypes.data) == 0
        assert st[0].tolist() == [7, 8, 9, 2 ** 41 + rep, 11]
    a.close()


@pytest.mark.gpu
def test_map_regrows_its_staging_for_unusual_records(tmp_path):
    # very long read names and short reads: a FASTQ window holds far more name bytes than the staging buffers were sized
    # for (fem_main.cc: kRegrow), with the text rendered on the device and on the host
    rng = np.random.default_rng(8)
    seqs = [util.rand_seq(rng, 120_000)]
    reads = util.make_reads(rng, seqs, 700, 64, 1)
    rnames = ["read_%d_%s" % (i, "x" * (250 + i % 90)) for i in range(len(reads))]
    quals = ["".join(chr(40 + (i + j) % 30) for j in range(64)) for i in range(len(reads))]
    fa, fq = tmp_path / "r.fa", tmp_path / "r.fq"
    fa.write_bytes(b">chrL\n" + seqs[0] + b"\n")
    with open(fq, "wb") as f:
        for n, r, q in zip(rnames, reads, quals):
            f.write(b"@" + n.encode() + b"\n" + r + b"\n+\n" + q.encode() + b"\n")
    ref = fo.Reference(seqs)
    idx = fo.OracleIndex(ref)
    want = fo.map_reads(ref, idx, fo.ReadBatch(reads), e=1)
    ix = str(tmp_path / "r.idx")
    assert run("index", "12", "3", str(fa), ix).returncode == 0
    exp = "@SQ\tSN:chrL\tLN:%d\n" % len(seqs[0]) + expected_sam(["chrL"], reads, rnames, quals, want)
    for env in (None, {"FEM_HOST_FORMAT": "1"}, {"FEM_HOST_FORMAT": "1", "FEM_SPLICE": "0"}, {"FEM_PACK_BASES": "0"}, {"FEM_HOST_QUALS": "1"}):
        out = str(tmp_path / "o.sam")
        r = run("map", "-e", "1", "-t", "4", "--ref", str(fa), "--index", ix, "--read1", str(fq), "-o", out, "--batch", "200", env=env)
        assert r.returncode == 0, r.stderr.decode()
        assert open(out).read() == exp
    assert want.stats[1] > 300
