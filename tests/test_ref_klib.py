"""libfemhost's sequence readers and the mapping sort against THE REFERENCE'S OWN klib code: oracle/_ref/libfemref_klib.so
is the reference's src/kseq.h and src/ksort.h, compiled from where they lie under /root/reference behind the harness
oracle/ref_klib.c (`make -C oracle ref`; everything else of the reference needs htslib and cannot be built here).

  * records: kseq_read as src/sequence_batch.c:47-66 drives it (zero-length records skipped, any return below -1 fatal)
    vs fem_seqfile_read (sequential), fem_seqfile_read_bytes and fem_seqfile_plan/_fill (multi-threaded), plain and gzip
  * order of a read's mappings: radix_sort_mapping (src/align.c:53-57) vs the oracle's restatement, which the host tail
    and the device ordering kernel are compared with elsewhere (tests/test_host.py, tests/test_gpu_tail.py)

Without the library (a checkout that never saw /root/reference) the committed vectors under tests/golden/klib_*.npz,
made from it by tests/golden/make_klib_golden.py, stand in."""
import gzip
import os

import numpy as np
import pytest

from fem_amd import host
from oracle import fem_oracle as fo
from oracle import ref_klib

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
needs_ref = pytest.mark.skipif(not ref_klib.available(), reason="oracle/_ref/libfemref_klib.so not built (no /root/reference here)")


def loader_view(records, last_rc):
    """What load_batch_of_sequences_into_sequence_batch keeps (src/sequence_batch.c:47-66): zero-length records are
    skipped; a return value other than -1 at the end is "Didn't reach the end of sequence file" -> exit."""
    return [r for r in records if len(r[2]) > 0], last_rc != -1


def ours_sequential(path):
    try:
        s = host.read_sequences(path)
    except ValueError:
        return None, True
    return [(s.name(i).encode(), s.seq(i), s.qual(i) if s.quals is not None else None) for i in range(s.n)], False


def ours_planned(path, approx, threads):
    try:
        parts = host.read_planned_batches(path, approx, threads=threads)
    except ValueError:
        return None, True
    out = []
    for p in parts:
        out += [(p.name(i).encode(), p.seq(i), p.qual(i) if p.quals is not None else None) for i in range(p.n)]
    return out, False


def same_records(ref, mine):
    all_qual = all(r[3] is not None for r in ref)
    assert len(ref) == len(mine)
    for (name, _comment, seq, qual), (n2, s2, q2) in zip(ref, mine):
        assert name == n2 and seq == s2
        if all_qual and ref:
            assert qual == q2  # (a batch keeps its qualities only if every record has them, fem_host.cc finish_seqset)


HAND_MADE = {
    "four_line": b"@r1 first comment\nACGT\n+\nIIII\n@r2\tTAB comment\nGGCC\n+r2\n!!!!\n",
    "crlf": b"@r1 c\r\nACGT\r\n+\r\nIIII\r\n@r2\r\nGG\r\n+\r\nJJ\r\n",
    "multi_line": b"@a\nACGT\nACG\n+\nIIII\nIII\n@b x\nTTTT\n+\nJJJJ\n",
    "fasta": b">s1 desc\nACGTACGT\nACGT\n>s2\nGG\n\n>s3\nTTTT",
    "mixed": b">f\nACGT\n@q\nGGGG\n+\nIIII\n>g\nCC\n",
    "at_plus_quals": b"@r1\nACGT\n+\n@+@+\n@r2\nACGT\n+\n+@+@\n@r3\nAC\n+\n@@\n",
    "blank_lines": b"\n\n@r1\nACGT\n+\nIIII\n\n\n@r2\nGG\n+\nII\n\n",
    "leading_junk": b"junk line\nmore junk\n@r1\nACGT\n+\nIIII\n",
    "no_final_newline": b"@r1\nACGT\n+\nIIII\n@r2\nGG\n+\nII",
    "zero_length": b"@e1\n\n+\n\n@r1\nACGT\n+\nIIII\n@e2\n\n+\n\n",
    "truncated_qual": b"@r1\nACGT\n+\nIIII\n@r2\nACGT\n+\nII\n",
    "missing_qual_at_eof": b"@r1\nACGT\n+\nIIII\n@r2\nACGT\n+\n",
    "header_only": b"@r1\n",
    "empty": b"",
    "only_newlines": b"\n\n\n",
    "lower_and_n": b"@r1\nacgtNNnn\n+\nIIIIIIII\n",
    "long_name": b"@" + b"n" * 300 + b" " + b"c" * 500 + b"\nACGT\n+\nIIII\n",
    "qual_longer": b"@r1\nACGT\n+\nIIIIII\n@r2\nGG\n+\nII\n",
    "plus_in_seq_line": b"@r1\nAC+GT\n+\nIIIII\n",
    "gt_in_fastq": b"@r1\nACGT\n+\nIIII\n>f1\nACGT\n@r2\nGG\n+\nII\n",
}


@needs_ref
@pytest.mark.parametrize("name", sorted(HAND_MADE))
def test_readers_follow_kseq_on_hand_made_files(tmp_path, name):
    data = HAND_MADE[name]
    for gz in (False, True):
        p = str(tmp_path / (name + (".fq.gz" if gz else ".fq")))
        with (gzip.open(p, "wb") if gz else open(p, "wb")) as f:
            f.write(data)
        ref, fatal = loader_view(*ref_klib.kseq_records(p))
        mine, failed = ours_sequential(p)
        assert failed == fatal, name
        if not fatal:
            same_records(ref, mine)
        for approx, threads in ((0, 1), (64, 3)):
            mine, failed = ours_planned(p, approx, threads)
            assert failed == fatal, (name, approx)
            if not fatal:
                same_records(ref, mine)


def _random_file(rng):
    """FASTQ-like text with the irregularities kseq tolerates, then a few random byte edits."""
    out = []
    for i in range(int(rng.integers(1, 40))):
        ln = int(rng.integers(0, 70))
        seq = bytes(rng.choice(list(b"ACGTNacgt"), size=ln).astype(np.uint8))
        qual = bytes(rng.integers(33, 75, size=ln).astype(np.uint8))
        kind = rng.random()
        name = b"r%d" % i + (b" comment %d" % i if rng.random() < 0.5 else b"")
        eol = b"\r\n" if rng.random() < 0.1 else b"\n"
        if kind < 0.7:
            rec = b"@" + name + eol + seq + eol + b"+" + eol + qual + eol
        elif kind < 0.8 and ln > 4:  # multi-line
            h = ln // 2
            rec = b"@" + name + eol + seq[:h] + eol + seq[h:] + eol + b"+" + eol + qual[:h] + eol + qual[h:] + eol
        elif kind < 0.9:
            rec = b">" + name + eol + seq + eol
        else:
            rec = b"@" + name + eol + seq + eol + b"+" + name + eol + qual + eol + b"\n"
        out.append(rec)
    data = bytearray(b"".join(out))
    for _ in range(int(rng.integers(0, 3))):  # damage: delete, insert or change a byte
        if not data:
            break
        at = int(rng.integers(0, len(data)))
        op = rng.random()
        if op < 0.4:
            del data[at]
        elif op < 0.7:
            data.insert(at, int(rng.choice(list(b"@+>\nAI "))))
        else:
            data[at] = int(rng.choice(list(b"@+>\nAI ")))
    if rng.random() < 0.3 and data:
        data = data[:int(rng.integers(1, len(data) + 1))]  # cut off
    return bytes(data)


@needs_ref
def test_readers_follow_kseq_on_random_damaged_files(tmp_path):
    rng = np.random.default_rng(2024)
    n_fatal = n_ok = 0
    for case in range(1500):
        data = _random_file(rng)
        p = str(tmp_path / ("f%d.fq" % case))
        with open(p, "wb") as f:
            f.write(data)
        ref, fatal = loader_view(*ref_klib.kseq_records(p))
        mine, failed = ours_sequential(p)
        assert failed == fatal, (case, data)
        if not fatal:
            same_records(ref, mine)
        mine, failed = ours_planned(p, int(rng.choice([0, 50, 300])), int(rng.integers(1, 5)))
        assert failed == fatal, (case, data)
        if not fatal:
            same_records(ref, mine)
        n_fatal += fatal
        n_ok += not fatal
    assert n_fatal > 300 and n_ok > 300


def _sort_cases():
    rng = np.random.default_rng(77)
    cases = []
    for n in [0, 1, 2, 5, 63, 64, 65, 66, 100, 128, 129, 200, 257, 448, 1000, 5000]:
        for ties in (2, 8, 1 << 40):
            # MappingSortKey: edit distance << 60 | direction << 59 | position (src/align.c:53); few distinct values = many ties
            ed = rng.integers(0, 8, size=n).astype(np.uint64) << np.uint64(60)
            direction = rng.integers(0, 2, size=n).astype(np.uint64) << np.uint64(59)
            pos = rng.integers(0, ties, size=n).astype(np.uint64) * np.uint64(3 if ties < 100 else 1)
            cases.append(ed | direction | pos)
    return cases


def _oracle_sort(keys):
    k = keys.copy()
    perm = np.zeros(len(k), np.uint32)
    fo.lib().fo_sort_mapping_keys(k.ctypes.data, perm.ctypes.data, len(k))
    return k, perm


@needs_ref
def test_mapping_order_follows_the_reference_radix_sort():
    n_tied = 0
    for keys in _sort_cases():
        want_k, want_perm = ref_klib.radix_sort(keys)
        got_k, got_perm = _oracle_sort(keys)
        assert np.array_equal(want_k, got_k)
        assert np.array_equal(want_perm, got_perm), "order among equal keys"
        n_tied += len(keys) - len(np.unique(keys))
    assert n_tied > 5000


def test_golden_vectors_made_from_the_reference_klib(tmp_path):
    """The same two comparisons against vectors committed under tests/golden/ (made by make_klib_golden.py from
    oracle/_ref): they hold wherever the tests run, with or without the reference."""
    g = np.load(os.path.join(GOLDEN, "klib_sort.npz"))
    for i in range(int(g["n_cases"])):
        keys, perm = g["keys_%d" % i], g["perm_%d" % i]
        got_k, got_perm = _oracle_sort(keys)
        assert np.array_equal(got_perm, perm) and np.array_equal(got_k, keys[perm])
    g = np.load(os.path.join(GOLDEN, "klib_kseq.npz"), allow_pickle=False)
    for i in range(int(g["n_cases"])):
        data = g["file_%d" % i].tobytes()
        p = str(tmp_path / ("g%d.fq" % i))
        with open(p, "wb") as f:
            f.write(data)
        fatal = bool(g["fatal_%d" % i])
        mine, failed = ours_sequential(p)
        assert failed == fatal
        if fatal:
            continue
        names = g["names_%d" % i].tobytes().split(b"\x00")[:-1] if g["names_%d" % i].size else []
        seqs = g["seqs_%d" % i].tobytes().split(b"\x00")[:-1] if g["seqs_%d" % i].size else []
        assert [m[0] for m in mine] == names and [m[1] for m in mine] == seqs
        planned, failed = ours_planned(p, 64, 2)
        assert not failed and [m[1] for m in planned] == seqs


def _random_fasta(rng):
    """Multi-line FASTA as references come, plus what kseq tolerates or reacts to: blank lines, '>' inside lines, empty
    records, comments, junk in front, lines that begin with '@' or '+', a carriage return, no final newline."""
    out = [b"junk before\n"] if rng.random() < 0.1 else []
    for i in range(int(rng.integers(1, 12))):
        ln = int(rng.integers(0, 400))
        width = int(rng.integers(5, 80))
        seq = bytes(rng.choice(list(b"ACGTNacgtn>"), size=ln, p=[.2, .2, .2, .2, .05, .03, .03, .03, .03, .02, .01]).astype(np.uint8))
        lines = [seq[j:j + width] for j in range(0, ln, width)]
        if rng.random() < 0.15:
            lines.insert(int(rng.integers(0, len(lines) + 1)), b"")  # blank line
        quirk = rng.random()
        if quirk < 0.03 and lines:
            lines[int(rng.integers(0, len(lines)))] = b"@odd"
        elif quirk < 0.06 and lines:
            lines[int(rng.integers(0, len(lines)))] = b"+odd"
        elif quirk < 0.09 and lines:
            lines[int(rng.integers(0, len(lines)))] += b"\r"
        name = b"s%d" % i + (b"\tdesc %d" % i if rng.random() < 0.4 else b"")
        out.append(b">" + name + b"\n" + b"".join(l + b"\n" for l in lines))
    data = b"".join(out)
    return data[:-1] if rng.random() < 0.2 and data.endswith(b"\n") else data


@needs_ref
def test_parallel_fasta_reader_follows_kseq(tmp_path, monkeypatch):
    # fem_seqfile_read on a plain FASTA file is read_fasta_parallel (fem_host.cc) when the file is regular enough, the
    # sequential reader otherwise: either way kseq's records.  Small pieces so that every body is cut many times.
    rng = np.random.default_rng(31)
    n_multi = 0
    for case in range(600):
        monkeypatch.setenv("FEM_FASTA_PIECE", str(int(rng.choice([16, 40, 200, 100000]))))
        data = _random_fasta(rng)
        p = str(tmp_path / ("r%d.fa" % case))
        with open(p, "wb") as f:
            f.write(data)
        ref, fatal = loader_view(*ref_klib.kseq_records(p))
        mine, failed = ours_sequential(p)
        assert failed == fatal, (case, data)
        if not fatal:
            assert len(ref) == len(mine), (case, data)
            for (name, _c, seq, _q), (n2, s2, _q2) in zip(ref, mine):
                assert name == n2 and seq == s2, (case, data)
            n_multi += len(ref) > 1
    assert n_multi > 300
