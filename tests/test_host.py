"""Host-side product code (libfemhost.so) against the oracle and against the file formats; runs without a GPU."""
import gzip
import os

import numpy as np
import pytest

from fem_amd import host
from oracle import fem_oracle as fo
from tests import util


def oracle_case(seed, e, a=1, L=100, n_reads=250, repeat=True):
    rng = np.random.default_rng(seed)
    if repeat:
        seqs = util.repeat_rich_reference(rng, n_seq=3, unit_len=max(300, 2 * L), n_units=3, copies=110, spacer=150)
    else:
        seqs = [util.rand_seq(rng, 150_000), util.rand_seq(rng, 80_000)]
    reads = util.make_reads(rng, seqs, n_reads, L, e, n_rate=0.002)
    ref = fo.Reference(seqs)
    idx = fo.OracleIndex(ref)
    batch = fo.ReadBatch(reads)
    res = fo.map_reads(ref, idx, batch, e=e, a=a)
    return seqs, reads, ref, batch, res


def as_device_layout(res):
    """Oracle candidates in the layout libfemhip hands back (begin/count per (read, strand))."""
    begin = res.cand_off[:-1].astype(np.uint32)
    count = np.diff(res.cand_off.astype(np.int64)).astype(np.uint32)
    return begin, count, res.cands, res.v_ed, res.v_end


@pytest.mark.parametrize("e,L,repeat", [(3, 100, True), (7, 150, True), (2, 80, False), (1, 64, True)])
def test_tail_records_equal_oracle(e, L, repeat):
    seqs, reads, ref, batch, res = oracle_case(40 + e, e, L=L, repeat=repeat)
    tref = host.TailReference(ref.text, ref.off, ref.len)
    begin, count, cand, ed, end = as_device_layout(res)
    for threads in (1, 3):
        got = host.tail_records(e, tref, batch.bases, batch.off, begin, count, cand, ed, end, threads=threads)
        assert np.array_equal(got.rec_off, res.rec_off)
        assert np.array_equal(got.flag, res.r_flag)
        assert np.array_equal(got.tid, res.r_tid)
        assert np.array_equal(got.pos0, res.r_pos)
        assert np.array_equal(got.nm, res.r_nm)
        assert np.array_equal(got.cigar_off, res.cig_off)
        assert np.array_equal(got.cigar, res.cig)
        assert np.array_equal(got.md_off, res.md_off)
        assert np.array_equal(got.md, res.md)
    per_read = np.diff(res.rec_off.astype(np.int64))
    if repeat and e >= 3:
        assert per_read.max() > 64, "fixture must exercise klib's radix path (>64 mappings of one read)"
    assert np.any((res.cig & 0xF) == 1) and np.any((res.cig & 0xF) == 2), "fixture must contain I and D"


def expected_sam(seqs_names, reads, names, quals, res):
    """SAM v1 text for the oracle's records (field rules of src/align.c:546-632, src/map.c:50-55)."""
    lines = []
    for r in range(len(reads)):
        lo, hi = int(res.rec_off[r]), int(res.rec_off[r + 1])
        for j in range(lo, hi):
            primary = j == lo
            lines.append("\t".join([
                names[r], str(int(res.r_flag[j])), seqs_names[int(res.r_tid[j])], str(int(res.r_pos[j]) + 1), "255",
                res.cigar_str(j), "*", "0", "0",
                reads[r].decode().upper() if primary else "*", quals[r] if primary else "*",
                "NM:i:%d" % int(res.r_nm[j]), "MD:Z:" + res.md_str(j)]))
    return "".join(l + "\n" for l in lines)


def test_sam_text_follows_the_record_rules():
    e = 3
    seqs, reads, ref, batch, res = oracle_case(7, e, n_reads=120)
    names = ["r%d" % i for i in range(len(reads))]
    quals = ["".join(chr(33 + (i * 7 + j) % 40) for j in range(len(r))) for i, r in enumerate(reads)]
    tref = host.TailReference(ref.text, ref.off, ref.len, names=["chrA", "chrB", "chrC"])
    begin, count, cand, ed, end = as_device_layout(res)
    q = np.frombuffer("".join(quals).encode(), np.uint8)
    text = host.tail_sam(e, tref, names, batch.bases, batch.off, q, begin, count, cand, ed, end, threads=2)
    assert text == expected_sam(["chrA", "chrB", "chrC"], reads, names, quals, res)
    assert host.sam_header(tref) == "".join("@SQ\tSN:%s\tLN:%d\n" % (n, len(s)) for n, s in zip(["chrA", "chrB", "chrC"], seqs))
    # unmapped reads emit nothing; secondary records carry '*' for SEQ and QUAL and flag 256
    n_mapped = int(np.count_nonzero(np.diff(res.rec_off.astype(np.int64))))
    assert len({l.split("\t")[0] for l in text.splitlines()}) == n_mapped
    assert any(int(l.split("\t")[1]) & 256 and l.split("\t")[9] == "*" for l in text.splitlines())


def test_fasta_fastq_parsing_follows_kseq(tmp_path):
    fa = tmp_path / "ref.fa"
    fa.write_bytes(b">chr1 first comment\nACGTAC\nGTNN\n\nacgt\n>empty\n>chr2\tx\r\nGGCC\r\nTT\r\n>last\nA")
    s = host.read_sequences(str(fa))
    assert [s.name(i) for i in range(s.n)] == ["chr1", "chr2", "last"]  # zero-length record skipped
    assert [s.seq(i) for i in range(s.n)] == [b"ACGTACGTNNacgt", b"GGCCTT", b"A"]
    assert s.quals is None
    fq = tmp_path / "reads.fq.gz"
    with gzip.open(str(fq), "wb") as f:
        f.write(b"@r0 desc\nACGT\n+\nIIII\n@r1\nGGGTTT\n+r1\n@@@+++\n@r2/1\nAC\nGT\n+\nII\nII\n")
    s = host.read_sequences(str(fq))
    assert [s.name(i) for i in range(s.n)] == ["r0", "r1", "r2/1"]
    assert [s.seq(i) for i in range(s.n)] == [b"ACGT", b"GGGTTT", b"ACGT"]
    assert [s.qual(i) for i in range(s.n)] == [b"IIII", b"@@@+++", b"IIII"]
    s2 = host.read_sequences(str(fq), max_seqs=2)
    assert s2.n == 2
    bad = tmp_path / "bad.fq"
    bad.write_bytes(b"@r0\nACGT\n+\nII\n")
    with pytest.raises(ValueError):
        host.read_sequences(str(bad))


def test_parallel_fastq_reader_equals_sequential_reader(tmp_path):
    rng = np.random.default_rng(11)
    n, L = 30_000, 100
    recs = []
    for i in range(n):
        ln = L if i % 7 else int(rng.integers(1, 2 * L))
        seq = util.rand_seq(rng, ln)
        qual = bytes(rng.integers(33, 74, size=ln).astype(np.uint8))  # includes '@' and '+' as first characters
        recs.append(b"@r%d extra\n" % i + seq + (b"\r\n" if i % 11 == 0 else b"\n") + b"+\n" + qual + b"\n")
    recs.insert(5, b"@empty\n\n+\n\n")  # zero-length record: skipped by both readers
    fq = tmp_path / "big.fq"
    fq.write_bytes(b"".join(recs))
    assert fq.stat().st_size > 4 << 20
    whole = host.read_sequences(str(fq))
    assert whole.n == n
    for approx in (0, 1 << 20, 700_001):
        parts = host.read_sequences_in_chunks(str(fq), approx, threads=4)
        assert sum(p.n for p in parts) == n
        assert b"".join(p.bases.tobytes() for p in parts) == whole.bases.tobytes()
        assert b"".join(p.quals.tobytes() for p in parts) == whole.quals.tobytes()
        assert b"".join(p.names_raw.tobytes() for p in parts) == whole.names_raw.tobytes()
        lens = np.concatenate([np.diff(p.off.astype(np.int64)) for p in parts])
        assert np.array_equal(lens, np.diff(whole.off.astype(np.int64)))
    # multi-line FASTQ and gzip take the sequential path and give the same records
    ml = tmp_path / "multiline.fq"
    ml.write_bytes(b"@a\nACGT\nACG\n+\nIIII\nIII\n@b x\nTTTT\n+\nJJJJ\n")
    parts = host.read_sequences_in_chunks(str(ml), 8, threads=3)
    assert [p.seq(i) for p in parts for i in range(p.n)] == [b"ACGTACG", b"TTTT"]
    gzp = tmp_path / "big.fq.gz"
    with gzip.open(str(gzp), "wb", compresslevel=1) as f:
        f.write(fq.read_bytes())
    parts = host.read_sequences_in_chunks(str(gzp), 1 << 20, threads=4)
    assert b"".join(p.bases.tobytes() for p in parts) == whole.bases.tobytes()


def test_two_phase_reader_equals_sequential_reader(tmp_path):
    # fem_seqfile_plan + fem_seqfile_fill (what `FEM map` uses to parse straight into the device library's pinned staging
    # buffers) against the kseq-rule sequential reader: plain FASTQ (scanned twice, never copied in between), CRLF,
    # '@' / '+' leading quality lines, blank lines, zero-length records, multi-line and gzip (held by the plan)
    rng = np.random.default_rng(12)
    n, L = 20_000, 100
    recs = []
    for i in range(n):
        ln = L if i % 5 else int(rng.integers(1, 3 * L))
        seq = util.rand_seq(rng, ln)
        qual = bytes(rng.integers(33, 74, size=ln).astype(np.uint8))
        recs.append(b"@q%d some comment\n" % i + seq + (b"\r\n" if i % 13 == 0 else b"\n") + b"+\n" + qual + (b"\n\n" if i % 17 == 0 else b"\n"))
    recs.insert(3, b"@empty\n\n+\n\n")
    fq = tmp_path / "two_phase.fq"
    fq.write_bytes(b"".join(recs)[:-1])  # no newline at the end of the file
    whole = host.read_sequences(str(fq))
    assert whole.n == n
    for approx, threads in ((0, 4), (300_000, 3), (1 << 20, 8), (64, 2)):
        if approx == 64 and n > 5000:
            continue
        parts = host.read_planned_batches(str(fq), approx, threads=threads)
        assert sum(p.n for p in parts) == n
        assert b"".join(p.bases[:int(p.off[p.n])].tobytes() for p in parts) == whole.bases.tobytes()
        assert b"".join(p.quals[:int(p.off[p.n])].tobytes() for p in parts) == whole.quals.tobytes()
        assert b"".join(p.names_raw[:int(p.name_off[p.n])].tobytes() for p in parts) == whole.names_raw.tobytes()
        assert all(int(p.off[0]) == 0 and p.max_len == int(np.diff(p.off.astype(np.int64)).max()) for p in parts)
        assert all(not p.bases[int(p.off[p.n]):int(p.off[p.n]) + 64].any() for p in parts), "64 zero bytes behind the last read"
    ml = tmp_path / "ml.fq"
    ml.write_bytes(b"@a\nACGT\nACG\n+\nIIII\nIII\n@b x\nTTTT\n+\nJJJJ\n")
    parts = host.read_planned_batches(str(ml), 8, threads=3)
    assert [p.seq(i) for p in parts for i in range(p.n)] == [b"ACGTACG", b"TTTT"]
    assert [p.name(i) for p in parts for i in range(p.n)] == ["a", "b"]
    gzp = tmp_path / "two_phase.fq.gz"
    with gzip.open(str(gzp), "wb", compresslevel=1) as f:
        f.write(fq.read_bytes())
    parts = host.read_planned_batches(str(gzp), 1 << 20, threads=4)
    assert b"".join(p.bases[:int(p.off[p.n])].tobytes() for p in parts) == whole.bases.tobytes()
    assert host.read_planned_batches(str(tmp_path / "two_phase.fq"), 0, threads=1)[0].n == n
    empty = tmp_path / "empty.fq"
    empty.write_bytes(b"")
    assert host.read_planned_batches(str(empty), 0) == []
    bad = tmp_path / "bad2.fq"
    bad.write_bytes(b"@r0\nACGT\n+\nII\n")
    with pytest.raises(ValueError):
        host.read_planned_batches(str(bad), 0)


def test_records_sam_parts_equals_the_concatenated_text():
    e = 3
    seqs, reads, ref, batch, res = oracle_case(9, e, n_reads=300)
    names = ["read_%d" % i for i in range(len(reads))]
    quals = ["".join(chr(33 + (i * 5 + j) % 41) for j in range(len(r))) for i, r in enumerate(reads)]
    tref = host.TailReference(ref.text, ref.off, ref.len, names=["chrA", "a_much_longer_sequence_name", "c"])
    q = np.frombuffer("".join(quals).encode(), np.uint8)
    begin, count, cand, ed, end = as_device_layout(res)
    want = host.tail_sam(e, tref, names, batch.bases, batch.off, q, begin, count, cand, ed, end, threads=2)

    class Rec:
        pass
    rec = Rec()
    rec.rec_begin, rec.flag, rec.tid, rec.pos0, rec.nm = res.rec_off, res.r_flag.copy(), res.r_tid, res.r_pos, res.r_nm
    rec.cigar_off, rec.cigar, rec.md_off, rec.md = res.cig_off, res.cig, res.md_off, res.md
    assert host.records_sam(tref, names, batch.bases, batch.off, q, rec, threads=3) == want
    for threads in (1, 2, 7):
        text, asserted = host.records_sam(tref, names, batch.bases, batch.off, q, rec, threads=threads, parts=True)
        assert text == want and asserted == 0
    # a record carrying the "reference would have asserted" marker (0x8000): the bit never reaches the FLAG column
    j = int(res.rec_off[np.flatnonzero(np.diff(res.rec_off.astype(np.int64)))[0]])
    rec.flag[j] |= 0x8000
    text, asserted = host.records_sam(tref, names, batch.bases, batch.off, q, rec, threads=2, parts=True)
    assert asserted == 1 and text == want


def test_index_file_is_byte_compatible(tmp_path):
    rng = np.random.default_rng(2)
    ref = fo.Reference([util.rand_seq(rng, 30_000), util.rand_seq(rng, 999)])
    idx = fo.OracleIndex(ref)
    a, b = str(tmp_path / "a.idx"), str(tmp_path / "b.idx")
    idx.save(a)
    host.index_save(b, 12, 3, idx.lookup, idx.occ[:idx.n_occ])
    assert open(a, "rb").read() == open(b, "rb").read()
    k, step, lookup, occ = host.index_load(a)
    assert (k, step) == (12, 3)
    assert np.array_equal(lookup, idx.lookup) and np.array_equal(occ, idx.occ[:idx.n_occ])


def test_synthetic_generator_is_reproducible_and_shardable():
    text, off, lens = host.synth_reference(5, [200_000, 50_000, 70], threads=4)
    text2, _, _ = host.synth_reference(5, [200_000, 50_000, 70], threads=1)
    assert np.array_equal(text, text2)
    body = text[:int(lens.sum())]
    assert set(np.unique(body).tolist()) == {65, 67, 71, 84}
    counts = np.bincount(body, minlength=256)[[65, 67, 71, 84]] / len(body)
    assert np.all(np.abs(counts - 0.25) < 0.01)
    bases, offs = host.synth_reads(9, text, off, lens, 1000, 100, 3, threads=4)
    lo, _ = host.synth_reads(9, text, off, lens, 400, 100, 3, first_read=0, threads=2)
    hi, _ = host.synth_reads(9, text, off, lens, 600, 100, 3, first_read=400, threads=3)
    assert np.array_equal(bases[:40_000], lo[:40_000]) and np.array_equal(bases[40_000:100_000], hi[:60_000])
    # the oracle maps most of them, on both strands
    ref = fo.Reference([text[int(o):int(o) + int(l)].tobytes() for o, l in zip(off, lens)])
    idx = fo.OracleIndex(ref)
    res = fo.map_reads(ref, idx, fo.ReadBatch.from_arrays(bases, offs), e=3, stages=fo.STAGE_SEED | fo.STAGE_VERIFY)
    assert res.stats[1] > 750
    assert 0.3 < np.mean(res.m_dir) < 0.7


def test_summary_slot_arithmetic_is_exact():
    """The seed kernel finds a bucket's summary word with q = uint32(float32(h >> 3) * 0.33333334f) (fem_kernels.hip.h,
    summary_slot) and its "two or more" bit with (r * 11) >> 5: both must equal the integer divisions the table is built
    with, for every bucket the table may cover (h >> 3 < 2^22)."""
    x = np.arange(1 << 22, dtype=np.uint32)
    q = (x.astype(np.float32) * np.float32(0.33333334)).astype(np.uint32)
    assert np.array_equal(q, x // 3)
    r = np.arange(24, dtype=np.uint32)
    assert np.array_equal((r * 11) >> 5, r // 3)


def test_gzip_input_is_parsed_window_by_window(tmp_path):
    # gzip streams are inflated a window at a time and each window goes through the multi-threaded 4-line parser
    # (fem_host.cc: fast_view); records that straddle a window's end, a switch to the sequential reader in the middle of
    # the stream (multi-line record), and a truncated stream
    rng = np.random.default_rng(13)
    n, L = 12_000, 100
    recs = []
    for i in range(n):
        ln = L if i % 4 else int(rng.integers(1, 3 * L))
        seq = util.rand_seq(rng, ln)
        qual = bytes(rng.integers(33, 74, size=ln).astype(np.uint8))
        recs.append(b"@g%d c\n" % i + seq + b"\n+\n" + qual + (b"\n\n" if i % 19 == 0 else b"\n"))
    plain = tmp_path / "w.fq"
    plain.write_bytes(b"".join(recs)[:-1])
    whole = host.read_sequences(str(plain))
    assert whole.n == n
    gzp = tmp_path / "w.fq.gz"
    with gzip.open(str(gzp), "wb", compresslevel=1) as f:
        f.write(plain.read_bytes())

    def same(parts, ref, n_expected):
        assert sum(p.n for p in parts) == n_expected
        assert b"".join(p.bases[:int(p.off[p.n])].tobytes() for p in parts) == ref.bases.tobytes()
        assert b"".join(p.quals[:int(p.off[p.n])].tobytes() for p in parts) == ref.quals.tobytes()
        assert b"".join(p.names_raw[:int(p.name_off[p.n])].tobytes() for p in parts) == ref.names_raw.tobytes()

    for approx, threads in ((70_000, 3), (300_000, 8), (1 << 22, 4), (0, 4)):
        parts = host.read_planned_batches(str(gzp), approx, threads=threads)
        same(parts, whole, n)
        if approx == 70_000:
            assert len(parts) > 20  # many windows
    chunks = host.read_sequences_in_chunks(str(gzp), 200_000, threads=4)
    assert b"".join(p.bases.tobytes() for p in chunks) == whole.bases.tobytes()
    # a multi-line record in the middle: the windows before it are parsed by the fast reader, the rest by the exact one
    mixed = b"".join(recs[:5000]) + b"@ml x\nACGT\nACG\n+\nIIII\nIII\n" + b"".join(recs[5000:])
    mp, mg = tmp_path / "m.fq", tmp_path / "m.fq.gz"
    mp.write_bytes(mixed)
    with gzip.open(str(mg), "wb", compresslevel=1) as f:
        f.write(mixed)
    ref = host.read_sequences(str(mp))
    assert ref.n == n + 1
    same(host.read_planned_batches(str(mg), 150_000, threads=4), ref, n + 1)
    # truncated stream: the reference exits with "Didn't reach the end of sequence file" (src/sequence_batch.c:63-66)
    cut = tmp_path / "cut.fq.gz"
    cut.write_bytes(gzp.read_bytes()[:gzp.stat().st_size * 2 // 3])
    with pytest.raises(ValueError):
        host.read_planned_batches(str(cut), 100_000, threads=2)


def _bgzf(data, block=65280, level=1, eof_block=True):
    """bgzip's format (SAM specification 4.1): gzip members with the 'BC' extra subfield, one per <= 64 KiB of input."""
    import struct
    import zlib
    out = []
    for at in list(range(0, len(data), block)) + ([None] if eof_block else []):
        chunk = b"" if at is None else data[at:at + block]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        payload = c.compress(chunk) + c.flush()
        bsize = 18 + len(payload) + 8
        out.append(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
                   + payload + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    return b"".join(out)


def test_bgzf_input_is_inflated_block_parallel(tmp_path):
    # bgzip-compressed FASTQ / FASTA: the blocks of a window are inflated by all threads (fem_host.cc: bgzf_*), the window
    # goes through the multi-threaded parser; the sequential reader (references, multi-line records) reads block by block
    rng = np.random.default_rng(14)
    n, L = 9_000, 100
    recs = []
    for i in range(n):
        ln = L if i % 6 else int(rng.integers(1, 3 * L))
        seq = util.rand_seq(rng, ln)
        qual = bytes(rng.integers(33, 74, size=ln).astype(np.uint8))
        recs.append(b"@z%d c\n" % i + seq + b"\n+\n" + qual + b"\n")
    text = b"".join(recs)
    plain = tmp_path / "b.fq"
    plain.write_bytes(text)
    whole = host.read_sequences(str(plain))
    assert whole.n == n
    for block, name in ((65280, "b.fq.gz"), (777, "small_blocks.fq.gz")):
        p = tmp_path / name
        p.write_bytes(_bgzf(text, block=block))
        assert gzip.decompress(p.read_bytes()) == text  # (it is a valid multi-member gzip file)
        for approx, threads in ((100_000, 4), (1 << 21, 8), (0, 3)):
            parts = host.read_planned_batches(str(p), approx, threads=threads)
            assert sum(q.n for q in parts) == n
            assert b"".join(q.bases[:int(q.off[q.n])].tobytes() for q in parts) == whole.bases.tobytes()
            assert b"".join(q.quals[:int(q.off[q.n])].tobytes() for q in parts) == whole.quals.tobytes()
            assert b"".join(q.names_raw[:int(q.name_off[q.n])].tobytes() for q in parts) == whole.names_raw.tobytes()
        s = host.read_sequences(str(p))  # sequential reader, block by block
        assert s.n == n and s.bases.tobytes() == whole.bases.tobytes()
    # a bgzipped multi-line FASTA reference
    fa = b"".join(b">chr%d d\n" % i + b"\n".join(util.rand_seq(rng, 200_000)[j:j + 60] for j in range(0, 200_000, 60)) + b"\n" for i in range(3))
    fp, fz = tmp_path / "r.fa", tmp_path / "r.fa.gz"
    fp.write_bytes(fa)
    fz.write_bytes(_bgzf(fa))
    a, b = host.read_sequences(str(fp)), host.read_sequences(str(fz))
    assert a.n == b.n == 3 and a.bases.tobytes() == b.bases.tobytes() and a.names_raw.tobytes() == b.names_raw.tobytes()
    # switch to the sequential reader in the middle (multi-line record), without the end-of-file block
    mixed = b"".join(recs[:4000]) + b"@ml x\nACGT\nACG\n+\nIIII\nIII\n" + b"".join(recs[4000:])
    mp, mz = tmp_path / "m.fq", tmp_path / "m.fq.gz"
    mp.write_bytes(mixed)
    mz.write_bytes(_bgzf(mixed, block=5000, eof_block=False))
    ref = host.read_sequences(str(mp))
    parts = host.read_planned_batches(str(mz), 120_000, threads=4)
    assert sum(q.n for q in parts) == ref.n == n + 1
    assert b"".join(q.bases[:int(q.off[q.n])].tobytes() for q in parts) == ref.bases.tobytes()
    # damage: cut inside a block, and a flipped payload byte (CRC)
    good = (tmp_path / "b.fq.gz").read_bytes()
    cut = tmp_path / "cut_b.fq.gz"
    cut.write_bytes(good[:len(good) // 2])
    with pytest.raises(ValueError):
        host.read_planned_batches(str(cut), 100_000, threads=2)
    bad = bytearray(good)
    bad[len(bad) // 3] ^= 0x55
    flip = tmp_path / "flip.fq.gz"
    flip.write_bytes(bytes(bad))
    with pytest.raises(ValueError):
        host.read_planned_batches(str(flip), 100_000, threads=2)
    with pytest.raises(ValueError):
        host.read_sequences(str(flip))


@pytest.mark.parametrize("kind", ["gzip", "bgzf", "plain"])
def test_sequential_reads_then_batches_on_one_handle(tmp_path, kind):
    # fem_seqfile_read(f, 10) followed by fem_seqfile_read_bytes / fem_seqfile_plan on the SAME handle: the sequential
    # reader buffers inflated bytes (1 MB per gzread, or one BGZF block) the window reader knows nothing of, so a
    # compressed source must stay with the sequential reader from then on — no record may be lost (ADVICE round 2:
    # read(10) + a read_bytes loop returned 10 of 5000 records on .fq.gz)
    import ctypes as C
    rng = np.random.default_rng(77)
    n = 5000
    recs = [b"@q%d\n" % i + util.rand_seq(rng, 100) + b"\n+\n" + b"I" * 100 + b"\n" for i in range(n)]
    text = b"".join(recs)
    p = tmp_path / ("mix.fq" + {"gzip": ".gz", "bgzf": ".bgz", "plain": ""}[kind])
    if kind == "gzip":
        with gzip.open(str(p), "wb", compresslevel=1) as f:
            f.write(text)
    elif kind == "bgzf":
        p.write_bytes(_bgzf(text))
    else:
        p.write_bytes(text)
    L = host.lib()
    for second in ("read_bytes", "plan"):
        f = L.fem_seqfile_open(str(p).encode())
        assert f
        names = []
        try:
            s = host.SeqSet()
            assert L.fem_seqfile_read(f, 10, C.byref(s)) == 0
            first = host.Sequences(s)
            L.fem_seqset_free(C.byref(s))
            names += [first.name(i) for i in range(first.n)]
            assert first.n == 10
            while True:
                if second == "read_bytes":
                    s = host.SeqSet()
                    assert L.fem_seqfile_read_bytes(f, 60_000, 4, C.byref(s)) == 0
                    part = host.Sequences(s)
                    L.fem_seqset_free(C.byref(s))
                    if part.n == 0:
                        break
                    names += [part.name(i) for i in range(part.n)]
                else:
                    plan, shape = C.c_void_p(), host.BatchShape()
                    assert L.fem_seqfile_plan(f, 60_000, 4, C.byref(plan), C.byref(shape)) == 0
                    if shape.n_reads == 0:
                        L.fem_batch_plan_free(plan)
                        break
                    k, nb = int(shape.n_reads), int(shape.n_bases)
                    bases, off = np.zeros(nb + 64, np.uint8), np.zeros(k + 1, np.uint64)
                    quals, nm, nmo = np.zeros(nb + 1, np.uint8), np.zeros(int(shape.n_name_bytes) + 1, np.uint8), np.zeros(k + 1, np.uint64)
                    assert L.fem_seqfile_fill(f, plan, 4, bases.ctypes.data, off.ctypes.data, quals.ctypes.data, nm.ctypes.data, nmo.ctypes.data) == 0
                    names += [nm[int(nmo[i]):int(nmo[i + 1])].tobytes().decode() for i in range(k)]
        finally:
            L.fem_seqfile_close(f)
        assert names == ["q%d" % i for i in range(n)], (kind, second, len(names))


def test_packed_fill_equals_the_characters(tmp_path):
    # fem_seqfile_fill_packed (the parser writing two bits per base straight into the staging that
    # fem_dev_commit_stage_packed sends): unpacked as the device unpacks it, the batch is the one fem_seqfile_fill gives,
    # byte for byte — N, lower case, IUPAC, CRLF, every batch size, plain and gzip
    rng = np.random.default_rng(77)
    for L, n in ((100, 6000), (101, 3000), (37, 2000), (150, 1500)):
        odd = [ord(c) for c in "NnacgtRY."]
        recs = []
        for i in range(n):
            seq = bytearray(util.rand_seq(rng, L))
            if i % 7 == 0:
                for at in rng.integers(0, L, int(rng.integers(1, 4))):
                    seq[int(at)] = odd[int(rng.integers(0, len(odd)))]
            if i == 0:
                seq[0] = 78
            if i == n - 1:
                seq[L - 1] = ord("n")
            qual = bytes(rng.integers(33, 74, size=L).astype(np.uint8))
            recs.append(b"@p%d c\n" % i + bytes(seq) + (b"\r\n" if i % 11 == 0 else b"\n") + b"+\n" + qual + b"\n")
        fq = tmp_path / ("packed_%d.fq" % L)
        fq.write_bytes(b"".join(recs))
        whole = host.read_sequences(str(fq))
        for approx, threads in ((0, 4), (200_000, 3), (1 << 20, 8)):
            parts = host.read_planned_batches(str(fq), approx, threads=threads, packed=True)
            assert all(isinstance(p, host.PackedBatch) for p in parts) and sum(p.n for p in parts) == n
            assert b"".join(p.unpack().tobytes() for p in parts) == whole.bases.tobytes()
            assert b"".join(p.quals[:p.n * L].tobytes() for p in parts) == whole.quals.tobytes()
            assert b"".join(p.names_raw[:int(p.name_off[p.n])].tobytes() for p in parts) == whole.names_raw.tobytes()
            n_odd = sum(int(np.count_nonzero(~np.isin(p.unpack(), [65, 67, 71, 84]))) for p in parts)
            assert sum(p.n_exc for p in parts) == n_odd > 0
        if L == 100:
            gzp = tmp_path / "packed.fq.gz"
            with gzip.open(str(gzp), "wb", compresslevel=1) as f:
                f.write(fq.read_bytes())
            parts = host.read_planned_batches(str(gzp), 1 << 19, threads=4, packed=True)
            assert b"".join(p.unpack().tobytes() for p in parts) == whole.bases.tobytes()
    # reads of different lengths, and a batch with too many odd characters, come back as characters
    mixed = tmp_path / "mixed.fq"
    mixed.write_bytes(b"@a\nACGTACGT\n+\nIIIIIIII\n@b\nACGTA\n+\nIIIII\n")
    parts = host.read_planned_batches(str(mixed), 0, packed=True)
    assert len(parts) == 1 and isinstance(parts[0], host.PlannedBatch) and parts[0].seq(1) == b"ACGTA"
    noisy = tmp_path / "noisy.fq"
    noisy.write_bytes(b"".join(b"@n%d\nACGTNNNNNNNNACGTACGT\n+\nIIIIIIIIIIIIIIIIIIII\n" % i for i in range(50)))
    parts = host.read_planned_batches(str(noisy), 0, packed=True)
    assert isinstance(parts[0], host.PlannedBatch) and parts[0].n == 50 and parts[0].seq(49) == b"ACGTNNNNNNNNACGTACGT"


def test_packed_generator_equals_the_character_generator():
    text, off, lens = host.synth_reference(6, [300_000, 40_000], threads=4)
    for L, n in ((100, 10_000), (150, 5_001), (101, 3_000)):
        bases, _ = host.synth_reads(11, text, off, lens, n, L, 3, first_read=17, threads=3)
        bpr = (L + 3) // 4
        out = np.full(((n * bpr + 7) & ~7) + 8, 0xAA, np.uint8)
        host.synth_reads_packed(11, text, off, lens, n, L, 3, out, first_read=17, threads=4)
        c = out[:n * bpr].reshape(n, bpr)
        q = np.stack([(c >> (2 * j)) & 3 for j in range(4)], axis=2).reshape(n, bpr * 4)
        assert np.array_equal(util.ACGT[q[:, :L]].reshape(-1), bases[:n * L])
        assert not q[:, L:].any() and not out[n * bpr:(n * bpr + 7) & ~7].any() and out[-1] == 0xAA


def test_reads_taken_by_reference_from_the_files_mapping(tmp_path):
    # fem_seqfile_fill_packed_refs (what `FEM map` runs with FEM_HOST_FORMAT=1): name, bases and qualities of every read are
    # where the pointers say, in the file's mapping; the codes unpack to the same bases.  Where that cannot be — a gzip file
    # (2), more characters outside ACGT than the batch may carry (1) — the plan is left alone and fem_seqfile_fill takes it.
    rng = np.random.default_rng(303)
    L, n = 75, 4000
    recs = []
    for i in range(n):
        seq = bytearray(util.rand_seq(rng, L))
        if i % 9 == 0:
            seq[int(rng.integers(0, L))] = ord("N")
        qual = bytes(rng.integers(33, 74, size=L).astype(np.uint8))
        recs.append((b"r%d/1 x" % i, bytes(seq), qual))
    fq = tmp_path / "refs.fq"
    fq.write_bytes(b"".join(b"@" + nm + b"\n" + sq + b"\n+\n" + ql + b"\n" for nm, sq, ql in recs))
    want = [(nm.split(b" ")[0], sq, ql) for nm, sq, ql in recs]
    for approx, threads in ((0, 3), (100_000, 4)):
        got = host.read_refs_batches(str(fq), approx, threads=threads)
        assert all(rc == 0 for rc, _ in got)
        flat = [r for _, rs in got for r in rs]
        assert [(nm, sq, ql) for nm, sq, ql, _ in flat] == want
        assert all(sq == un for _, sq, _, un in flat)  # the 2-bit codes + exceptions rebuild the same bases
    # 1: no room for the batch's N's — the same plan goes through fem_seqfile_fill
    got = host.read_refs_batches(str(fq), 0, threads=2, exc_cap=3)
    assert [rc for rc, _ in got] == [1] and [r[:3] for r in got[0][1]] == want
    # 2: a gzip file's windows are reused: no pointers into them
    gzp = tmp_path / "refs.fq.gz"
    with gzip.open(str(gzp), "wb", compresslevel=1) as f:
        f.write(fq.read_bytes())
    got = host.read_refs_batches(str(gzp), 1 << 18, threads=3)
    assert all(rc == 2 for rc, _ in got) and [r[:3] for _, rs in got for r in rs] == want


def test_qualities_filled_into_the_text_the_device_left_open():
    """fem_sam_fill_quals (fem_host.cc): the host half of fem_dev_commit_names_stage — read r's quality string goes to
    text + qual_at[r]; UINT64_MAX = the read has no record; a field that would end behind the text is refused."""
    import ctypes as C
    from fem_amd import host
    rng = np.random.default_rng(5)
    n = 20_000
    lens = rng.integers(1, 120, n).astype(np.uint64)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    quals = rng.integers(33, 100, int(off[-1])).astype(np.uint8)
    text = np.full(int(off[-1]) * 2 + 64, ord("."), np.uint8)
    qual_at = np.full(n, np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64)
    at = 7
    want = text.copy()
    for r in range(n):
        if r % 5 == 3:
            continue  # (no record)
        qual_at[r] = at
        want[at:at + int(lens[r])] = quals[int(off[r]):int(off[r + 1])]
        at += int(lens[r]) + int(rng.integers(0, 3))
    L = host.lib()
    for threads in (1, 5):
        got = text.copy()
        assert L.fem_sam_fill_quals(got.ctypes.data, len(got), qual_at.ctypes.data, n, quals.ctypes.data, off.ctypes.data, 0, threads) == 0
        assert np.array_equal(got, want)
    # reads of one length, no offset table
    n2, L2 = 5000, 36
    q2 = rng.integers(33, 100, n2 * L2).astype(np.uint8)
    at2 = (np.arange(n2, dtype=np.uint64) * np.uint64(L2 + 3)) + np.uint64(2)
    t2 = np.zeros(n2 * (L2 + 3) + 8, np.uint8)
    assert L.fem_sam_fill_quals(t2.ctypes.data, len(t2), at2.ctypes.data, n2, q2.ctypes.data, None, L2, 4) == 0
    for r in (0, 1, n2 - 1):
        assert np.array_equal(t2[int(at2[r]):int(at2[r]) + L2], q2[r * L2:(r + 1) * L2])
    bad = at2.copy()
    bad[n2 - 1] = len(t2) - 5  # the field would end behind the text
    assert L.fem_sam_fill_quals(t2.ctypes.data, len(t2), bad.ctypes.data, n2, q2.ctypes.data, None, L2, 4) == -1
    assert L.fem_sam_fill_quals(None, 0, bad.ctypes.data, n2, q2.ctypes.data, None, L2, 4) == -1
