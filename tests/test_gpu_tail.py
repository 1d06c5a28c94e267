"""Device mapping tail (fem_dev_fetch_records) vs the CPU oracle's records, field for field.  Needs a GPU: -m gpu."""
import numpy as np
import pytest

from oracle import fem_oracle as fo
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["default", "tiny-staging"])
def dev(request):
    import os
    from fem_amd import Device
    # "tiny-staging": the first-pass CIGAR/MD staging holds one run and two characters, so nearly every record
    # goes through the overflow pass of the traceback kernel
    os.environ["FEM_TEST_TINY_BUFFERS"] = "1" if request.param == "tiny-staging" else "0"
    d = Device(0)
    os.environ.pop("FEM_TEST_TINY_BUFFERS")
    yield d
    d.close()


def run_both(dev, seqs, reads, e, a=1):
    ref = fo.Reference(seqs)
    idx = fo.OracleIndex(ref)
    batch = fo.ReadBatch(reads)
    want = fo.map_reads(ref, idx, batch, e=e, a=a)
    dev.upload_reference(seqs)
    dev.upload_index(12, 3, idx.lookup, idx.occ[:idx.n_occ])
    dev.stage_reads(batch.bases, batch.off)
    dev.map_staged(e=e, a=a)
    got = dev.fetch_records()
    return want, got


def assert_same_records(want, got):
    assert np.array_equal(got.rec_begin, want.rec_off), "records per read"
    assert np.array_equal(got.flag, want.r_flag), "FLAG"
    assert np.array_equal(got.tid, want.r_tid), "reference id"
    assert np.array_equal(got.pos0, want.r_pos), "POS"
    assert np.array_equal(got.nm, want.r_nm), "NM"
    assert np.array_equal(got.cigar_off, want.cig_off), "CIGAR lengths"
    assert np.array_equal(got.cigar, want.cig), "CIGAR"
    assert np.array_equal(got.md_off, want.md_off), "MD lengths"
    assert np.array_equal(got.md, want.md), "MD"
    assert np.array_equal(got.stats, want.stats)


@pytest.mark.parametrize("e,L,repeat", [(3, 100, True), (7, 150, True), (2, 80, False), (1, 64, True), (5, 125, True),
                                        (0, 60, False), (3, 100, False)])
def test_records_equal_oracle(dev, e, L, repeat):
    rng = np.random.default_rng(400 + 10 * e + L)
    if repeat:
        seqs = util.repeat_rich_reference(rng, n_seq=3, unit_len=max(300, 2 * L), n_units=3, copies=110, spacer=150)
    else:
        seqs = [util.rand_seq(rng, 150_000), util.rand_seq(rng, 80_000)]
    reads = util.make_reads(rng, seqs, 300, L, e, n_rate=0.002)
    want, got = run_both(dev, seqs, reads, e)
    assert want.stats[4] > 100
    assert_same_records(want, got)
    per_read = np.diff(want.rec_off.astype(np.int64))
    if repeat and e >= 3:
        assert per_read.max() > 64, "fixture must exercise klib's radix path (>64 mappings of one read)"
    if e >= 2:
        assert np.any((want.cig & 0xF) == 1) and np.any((want.cig & 0xF) == 2), "fixture must contain I and D"


@pytest.mark.parametrize("e,L", [(3, 100), (7, 150), (2, 64)])
def test_substitution_only_reads_and_the_diagonal_shortcut(dev, e, L):
    # trace_ident_kernel finishes a record whose ed edits are all mismatches on the end position's diagonal without the
    # recurrence (CIGAR `L M`, MD from the mismatching columns): reads with 0..e substitutions, some at the first and
    # last base, some adjacent, some an N in the read or over an N / lower-case base of the reference, both strands
    rng = np.random.default_rng(9100 + e)
    s0 = bytearray(util.rand_seq(rng, 120_000))
    s0[40_000:40_030] = b"N" * 30
    s0[70_000:70_400] = bytes(s0[70_000:70_400]).lower()
    seqs = [bytes(s0), util.rand_seq(rng, 60_000)]
    nxt = {65: 67, 67: 71, 71: 84, 84: 65}
    reads = []
    for i in range(400):
        sq = seqs[i % 2]
        where = (39_990, 69_990, 70_100)[i % 3] + int(rng.integers(0, 60)) if i % 7 == 0 and sq is seqs[0] else int(rng.integers(0, len(sq) - L))
        r = bytearray(sq[where:where + L].upper())
        k = int(rng.integers(0, e + 1))
        at = set(int(x) for x in rng.integers(0, L, k))
        if i % 5 == 0 and k:
            at = set(list(at)[:max(0, k - 2)]) | ({0, L - 1} if i % 10 == 0 else {L // 2, L // 2 + 1})
        for x in sorted(at)[:e]:
            r[x] = 78 if i % 13 == 0 else nxt.get(r[x], 65)
        r = bytes(r)
        reads.append(fo.revcomp(r) if i % 2 else r)
    want, got = run_both(dev, seqs, reads, e)
    assert want.stats[4] > 300
    n_ops = np.diff(want.cig_off.astype(np.int64))
    assert (n_ops == 1).sum() > 250 and (want.r_nm[n_ops == 1] > 0).sum() > 100, "fixture must hold mismatch-only records"
    assert_same_records(want, got)


def test_thousands_of_mappings_per_read(dev):
    # one unit ~1500 times: reads with far more mappings than the ordering kernel keeps in LDS, several radix levels
    rng = np.random.default_rng(77)
    unit = util.rand_seq(rng, 150)
    parts = [util.rand_seq(rng, 50_000)]
    for _ in range(2500):
        parts.append(util.mutate(rng, unit, int(rng.integers(0, 2))))
        parts.append(util.rand_seq(rng, int(rng.integers(10, 50))))
    seqs = [b"".join(parts), util.rand_seq(rng, 20_000)]
    reads = [util.mutate(rng, unit[s:s + 103], int(rng.integers(0, 3)))[:100] for s in rng.integers(0, 45, size=12)]
    reads = [r if rng.random() < 0.5 else util.revcomp(r) for r in reads]
    reads += util.make_reads(rng, seqs, 30, 100, 3)
    want, got = run_both(dev, seqs, reads, e=3)
    assert np.diff(want.rec_off.astype(np.int64)).max() > 2048
    assert_same_records(want, got)


def test_ragged_lengths_case_and_sequence_ends(dev):
    # lower-case reference stretches and lower-case reads: the traceback compares raw characters (src/align.c:355),
    # so these take its odd paths; reads of many lengths; reads at both ends of a sequence
    rng = np.random.default_rng(78)
    s0 = util.rand_seq(rng, 120_000)
    s0 = s0[:30_000] + s0[30_000:60_000].lower() + s0[60_000:]
    seqs = [s0, util.rand_seq(rng, 3000)]
    reads = []
    for L in (20, 35, 59, 60, 61, 64, 99, 100, 101, 127, 128, 129, 200, 255, 256, 257, 300):
        reads += util.make_reads(rng, seqs, 8, L, 2)
    reads += [r.lower() for r in util.make_reads(rng, seqs, 20, 100, 2)]
    reads += [s0[29_950:30_050], s0[59_950:60_050], s0[40_000:40_100], s0[40_000:40_100].upper()]
    reads += [seqs[1][:100], seqs[1][3:103], seqs[1][-100:], seqs[1][-103:-3]]
    want, got = run_both(dev, seqs, reads, e=2)
    assert_same_records(want, got)


def test_empty_and_unmapped(dev):
    rng = np.random.default_rng(79)
    seqs = [util.rand_seq(rng, 50_000)]
    want, got = run_both(dev, seqs, [], e=3)
    assert got.n_records == 0 and len(got.rec_begin) == 1
    reads = [util.rand_seq(rng, 100) for _ in range(20)]  # random reads: nothing maps
    want, got = run_both(dev, seqs, reads, e=3)
    assert_same_records(want, got)
    assert got.n_records == 0


def test_device_tail_equals_host_tail_at_scale(dev):
    # 300 k reads of the bench workload's shape (C2-like: 5 Mbp, 100 bp, e=3): too many for the oracle's traceback in
    # a test, so the device records are compared with libfemhost's mapping tail — a second, independent
    # implementation that the CPU suite pins against the oracle (tests/test_host.py)
    from fem_amd import host
    n = 300_000
    text, off, lens = host.synth_reference(2, [5_000_000], threads=8)
    bases, offs = host.synth_reads(2, text, off, lens, n, 100, 3, threads=8)
    dev.upload_reference([text[:5_000_000]])
    dev.build_index(12, 3, fetch=False)
    dev.stage_reads(bases, offs)
    dev.map_staged(e=3)
    res = dev.fetch()
    got = dev.fetch_records()
    tref = host.TailReference(text, off, lens)
    want = host.tail_records(3, tref, bases, offs, res.cand_begin, res.cand_count, res.cand, res.ed, res.end, threads=8)
    assert got.n_records == int(res.stats[4]) and got.n_records > 200_000
    assert np.array_equal(got.rec_begin, want.rec_off)
    assert np.array_equal(got.flag, want.flag) and np.array_equal(got.tid, want.tid)
    assert np.array_equal(got.pos0, want.pos0) and np.array_equal(got.nm, want.nm)
    assert np.array_equal(got.cigar_off, want.cigar_off) and np.array_equal(got.cigar, want.cigar)
    assert np.array_equal(got.md_off, want.md_off) and np.array_equal(got.md, want.md)
    # size-independent properties: read bases consumed by every CIGAR, NM within the threshold
    ops, lens_ = got.cigar & 0xF, got.cigar >> 4
    consumed = np.add.reduceat(np.where(ops != 2, lens_, 0), got.cigar_off[:-1].astype(np.int64))
    assert np.mean(consumed == 100) > 0.999  # the 'S' fold can move read-end bases into a deletion (src/align.c:466-469)
    assert got.nm.max() <= 3
