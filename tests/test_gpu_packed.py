"""The packed read transfer of fem_dev_stage_reads (include/fem_hip.h): batches of equal-length reads cross the link as
two bits per base for A C G T + position and value of every other byte, and are rebuilt byte for byte on the device.
Results must be those of the character transfer (FEM_NO_PACK=1) and of the oracle, whatever the characters (the traceback
compares characters, src/align.c:289-300: a lower-case read must come out as it does in the reference).
Needs a GPU: -m gpu."""
import os

import numpy as np
import pytest

from oracle import fem_oracle as fo
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from fem_amd import Device
    rng = np.random.default_rng(4242)
    seqs = [util.rand_seq(rng, 300_000), util.rand_seq(rng, 50_000)]
    ref = fo.Reference(seqs)
    idx = fo.OracleIndex(ref)
    dev = Device(0)
    dev.upload_reference(seqs)
    dev.upload_index(12, 3, idx.lookup, idx.occ[:idx.n_occ])
    yield rng, seqs, ref, idx, dev
    dev.close()


def _odd_characters(rng, reads, share):
    """A share of the reads gets lower case, N, IUPAC codes, punctuation and bytes >= 0x80 at random places."""
    odd = [ord(c) for c in "NnRYKM.-*@"] + [0x00, 0x7F, 0x80, 0xC1, 0xFF]
    out = []
    for r in reads:
        r = bytearray(r)
        u = rng.random()
        if u < share / 2:
            for at in rng.integers(0, len(r), int(rng.integers(1, 4))):
                r[int(at)] = odd[int(rng.integers(0, len(odd)))]
        elif u < share * 0.55:  # (a few whole reads in lower case: each byte of theirs is sent as it is)
            r = bytearray(bytes(r).lower())
        out.append(bytes(r))
    return out


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a.per_strand(), b.per_strand())) and np.array_equal(a.stats, b.stats)


@pytest.mark.parametrize("L,e,n", [(100, 3, 3000), (150, 7, 1200), (101, 3, 900), (61, 1, 700), (37, 1, 300), (64, 2, 70000)])
def test_packed_transfer_gives_the_results_of_the_character_transfer_and_the_oracle(setup, L, e, n):
    rng, seqs, ref, idx, dev = setup
    reads = _odd_characters(rng, util.make_reads(rng, seqs, n, L, e), 0.3)
    # first and last characters of the batch and of single reads are the packing's edge cases
    reads[0] = b"N" + reads[0][1:]
    reads[-1] = reads[-1][:-1] + b"n"
    reads[1] = reads[1][:L - 2] + b"NN"
    batch = fo.ReadBatch(reads)
    want = fo.map_reads(ref, idx, batch, e=e, stages=fo.STAGE_SEED | fo.STAGE_VERIFY)
    got = dev.map_batch(batch.bases, batch.off, e=e)
    n_bytes, packed = dev.stage_info()
    assert packed
    n_odd = sum(sum(1 for c in r if c not in b"ACGT") for r in reads)
    code_bytes = (n * ((L + 3) // 4) + 7) // 8 * 8
    assert n_bytes == code_bytes + 5 * n_odd
    os.environ["FEM_NO_PACK"] = "1"
    try:
        plain = dev.map_batch(batch.bases, batch.off, e=e)
        n_plain, packed_plain = dev.stage_info()
    finally:
        del os.environ["FEM_NO_PACK"]
    assert not packed_plain and n_plain == n * L + 8 * (n + 1)
    assert _same(got, plain)
    off, cand, ed, end = got.per_strand()
    assert np.array_equal(got.stats, want.stats)
    assert np.array_equal(off, want.cand_off) and np.array_equal(cand, want.cands) and np.array_equal(ed, want.v_ed)
    assert np.array_equal(end[ed != 0xFF], want.v_end[want.v_ed != 0xFF])
    assert want.stats[1] > 0.5 * n or L < 40  # (37-base reads are too short for three seeds per group: gates only)


def test_records_after_a_packed_transfer(setup):
    # the device tail reads the expanded characters too (traceback, MD): records vs the oracle's
    rng, seqs, ref, idx, dev = setup
    reads = _odd_characters(rng, util.make_reads(rng, seqs, 1500, 100, 3), 0.3)
    batch = fo.ReadBatch(reads)
    dev.stage_reads(batch.bases, batch.off)
    assert dev.stage_info()[1]
    dev.map_staged(e=3)
    got = dev.fetch_records()
    os.environ["FEM_NO_PACK"] = "1"
    try:
        dev.stage_reads(batch.bases, batch.off)
        assert not dev.stage_info()[1]
        dev.map_staged(e=3)
        plain = dev.fetch_records()
    finally:
        del os.environ["FEM_NO_PACK"]
    for f in ("rec_begin", "flag", "tid", "pos0", "nm", "cigar_off", "cigar", "md_off", "md"):
        assert np.array_equal(getattr(got, f), getattr(plain, f)), f
    assert got.n_records > 1000


def test_batches_the_packing_declines(setup):
    rng, seqs, ref, idx, dev = setup
    # reads of different lengths: characters + offsets
    mixed = util.make_reads(rng, seqs, 300, 100, 3) + util.make_reads(rng, seqs, 300, 90, 3)
    b = fo.ReadBatch(mixed)
    got = dev.map_batch(b.bases, b.off, e=3)
    assert not dev.stage_info()[1]
    want = fo.map_reads(ref, idx, b, e=3, stages=fo.STAGE_SEED | fo.STAGE_VERIFY)
    assert np.array_equal(got.stats, want.stats) and np.array_equal(got.per_strand()[1], want.cands)
    # more than one character in sixteen outside ACGT: the list of their positions would outweigh the saving
    noisy = [bytes(bytearray(r[:50]) + b"N" * 50) for r in util.make_reads(rng, seqs, 400, 100, 3)]
    b = fo.ReadBatch(noisy)
    got = dev.map_batch(b.bases, b.off, e=3)
    assert not dev.stage_info()[1]
    want = fo.map_reads(ref, idx, b, e=3, stages=fo.STAGE_SEED | fo.STAGE_VERIFY)
    assert np.array_equal(got.stats, want.stats) and np.array_equal(got.per_strand()[1], want.cands)
    # an empty batch and a single read
    empty = fo.ReadBatch([])
    assert dev.map_batch(empty.bases, empty.off, e=3).stats[0] == 0
    one = fo.ReadBatch([mixed[0]])
    got = dev.map_batch(one.bases, one.off, e=3)
    assert dev.stage_info()[1] and got.stats[0] == 1


@pytest.mark.parametrize("L,e,n", [(100, 3, 3000), (150, 7, 1200), (101, 3, 900), (37, 1, 300), (64, 2, 70000)])
def test_packed_commit_gives_the_results_of_stage_reads_and_the_oracle(setup, L, e, n):
    # fem_dev_commit_stage_packed: the caller writes two bits per base + the exceptions into the pinned staging itself
    # (here numpy's restatement of the layout, fem_amd.device.pack_reads) — no host work in the library
    from fem_amd import device
    rng, seqs, ref, idx, dev = setup
    reads = _odd_characters(rng, util.make_reads(rng, seqs, n, L, e), 0.3)
    reads[0] = b"N" + reads[0][1:]
    reads[-1] = reads[-1][:-1] + b"n"
    batch = fo.ReadBatch(reads)
    want = fo.map_reads(ref, idx, batch, e=e, stages=fo.STAGE_SEED | fo.STAGE_VERIFY)
    hb, _ = dev.acquire_stage(n, n * L, slot=2)
    n_exc = device.pack_reads(batch.bases, n, L, hb)
    assert n_exc == sum(sum(1 for c in r if c not in b"ACGT") for r in reads)
    dev.commit_stage_packed(n, L, n_exc, slot=2)
    n_bytes, packed = dev.stage_info(2)
    bpr, exc_off, exc_cap = device.packed_layout(n, L)
    assert packed and n_bytes == exc_off + 5 * n_exc and bpr == (L + 3) // 4 and exc_off == (n * bpr + 7) // 8 * 8 and exc_cap >= n_exc
    dev.map_staged(e=e, slot=2)
    got = dev.fetch(slot=2)
    ref_run = dev.map_batch(batch.bases, batch.off, e=e, slot=1)
    assert _same(got, ref_run)
    off, cand, ed, end = got.per_strand()
    assert np.array_equal(got.stats, want.stats)
    assert np.array_equal(off, want.cand_off) and np.array_equal(cand, want.cands) and np.array_equal(ed, want.v_ed)
    assert np.array_equal(end[ed != 0xFF], want.v_end[want.v_ed != 0xFF])
    # the same staging committed again without another acquire (the reusable batch), and the device tail behind it
    dev.commit_stage_packed(n, L, n_exc, slot=2)
    dev.map_staged(e=e, slot=2)
    rec = dev.fetch_records(slot=2)
    dev.stage_reads(batch.bases, batch.off, slot=1)
    dev.map_staged(e=e, slot=1)
    rec1 = dev.fetch_records(slot=1)
    for f in ("rec_begin", "flag", "tid", "pos0", "nm", "cigar_off", "cigar", "md_off", "md"):
        assert np.array_equal(getattr(rec, f), getattr(rec1, f)), f


def test_packed_commit_refuses_what_it_cannot_take(setup):
    from fem_amd import FemError, device
    rng, seqs, ref, idx, dev = setup
    n, L = 64, 100
    hb, _ = dev.acquire_stage(n, n * L, slot=3)
    reads = util.make_reads(rng, seqs, n, L, 3)
    n_exc = device.pack_reads(fo.ReadBatch(reads).bases, n, L, hb)
    assert n_exc == 0
    with pytest.raises(FemError):
        dev.commit_stage_packed(n + 1, L, 0, slot=3)          # more reads than acquired
    with pytest.raises(FemError):
        dev.commit_stage_packed(n, 2000, 0, slot=3)           # beyond fem_dev_limits
    with pytest.raises(FemError):
        dev.commit_stage_packed(n, L, n * L // 16 + 1, slot=3)  # more exceptions than the form takes
    bpr, exc_off, _ = device.packed_layout(n, L)
    hb[exc_off:exc_off + 4] = np.array([n * L], np.uint32).view(np.uint8)  # a position outside the batch
    with pytest.raises(FemError):
        dev.commit_stage_packed(n, L, 1, slot=3)
    dev.commit_stage_packed(n, L, 0, slot=3)
    dev.map_staged(e=3, slot=3)
    assert dev.fetch_stats(slot=3)[0] == n
    # an empty batch
    dev.commit_stage_packed(0, L, 0, slot=3)
    dev.map_staged(e=3, slot=3)
    assert dev.fetch_stats(slot=3)[0] == 0
