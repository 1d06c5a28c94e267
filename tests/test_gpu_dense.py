"""The seed kernels the library selects BY ITSELF on naturally dense indexes, against the oracle (no FEM_FORCE_* set).

BASELINE configs C3-C5 (3 Gbp, ~60 entries per 12-mer bucket) run seed_select_kernel<R> + seed_join_kernel<R>; references between 50 and
200 Mbp (1-4 entries per bucket) run the 64-bit hash-join form seed_fast_kernel<R, true>.  Both are compared here bit
for bit with the oracle on references large enough that the library picks them unprompted
(reference path: src/filter.c:80-131,146-223).  Needs a GPU: -m gpu.
"""
import os

import numpy as np
import pytest

from oracle import fem_oracle as fo

pytestmark = pytest.mark.gpu


def _setup(seed, seq_lens):
    from fem_amd import Device, host
    for k in ("FEM_FORCE_GENERIC", "FEM_FORCE_HASH", "FEM_FORCE_DENSE", "FEM_NO_DENSE", "FEM_TEST_TINY_BUFFERS"):
        assert os.environ.get(k, "0") == "0", k
    text, off, lens = host.synth_reference(seed, seq_lens, threads=8)
    seqs = [text[int(o):int(o) + int(l)] for o, l in zip(off, lens)]
    ref = fo.Reference([s.tobytes() for s in seqs])
    idx = fo.OracleIndex(ref)
    dev = Device(0)
    dev.upload_reference(seqs)
    n, lookup, occ = dev.build_index(12, 3)  # also checks the device index build at this size, byte for byte
    assert n == idx.n_occ and np.array_equal(lookup, idx.lookup) and np.array_equal(occ, idx.occ[:n])
    return dict(text=text, off=off, lens=lens, ref=ref, idx=idx, dev=dev)


@pytest.fixture(scope="module")
def mid():  # 3 x 25 Mbp: 25 M entries in 16.8 M buckets -> seed_fast_kernel<R, true>
    d = _setup(71, [25_000_000] * 3)
    yield d
    d["dev"].close()


@pytest.fixture(scope="module")
def dense():  # 3 x 72 Mbp: 72 M entries, 4.3 per bucket -> seed_select_kernel<R> + seed_join_kernel<R>
    d = _setup(72, [72_000_000] * 3)
    yield d
    d["dev"].close()


def _compare(d, seed, n_reads, L, e, a=1, extra=()):
    from fem_amd import host
    bases, offsets = host.synth_reads(seed, d["text"], d["off"], d["lens"], n_reads, L, e, threads=8)
    reads = [bases[int(offsets[i]):int(offsets[i + 1])].tobytes() for i in range(n_reads)] + list(extra)
    batch = fo.ReadBatch(reads)
    want = fo.map_reads(d["ref"], d["idx"], batch, e=e, a=a, stages=fo.STAGE_SEED | fo.STAGE_VERIFY)
    d["dev"].set_timing(True)
    d["dev"].reset_timing()
    got = d["dev"].map_batch(batch.bases, batch.off, e=e, a=a)
    d["dev"].set_timing(False)
    off, cand, ed, end = got.per_strand()
    assert np.array_equal(off, want.cand_off), "candidate counts per (read, strand)"
    assert np.array_equal(cand, want.cands), "candidate locations"
    assert np.array_equal(ed, want.v_ed), "edit distances / accept set"
    assert np.array_equal(end[ed != 0xFF], want.v_end[want.v_ed != 0xFF]), "end offsets"
    assert np.array_equal(got.stats, want.stats), (got.stats, want.stats)
    assert want.stats[1] > 0.9 * n_reads
    return got


def _edge_reads(d, L):
    """Reads at the very start and end of every sequence (the remapped near-start entries of the 32-bit table, the
    range clip), forward and reverse."""
    from tests import util
    out = []
    for o, l in zip(d["off"], d["lens"]):
        s = d["text"][int(o):int(o) + int(l)]
        for at in (0, 1, 3, 5, 17, 300, 1000, 1023, 1024, 1030):
            r = s[at:at + L].tobytes()
            out += [r, util.revcomp(r)]
        for at in (int(l) - L, int(l) - L - 2, int(l) - L - 9):
            r = s[at:at + L].tobytes()
            out += [r, util.revcomp(r)]
    return out


@pytest.mark.parametrize("e,L,n", [(3, 100, 4000), (7, 150, 2500), (5, 125, 1500)])
def test_hash_join_form_as_selected(mid, e, L, n):
    _compare(mid, 710 + e, n, L, e, extra=_edge_reads(mid, L))
    ms, launches = mid["dev"].kernel_time(0)
    assert launches >= 1


@pytest.mark.parametrize("e,L,n", [(3, 100, 5000), (7, 150, 2500), (5, 125, 1500), (3, 100, 33), (2, 64, 700),
                                   # DP tables of 33-64 and 65-128 columns (take masks of two and four words per row), and beyond
                                   # what the selection kernel takes (the generic kernel finishes those reads)
                                   (3, 200, 1200), (3, 300, 1200), (1, 400, 600), (5, 330, 700), (3, 470, 500), (2, 600, 300)])
def test_dense_form_as_selected(dense, e, L, n):
    _compare(dense, 720 + e, n, L, e, extra=_edge_reads(dense, L))


def test_dense_form_a2(dense):
    _compare(dense, 729, 1500, 100, 3, a=2)


def test_dense_form_declines_what_its_32_bit_coordinates_cannot_hold():
    # more than 2^18 sequences: the remapped near-start entries (seq << 10 | pos) do not fit; the library must fall back to
    # the 64-bit hash-join form even when asked for the dense one, with identical results
    from fem_amd import Device
    from tests import util
    rng = np.random.default_rng(262)
    n_seq = (1 << 18) + 5
    flat = util.rand_seq(rng, n_seq * 130)
    seqs = [flat[i * 130:(i + 1) * 130] for i in range(n_seq)]
    reads = [seqs[int(i)][int(o):int(o) + 100] for i, o in zip(rng.integers(0, n_seq, 300), rng.integers(0, 27, 300))]
    reads = [util.mutate(rng, r + b"ACG", int(rng.integers(0, 3)))[:100] for r in reads]
    ref = fo.Reference(seqs)
    idx = fo.OracleIndex(ref)
    want = fo.map_reads(ref, idx, fo.ReadBatch(reads), e=3, stages=fo.STAGE_SEED | fo.STAGE_VERIFY)
    os.environ["FEM_FORCE_DENSE"] = "1"
    try:
        dev = Device(0)
    finally:
        os.environ.pop("FEM_FORCE_DENSE")
    try:
        dev.upload_reference(seqs)
        dev.upload_index(12, 3, idx.lookup, idx.occ[:idx.n_occ])
        assert dev.seed_kernel(e=3) != "seed_join_kernel"
        b = fo.ReadBatch(reads)
        got = dev.map_batch(b.bases, b.off, e=3)
        assert np.array_equal(got.stats, want.stats) and np.array_equal(got.per_strand()[1], want.cands)
        assert want.stats[1] > 150
    finally:
        dev.close()


# ---------------------------------------------------------------------------------------------------------------------
# The OVERLAPPED pipeline on a dense index — what bench.py's `value` times: four slots, four batches in flight, batch
# i + 1's seed_select_kernel running beside batch i's seed_join_kernel (launch_batch, fem_hip.hip).  Every batch's
# candidates / edit distances / end offsets / counters (and, in the second form, records) against the oracle, array for
# array (src/map.c:27-55 per read).
# ---------------------------------------------------------------------------------------------------------------------
_SHAPES = [(3, 100), (7, 150), (5, 125), (2, 64), (3, 101), (4, 100)]


def _pipeline_batches(d, seed, n_batches, sizes, with_records, extra_of=None):
    from fem_amd import host
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n_batches):
        e, L = _SHAPES[int(rng.integers(0, len(_SHAPES)))]
        n = int(rng.choice(sizes))
        bases, offsets = host.synth_reads(seed * 1000 + i, d["text"], d["off"], d["lens"], n, L, e, threads=8)
        reads = [bases[int(offsets[j]):int(offsets[j + 1])].tobytes() for j in range(n)]
        if extra_of is not None and i % 5 == 2:
            reads += extra_of(L)
        if i % 7 == 3 and n > 4:  # mixed lengths: characters + offsets instead of the packed transfer
            reads[1] = reads[1][:L - 7]
        if i % 6 == 1 and n:
            reads[0] = b"N" * 3 + reads[0][3:]
        b = fo.ReadBatch(reads)
        stages = fo.STAGE_SEED | fo.STAGE_VERIFY | (fo.STAGE_ALIGN if with_records else 0)
        out.append((b, e, fo.map_reads(d["ref"], d["idx"], b, e=e, threads=8, stages=stages)))
    return out


def _run_overlapped(dev, items, with_records, depth=4, n_slots=4, form="stage_reads"):
    from fem_amd import device

    def check(i):
        b, e, want = items[i]
        s = i % n_slots
        if with_records and i % 2 == 0:  # the device tail of this batch while the next batches' kernels run
            rec = dev.fetch_records(slot=s)
            assert np.array_equal(rec.stats, want.stats), i
            assert np.array_equal(rec.rec_begin, want.rec_off) and np.array_equal(rec.flag, want.r_flag), i
            assert np.array_equal(rec.pos0, want.r_pos) and np.array_equal(rec.nm, want.r_nm) and np.array_equal(rec.tid, want.r_tid), i
            assert np.array_equal(rec.cigar_off, want.cig_off) and np.array_equal(rec.cigar, want.cig), i
            assert np.array_equal(rec.md_off, want.md_off) and np.array_equal(rec.md, want.md), i
        got = dev.fetch(slot=s, copy=False)
        off, cand, ed, end = got.per_strand()
        assert np.array_equal(got.stats, want.stats), (i, got.stats, want.stats)
        assert np.array_equal(off, want.cand_off), i
        assert np.array_equal(cand, want.cands) and np.array_equal(ed, want.v_ed), i
        assert np.array_equal(end[ed != 0xFF], want.v_end[want.v_ed != 0xFF]), i

    for i, (b, e, want) in enumerate(items):
        if i >= depth:
            check(i - depth)
        s = i % n_slots
        n = len(b.off) - 1
        lens = np.diff(b.off.astype(np.int64))
        if form == "packed_commit" and n and lens.min() == lens.max():
            L = int(lens[0])
            hb, _ = dev.acquire_stage(n, n * L, slot=s)
            dev.commit_stage_packed(n, L, device.pack_reads(b.bases, n, L, hb), slot=s)
        else:
            dev.stage_reads(b.bases, b.off, slot=s)
        dev.map_staged(e=e, slot=s)
    for i in range(max(0, len(items) - depth), len(items)):
        check(i)


def test_overlapped_pipeline_on_a_dense_index_equals_the_oracle(dense):
    assert dense["dev"].seed_kernel(e=3) == "seed_join_kernel"
    items = _pipeline_batches(dense, 9101, 28, [0, 1, 900, 20_000, 45_000, 70_000], False, extra_of=lambda L: _edge_reads(dense, L))
    _run_overlapped(dense["dev"], items, False)
    _run_overlapped(dense["dev"], items[:12], False, form="packed_commit")


def test_overlapped_pipeline_with_records_on_a_dense_index_equals_the_oracle(dense):
    items = _pipeline_batches(dense, 9102, 24, [1, 700, 15_000, 30_000], True, extra_of=lambda L: _edge_reads(dense, L))
    _run_overlapped(dense["dev"], items, True)


@pytest.mark.parametrize("banked", [False, True])
def test_overlapped_pipeline_on_a_forced_dense_index(banked):
    # a small repeat-rich reference through the dense kernels (FEM_FORCE_DENSE=1), whole and cut into banks
    # (FEM_TEST_BANK_BASES): long lists, reads the join hands to the generic kernel, many candidates per strand
    from fem_amd import Device
    from tests import util
    rng = np.random.default_rng(515 + banked)
    seqs = util.repeat_rich_reference(rng, n_seq=4, unit_len=300, n_units=10, copies=30)
    bank_bases = int(0.6 * sum(len(s) + 2048 for s in seqs)) if banked else 0  # two banks of two sequences
    ref = fo.Reference(seqs)
    idx = fo.OracleIndex(ref)
    os.environ["FEM_FORCE_DENSE"] = "1"
    if bank_bases:
        os.environ["FEM_TEST_BANK_BASES"] = str(bank_bases)
    try:
        dev = Device(0)
    finally:
        os.environ.pop("FEM_FORCE_DENSE")
        os.environ.pop("FEM_TEST_BANK_BASES", None)
    try:
        dev.upload_reference(seqs)
        dev.upload_index(12, 3, idx.lookup, idx.occ[:idx.n_occ])
        assert dev.seed_kernel(e=3) == ("seed_join_banked_kernel" if bank_bases else "seed_join_kernel")
        items = []
        for i in range(26):
            e, L = _SHAPES[int(rng.integers(0, len(_SHAPES)))]
            n = int(rng.choice([0, 3, 400, 2500, 6000]))
            reads = util.make_reads(rng, seqs, n, L, e, n_rate=0.01 if i % 4 == 0 else 0.0)
            if i % 7 == 3 and n > 4:
                reads[2] = reads[2][:L - 9]
            b = fo.ReadBatch(reads)
            items.append((b, e, fo.map_reads(ref, idx, b, e=e, threads=8)))
        _run_overlapped(dev, items, True)
        _run_overlapped(dev, items[:10], False, form="packed_commit")
    finally:
        dev.close()


def test_a_batch_that_meets_an_idle_device_is_mapped_in_parts_with_the_same_result(dense):
    """launch_batch (fem_hip.hip): a batch committed to an idle device is copied, selected and joined in two parts (the
    pipeline's fill).  Same batch, same slot: in parts (idle device, the default), whole (FEM_NO_PARTS handle), in four
    parts (FEM_PARTS=4) — candidates, verification and counters must be identical; the dense index says what it runs on."""
    from fem_amd import Device, host
    d = dense
    assert "dense: 32-bit occurrence table, compact" in d["dev"].index_info() and "1 bank" in d["dev"].index_info()
    n, L, e = 300_000, 100, 3
    bases, offsets = host.synth_reads(991, d["text"], d["off"], d["lens"], n, L, e, threads=8)
    hb, _ = d["dev"].acquire_stage(n, n * L, slot=2)
    host.synth_reads_packed(991, d["text"], d["off"], d["lens"], n, L, e, hb, first_read=0, threads=8)
    d["dev"].commit_stage_packed(n, L, 0, slot=2)  # nothing in flight on this handle: in parts
    d["dev"].map_staged(e=e, a=1, slot=2)
    in_parts = d["dev"].fetch(slot=2)
    got = d["dev"].map_batch(bases, offsets, e=e, a=1, slot=1)  # (characters through fem_dev_stage_reads: the same reads)
    for x, y in zip(in_parts.per_strand(), got.per_strand()):
        assert np.array_equal(x, y)
    assert np.array_equal(in_parts.stats, got.stats)
    seqs = [d["text"][int(o):int(o) + int(l)] for o, l in zip(d["off"], d["lens"])]
    for env in ({"FEM_NO_PARTS": "1"}, {"FEM_PARTS": "4", "FEM_FORCE_DENSE": "1"}):  # (the second: the padded strided table)
        os.environ.update(env)
        try:
            other = Device(0)
        finally:
            for k in env:
                os.environ.pop(k)
        try:
            other.upload_reference(seqs)
            other.build_index(12, 3, fetch=False)
            assert ("strided with pads" in other.index_info()) == ("FEM_FORCE_DENSE" in env), other.index_info()
            hb2, _ = other.acquire_stage(n, n * L, slot=0)
            hb2[:len(hb)] = hb
            other.commit_stage_packed(n, L, 0, slot=0)
            other.map_staged(e=e, a=1, slot=0)
            r = other.fetch(slot=0)
            for x, y in zip(in_parts.per_strand(), r.per_strand()):
                assert np.array_equal(x, y), env
            assert np.array_equal(in_parts.stats, r.stats), env
        finally:
            other.close()
