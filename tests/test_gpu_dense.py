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


@pytest.mark.parametrize("e,L,n", [(3, 100, 5000), (7, 150, 2500), (5, 125, 1500), (3, 100, 33), (2, 64, 700)])
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
