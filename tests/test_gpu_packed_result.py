"""fem_dev_fetch_packed (include/fem_hip.h): the batch's outcome in the form that crosses the link — one byte per strand,
one offset per 256 strands, the candidates without padding — must say what fem_dev_fetch says, on every kernel path, for
strands with 255 candidates and more (big[]), and once the slot packs and sends home behind its kernels (the second batch
of a slot on).  Needs a GPU: -m gpu."""
import os

import numpy as np
import pytest

from oracle import fem_oracle as fo
from tests import util

pytestmark = pytest.mark.gpu


def _device(env):
    from fem_amd import Device
    for k in env:
        os.environ[k] = "1"
    try:
        return Device(0)
    finally:
        for k in env:
            os.environ.pop(k)


def _same(packed, plain, want=None):
    a, b = packed.per_strand(), plain.per_strand()
    assert np.array_equal(a[0], b[0]), "candidates per strand"
    for x, y in zip(a[1:], b[1:]):
        assert np.array_equal(x, y)
    assert np.array_equal(packed.stats, plain.stats)
    assert packed.n_candidates == int(a[0][-1])  # no padding among the packed candidates
    if want is not None:
        assert np.array_equal(a[0], want.cand_off) and np.array_equal(a[1], want.cands) and np.array_equal(a[2], want.v_ed)


@pytest.mark.parametrize("env", [(), ("FEM_FORCE_DENSE",), ("FEM_FORCE_GENERIC",), ("FEM_FORCE_HASH",), ("FEM_FORCE_DENSE", "FEM_TEST_TINY_BUFFERS")])
def test_packed_result_equals_plain_result(env):
    rng = np.random.default_rng(91 + len(env))
    seqs = util.repeat_rich_reference(rng) + [util.rand_seq(rng, 200_000)]
    ref = fo.Reference(seqs)
    idx = fo.OracleIndex(ref)
    dev = _device(env)
    try:
        dev.upload_reference(seqs)
        dev.upload_index(12, 3, idx.lookup, idx.occ[:idx.n_occ])
        # the same slot again and again: the first packed fetch packs at fetch time, the later ones find the arrays at home
        for i, (L, e, n) in enumerate([(100, 3, 4000), (100, 3, 4100), (150, 7, 900), (64, 2, 3000), (100, 3, 1), (100, 3, 0), (100, 3, 5000)]):
            reads = util.make_reads(rng, seqs, n, L, e, n_rate=0.001) if n else []
            batch = fo.ReadBatch(reads)
            want = fo.map_reads(ref, idx, batch, e=e, stages=fo.STAGE_SEED | fo.STAGE_VERIFY) if n else None
            dev.stage_reads(batch.bases, batch.off, slot=1)
            dev.map_staged(e=e, a=1, slot=1)
            packed = dev.fetch_packed(slot=1)
            if i % 3 == 1:
                packed = dev.fetch_packed(slot=1)  # (twice: nothing is packed or copied again)
            plain = dev.fetch(slot=1) if i != 3 else None  # (batch 3: the slot stays in the packed mode without a plain fetch between)
            if plain is not None:
                _same(packed, plain, want)
                dev.fetch_packed(slot=1)  # back to the packed mode for the next batch
            elif n:
                a = packed.per_strand()
                assert np.array_equal(a[0], want.cand_off) and np.array_equal(a[1], want.cands) and np.array_equal(a[2], want.v_ed)
            assert packed.n_reads == n
    finally:
        dev.close()


def test_strands_with_255_candidates_and_more_are_listed():
    """-a 0: every occurrence of every selected seed is a candidate; on a repeat-rich reference strands pass 255."""
    rng = np.random.default_rng(17)
    seqs = util.repeat_rich_reference(rng, n_seq=2, unit_len=300, n_units=3, copies=400, spacer=40)
    ref = fo.Reference(seqs)
    idx = fo.OracleIndex(ref)
    from fem_amd import Device
    dev = Device(0)
    try:
        dev.upload_reference(seqs)
        dev.upload_index(12, 3, idx.lookup, idx.occ[:idx.n_occ])
        reads = util.make_reads(rng, seqs, 600, 100, 2)
        batch = fo.ReadBatch(reads)
        for a in (0, 1):
            dev.stage_reads(batch.bases, batch.off)
            dev.map_staged(e=2, a=a)
            packed, plain = dev.fetch_packed(), dev.fetch()
            _same(packed, plain)
            if a == 0:
                assert packed.n_big > 0 and int(packed.counts().max()) >= 255
                assert np.all(packed.count[packed.big[:, 0]] == 255)
    finally:
        dev.close()
