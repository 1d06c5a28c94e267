"""A reference beyond the 32-bit coordinate of the dense-index kernels: 30 x 143.4 Mbp = 4.302 Gbp (1 433 999 910 index
entries, 85 per bucket).  The library keeps seed_select_kernel + seed_join_kernel by cutting the sequences into two banks
(16 + 14 sequences: cut about equal, fem_seed_dense.hip.h) instead of falling back to the 64-bit hash-join form; candidates, verification
and records against the oracle, bit for bit.  ~35 GB of host memory, ~1.5 minutes.  Needs a GPU: -m gpu."""
import numpy as np
import pytest

from fem_amd import host
from oracle import fem_oracle as fo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    from fem_amd import Device
    text, off, lens = host.synth_reference(43, [143_400_000] * 30, threads=16)
    seqs = [text[int(o):int(o) + int(l)] for o, l in zip(off, lens)]
    ref = fo.Reference([s.tobytes() for s in seqs])
    idx = fo.OracleIndex(ref, threads=16)
    dev = Device(0)
    dev.upload_reference(seqs)
    n, lookup, occ = dev.build_index(12, 3)
    assert n == idx.n_occ == 1_433_999_910
    assert np.array_equal(lookup, idx.lookup) and np.array_equal(occ, idx.occ[:n])
    yield text, off, lens, ref, idx, dev
    dev.close()


@pytest.mark.parametrize("e,a,L,n,seed", [(3, 1, 100, 60_000, 7), (7, 1, 150, 30_000, 8), (4, 2, 120, 20_000, 9)])
def test_two_banks_equal_the_oracle(big, e, a, L, n, seed):
    text, off, lens, ref, idx, dev = big
    assert dev.seed_kernel(e=e, a=a) == "seed_join_banked_kernel"
    bases, offs = host.synth_reads(seed, text, off, lens, n, L, e, threads=16)
    want = fo.map_reads(ref, idx, fo.ReadBatch.from_arrays(bases, offs), e=e, a=a, threads=16)
    got = dev.map_batch(bases, offs, e=e, a=a)
    o, cand, ed, end = got.per_strand()
    assert np.array_equal(got.stats, want.stats), (got.stats, want.stats)
    assert np.array_equal(o, want.cand_off) and np.array_equal(cand, want.cands)
    assert np.array_equal(ed, want.v_ed) and np.array_equal(end[ed != 0xFF], want.v_end[want.v_ed != 0xFF])
    assert want.stats[1] > 0.9 * n
    in_second = (want.cands >> np.uint64(32)) >= 16  # candidates among the second bank's fourteen sequences
    assert 0.35 * len(want.cands) < in_second.sum() < 0.6 * len(want.cands)
    rec = dev.fetch_records()
    assert np.array_equal(rec.rec_begin, want.rec_off) and np.array_equal(rec.tid, want.r_tid) and np.array_equal(rec.pos0, want.r_pos)
    assert np.array_equal(rec.cigar, want.cig) and np.array_equal(rec.md, want.md)
