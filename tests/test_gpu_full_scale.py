"""BASELINE configs C3 and C5 at their FULL reference size against the oracle, bit for bit: 24 x 125 Mbp, 999 999 912
index entries (the oracle's index built by 16 threads, ~30 s), 100 k reads of 100 bp at e=3 and 50 k reads of 150 bp at
e=7 through seed_dense_kernel as the library selects it.  (BASELINE's 50 M reads per config go through the same kernels
in bench.py and, statistically, in tests/test_gpu_properties.py; the oracle maps ~0.5 Mreads/s at this index size.)
Needs a GPU: -m gpu."""
import numpy as np
import pytest

from fem_amd import host
from oracle import fem_oracle as fo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    from fem_amd import Device
    text, off, lens = host.synth_reference(3, [125_000_000] * 24, threads=16)
    seqs = [text[int(o):int(o) + int(l)] for o, l in zip(off, lens)]
    ref = fo.Reference([s.tobytes() for s in seqs])
    idx = fo.OracleIndex(ref, threads=16)
    dev = Device(0)
    dev.upload_reference(seqs)
    yield text, off, lens, ref, idx, dev
    dev.close()


def test_device_index_is_byte_identical_at_3_gbp(full):
    text, off, lens, ref, idx, dev = full
    n, lookup, occ = dev.build_index(12, 3)  # construct_index (src/index.c:57-98) on the device
    assert n == idx.n_occ == 999_999_912
    assert np.array_equal(lookup, idx.lookup) and np.array_equal(occ, idx.occ[:n])
    assert dev.seed_kernel(e=3) == "seed_dense_kernel" and dev.seed_kernel(e=7) == "seed_dense_kernel"


@pytest.mark.parametrize("e,a,L,n,seed", [(3, 1, 100, 100_000, 3), (7, 1, 150, 50_000, 5), (5, 2, 125, 20_000, 11)])
def test_c3_c5_candidates_and_verification_equal_the_oracle(full, e, a, L, n, seed):
    text, off, lens, ref, idx, dev = full
    if dev.seed_kernel(e=e) != "seed_dense_kernel":
        dev.build_index(12, 3, fetch=False)
    bases, offs = host.synth_reads(seed, text, off, lens, n, L, e, threads=16)
    want = fo.map_reads(ref, idx, fo.ReadBatch.from_arrays(bases, offs), e=e, a=a, threads=16, stages=fo.STAGE_SEED | fo.STAGE_VERIFY)
    got = dev.map_batch(bases, offs, e=e, a=a)
    o, cand, ed, end = got.per_strand()
    assert np.array_equal(got.stats, want.stats), (got.stats, want.stats)
    assert np.array_equal(o, want.cand_off) and np.array_equal(cand, want.cands)
    assert np.array_equal(ed, want.v_ed) and np.array_equal(end[ed != 0xFF], want.v_end[want.v_ed != 0xFF])
    assert want.stats[1] > 0.93 * n


def test_c3_records_equal_the_oracle(full):
    # ... and the mapping tail on the device (ordering, traceback, CIGAR, MD) for reads mapped against the 3 Gbp reference
    text, off, lens, ref, idx, dev = full
    if dev.seed_kernel(e=3) != "seed_dense_kernel":
        dev.build_index(12, 3, fetch=False)
    n = 40_000
    bases, offs = host.synth_reads(33, text, off, lens, n, 100, 3, threads=16)
    want = fo.map_reads(ref, idx, fo.ReadBatch.from_arrays(bases, offs), e=3, threads=16)
    got = dev.map_batch(bases, offs, e=3)
    assert np.array_equal(got.stats, want.stats)
    rec = dev.fetch_records()
    assert np.array_equal(rec.rec_begin, want.rec_off) and np.array_equal(rec.flag, want.r_flag)
    assert np.array_equal(rec.tid, want.r_tid) and np.array_equal(rec.pos0, want.r_pos) and np.array_equal(rec.nm, want.r_nm)
    assert np.array_equal(rec.cigar_off, want.cig_off) and np.array_equal(rec.cigar, want.cig)
    assert np.array_equal(rec.md_off, want.md_off) and np.array_equal(rec.md, want.md)
    assert np.any((want.cig & 0xF) == 1) and np.any((want.cig & 0xF) == 2)
