"""BASELINE configs C3 and C5 at their FULL reference size against the oracle, bit for bit: 24 x 125 Mbp, 999 999 912
index entries (the oracle's index built by 16 threads, ~30 s), 100 k reads of 100 bp at e=3 and 50 k reads of 150 bp at
e=7 through seed_select_kernel + seed_join_kernel as the library selects them.  (BASELINE's 50 M reads per config go through the same kernels
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
    assert dev.seed_kernel(e=3) == "seed_join_kernel" and dev.seed_kernel(e=7) == "seed_join_kernel"


@pytest.mark.parametrize("e,a,L,n,seed", [(3, 1, 100, 100_000, 3), (7, 1, 150, 50_000, 5), (5, 2, 125, 20_000, 11)])
def test_c3_c5_candidates_and_verification_equal_the_oracle(full, e, a, L, n, seed):
    text, off, lens, ref, idx, dev = full
    if dev.seed_kernel(e=e) != "seed_join_kernel":
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
    if dev.seed_kernel(e=3) != "seed_join_kernel":
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


def test_diagonals_around_the_second_probe_boundaries(full):
    # The join's second bitmap probe shifts a value's slot by its bits above 17 and keeps values within e of a 2^17
    # boundary unseen (fem_seed_dense.hip.h).  Reads with a 2-base deletion / insertion in the middle have their seeds
    # on two diagonals two apart; here those diagonals sit at every offset -9..9 around multiples of 2^17 of the global
    # coordinate, in three sequences, on both strands, among ordinary reads (so that the lists are long and the probe runs).
    from tests import util
    text, off, lens, ref, idx, dev = full
    if dev.seed_kernel(e=3) != "seed_join_kernel":
        dev.build_index(12, 3, fetch=False)
    gap = 2048  # femk::kDenseGap: goff[s] = gap + s * (len + gap)
    special = []
    for s in (0, 5, 23):
        goff = gap + s * (125_000_000 + gap)
        seq = text[int(off[s]):int(off[s]) + int(lens[s])]
        for i in range(5):
            g0 = ((goff + 7_000_000 + 11_111_111 * i) // 131072 + 1) * 131072
            for delta in range(-9, 10):
                pos = g0 + delta - goff
                for shift in (0, 30):  # the seed's offset in the read moves the diagonal: v = G - start
                    p0 = pos - shift
                    dele = seq[p0:p0 + 48].tobytes() + seq[p0 + 50:p0 + 102].tobytes()
                    ins = seq[p0:p0 + 48].tobytes() + b"GT" + seq[p0 + 48:p0 + 98].tobytes()
                    special += [dele, util.revcomp(dele), ins, util.revcomp(ins)]
    n = 20_000
    bases, offs = host.synth_reads(77, text, off, lens, n, 100, 3, threads=16)
    reads = [bases[i * 100:(i + 1) * 100].tobytes() for i in range(n)] + special
    batch = fo.ReadBatch(reads)
    want = fo.map_reads(ref, idx, batch, e=3, threads=16, stages=fo.STAGE_SEED | fo.STAGE_VERIFY)
    got = dev.map_batch(batch.bases, batch.off, e=3)
    o, cand, ed, end = got.per_strand()
    assert np.array_equal(got.stats, want.stats), (got.stats, want.stats)
    assert np.array_equal(o, want.cand_off) and np.array_equal(cand, want.cands) and np.array_equal(ed, want.v_ed)
    # nearly all of the special reads map (2 edits <= e)
    per_read = np.add.reduceat(np.diff(want.cand_off.astype(np.int64)), np.arange(0, 2 * len(reads), 2))
    assert (per_read[n:] > 0).mean() > 0.95
