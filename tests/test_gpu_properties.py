"""Size-independent properties of the device path at BASELINE.json's sizes (no oracle needed at this scale),
plus a statistical cross-check against the SURVEY's probe of the true reference binary.  Needs a GPU: -m gpu."""
import numpy as np
import pytest

from fem_amd import host

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from fem_amd import Device
    d = Device(0)
    yield d
    d.close()


@pytest.fixture(scope="module")
def c2(dev):
    # BASELINE.json configs[1] shape: 5 Mbp reference, 100 bp reads, e=3 (2 M reads here; bench.py runs the 10 M)
    text, off, lens = host.synth_reference(2, [5_000_000], threads=16)
    dev.upload_reference([text[:5_000_000]])
    dev.build_index(12, 3, fetch=False)
    n = 2_000_000
    bases, offs = host.synth_reads(2, text, off, lens, n, 100, 3, threads=16)
    return text, off, lens, bases, offs, n


def test_counters_are_consistent_with_the_result_arrays(dev, c2):
    text, off, lens, bases, offs, n = c2
    r = dev.map_batch(bases, offs, e=3)
    o, cand, ed, end = r.per_strand()
    assert int(r.stats[0]) == n
    assert int(r.stats[3]) == len(cand) == int(r.cand_count.sum())
    assert int(r.stats[4]) == int(np.count_nonzero(ed != 0xFF))
    acc = np.add.reduceat(np.concatenate([(ed != 0xFF).astype(np.int64), [0]]), o[:-1].astype(np.int64))
    acc[np.diff(o.astype(np.int64)) == 0] = 0
    per_read = acc[0::2] + acc[1::2]
    assert int(r.stats[1]) == int(np.count_nonzero(per_read))
    assert np.all(ed[ed != 0xFF] <= 3)
    assert np.all((end[ed != 0xFF] >= 99) & (end[ed != 0xFF] <= 99 + 6))  # end offset in [L-1, L-1+2e]
    # candidates of a strand are strictly ascending and more than e apart (staged greedy, src/filter.c:45-78)
    d = np.diff(cand.astype(np.int64))
    inner = np.ones(len(cand) - 1, bool)
    inner[(o[1:-1] - 1).astype(np.int64)[(o[1:-1] > 0) & (o[1:-1] < len(cand))]] = False
    assert np.all(d[inner] > 3)
    assert int(r.stats[1]) > 0.8 * n  # reads come from the reference with <= e edits: most of them map


def test_sharding_and_repetition_do_not_change_results(dev, c2):
    text, off, lens, bases, offs, n = c2
    whole = dev.map_batch(bases, offs, e=3)
    again = dev.map_batch(bases, offs, e=3, slot=1)
    wo, wc, we, wn = whole.per_strand()
    ao, ac, ae, an = again.per_strand()
    assert np.array_equal(wo, ao) and np.array_equal(wc, ac) and np.array_equal(we, ae) and np.array_equal(wn, an)
    h = n // 2 + 7
    lo = dev.map_batch(bases[:h * 100 + 8], offs[:h + 1], e=3)
    hi = dev.map_batch(bases[h * 100:], offs[h:] - offs[h], e=3)
    assert np.array_equal(lo.stats + hi.stats, whole.stats)
    lo_o, lo_c, lo_e, lo_n = lo.per_strand()
    hi_o, hi_c, hi_e, hi_n = hi.per_strand()
    assert np.array_equal(np.concatenate([lo_c, hi_c]), wc)
    assert np.array_equal(np.concatenate([lo_e, hi_e]), we)


def test_reverse_complementing_the_reads_swaps_the_strands(dev, c2):
    text, off, lens, bases, offs, n = c2
    m = 200_000
    fwd = bases[:m * 100].reshape(m, 100)
    comp = np.zeros(256, np.uint8)
    comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    rc = comp[fwd[:, ::-1]].reshape(-1)
    a = dev.map_batch(np.concatenate([fwd.reshape(-1), np.zeros(8, np.uint8)]), offs[:m + 1], e=3)
    b = dev.map_batch(np.concatenate([rc, np.zeros(8, np.uint8)]), offs[:m + 1], e=3)
    assert np.array_equal(a.stats, b.stats)
    ao, ac, ae, an = a.per_strand()
    bo, bc, be, bn = b.per_strand()
    ca, cb = np.diff(ao.astype(np.int64)), np.diff(bo.astype(np.int64))
    assert np.array_equal(ca[0::2], cb[1::2]) and np.array_equal(ca[1::2], cb[0::2])


def test_device_index_equals_uploaded_index_results(dev, c2):
    text, off, lens, bases, offs, n = c2
    m = 300_000
    a = dev.map_batch(bases[:m * 100 + 8], offs[:m + 1], e=3)
    n_occ, lookup, occ = dev.build_index(12, 3, fetch=True)
    dev.upload_index(12, 3, lookup, occ)
    b = dev.map_batch(bases[:m * 100 + 8], offs[:m + 1], e=3)
    assert np.array_equal(a.stats, b.stats)
    assert np.array_equal(a.per_strand()[1], b.per_strand()[1])


def test_hg19_sized_reference_matches_the_survey_probe_statistics(dev):
    # SURVEY.md / BASELINE.md §2c, measured with the UNMODIFIED reference binary on the same input distribution
    # (24 x 125 Mbp iid reference, 100 bp reads, 0..3 edits, e=3): P/N = 1 534, C/N = 0.985, 98.4 % of reads mapped.
    # The probe's generator and seed are gone, so this is a statistical pin, not a bit-exact one.
    text, off, lens = host.synth_reference(3, [125_000_000] * 24, threads=16)
    dev.upload_reference([text[int(o):int(o) + int(l)] for o, l in zip(off, lens)])
    n_occ, _, _ = dev.build_index(12, 3, fetch=False)
    assert n_occ == 999_999_912  # index entries of the 3 Gbp / 24-sequence reference
    n = 400_000
    bases, offs = host.synth_reads(3, text, off, lens, n, 100, 3, threads=16)
    r = dev.map_batch(bases, offs, e=3)
    N, mapped, P, C, M = [int(x) for x in r.stats]
    assert abs(P / N - 1534) < 0.01 * 1534
    assert abs(C / N - 0.985) < 0.01
    assert abs(mapped / N - 0.984) < 0.01
    assert M >= mapped
