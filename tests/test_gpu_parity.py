"""HIP path vs the CPU oracle, bit for bit, through the C ABI (libfemhip.so).  Needs a GPU: -m gpu."""
import numpy as np
import pytest

from oracle import fem_oracle as fo
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["fast+generic", "hash+generic", "dense+generic", "dense-compact+generic", "generic-only", "tiny-buffers",
                                        "tiny-buffers+hash", "tiny-buffers+dense"])
def dev(request):
    # The library picks the seed kernel by itself: seed_fast_kernel<R, false> (lists in lanes) for sparse
    # indexes, <R, true> (64-bit hash join) for denser ones, seed_select_kernel<R> + seed_join_kernel<R> (11-mer frequency pairs, 32-bit coordinates, bitmap join) for
    # long lists, and the generic kernel for whatever those queue.  The environment hooks (read by fem_dev_open)
    # force the other forms so that every fixture runs through all four.
    import os
    from fem_amd import Device
    os.environ["FEM_FORCE_GENERIC"] = "1" if request.param == "generic-only" else "0"
    os.environ["FEM_FORCE_HASH"] = "1" if "hash" in request.param else "0"
    os.environ["FEM_FORCE_DENSE"] = "1" if "dense" in request.param else "0"
    # "dense-compact": the compact 32-bit occurrence table (lists at lookup[h]) instead of the strided one (lists at h << 7)
    os.environ["FEM_NO_STRIDED"] = "1" if "compact" in request.param else "0"
    # "tiny-buffers": candidate arrays, slow-read queue and arena start far too small, so every fixture goes through
    # the overflow flags, the growth of the buffers and the re-run of the batch in fem_dev_sync
    os.environ["FEM_TEST_TINY_BUFFERS"] = "1" if "tiny" in request.param else "0"
    d = Device(0)
    os.environ.pop("FEM_FORCE_GENERIC")
    os.environ.pop("FEM_FORCE_HASH")
    os.environ.pop("FEM_FORCE_DENSE")
    os.environ.pop("FEM_NO_STRIDED")
    os.environ.pop("FEM_TEST_TINY_BUFFERS")
    yield d
    d.close()


def run_both(dev, seqs, reads, e, a=1, build_on_device=False):
    ref = fo.Reference(seqs)
    idx = fo.OracleIndex(ref)
    batch = fo.ReadBatch(reads)
    want = fo.map_reads(ref, idx, batch, e=e, a=a, stages=fo.STAGE_SEED | fo.STAGE_VERIFY)
    dev.upload_reference(seqs)
    if build_on_device:
        n, lookup, occ = dev.build_index(12, 3)
        assert n == idx.n_occ
        assert np.array_equal(lookup, idx.lookup)
        assert np.array_equal(occ, idx.occ[:n])
    else:
        dev.upload_index(12, 3, idx.lookup, idx.occ[:idx.n_occ])
    got = dev.map_batch(batch.bases, batch.off, e=e, a=a)
    return want, got


def assert_same(want, got):
    off, cand, ed, end = got.per_strand()
    assert np.array_equal(off, want.cand_off), "candidate counts per (read, strand)"
    assert np.array_equal(cand, want.cands), "candidate locations"
    assert np.array_equal(ed, want.v_ed), "edit distances / accept set"
    assert np.array_equal(end[ed != 0xFF], want.v_end[want.v_ed != 0xFF]), "end offsets"
    assert np.array_equal(got.stats, want.stats), (got.stats, want.stats)


def test_config1_random_reference(dev):
    # BASELINE.json configs[0]: 1k synthetic 100 bp reads, e=3, 1 Mbp random reference, k=12 step=3
    rng = np.random.default_rng(1)
    seqs = [util.rand_seq(rng, 1_000_000)]
    reads = util.make_reads(rng, seqs, 1000, 100, 3)
    want, got = run_both(dev, seqs, reads, e=3, build_on_device=True)
    assert want.stats[1] > 700
    assert_same(want, got)


@pytest.mark.parametrize("e,a,L", [(3, 1, 100), (7, 1, 150), (2, 1, 75), (0, 1, 64), (3, 2, 100), (3, 0, 100), (5, 1, 125)])
def test_repeat_rich_multi_sequence(dev, e, a, L):
    # many candidates per strand: full groups of 8 (16-bit Myers lanes), LDS overflow into the arena, N runs
    rng = np.random.default_rng(10 * e + a + L)
    seqs = util.repeat_rich_reference(rng, n_seq=3, unit_len=max(300, 2 * L), n_units=6, copies=60, spacer=200)
    reads = util.make_reads(rng, seqs, 400, L, e, n_rate=0.003)
    want, got = run_both(dev, seqs, reads, e=e, a=a, build_on_device=(e == 3 and a == 1))
    assert_same(want, got)
    if e > 0:
        per_strand = np.diff(want.cand_off.astype(np.int64))
        assert per_strand.max() >= 8, "fixture must reach the 8-lane path"


def test_ragged_lengths_and_degenerate_reads(dev):
    rng = np.random.default_rng(5)
    seqs = [util.rand_seq(rng, 200_000), util.rand_seq(rng, 3000)]
    reads = []
    for L in (12, 20, 35, 59, 60, 61, 64, 99, 100, 101, 127, 128, 129, 200, 255, 256, 257, 300):
        reads += util.make_reads(rng, seqs, 6, L, 2)
    reads += [b"N" * 100, b"A" * 100, b"ACGT" * 25, b"acgt" * 25, reads[40].lower(), b"ACGTNNNN" * 12 + b"ACGT"]
    # reads hanging over both ends of a sequence (range clip, src/filter.c:133-144)
    reads += [seqs[1][:100], seqs[1][1:101], seqs[1][3:103], seqs[1][-100:], seqs[1][-103:-3], seqs[1][-104:-4]]
    want, got = run_both(dev, seqs, reads, e=2)
    assert_same(want, got)


def test_empty_batch(dev):
    rng = np.random.default_rng(6)
    seqs = [util.rand_seq(rng, 5000)]
    want, got = run_both(dev, seqs, [], e=3)
    assert got.n_reads == 0 and len(got.cand) == 0
    assert np.array_equal(got.stats, np.zeros(5, np.uint64))


def test_poly_a_and_n_runs(dev):
    # one k-mer with a huge bucket: poly-A and N runs hash to 0 (src/utils.h:92).  With that bucket holding most
    # of the index the DP runs into its +inf column (src/filter.c:9); both sides then pick the same (zeroed) seeds.
    rng = np.random.default_rng(8)
    seqs = [util.rand_seq(rng, 3000) + b"A" * 9000 + util.rand_seq(rng, 3000) + b"N" * 3000 + util.rand_seq(rng, 2000)]
    reads = [b"A" * 100, b"A" * 50 + util.rand_seq(rng, 50), seqs[0][2950:3050], seqs[0][11990:12090]]
    reads += util.make_reads(rng, seqs, 50, 100, 3)
    want, got = run_both(dev, seqs, reads, e=3)
    assert_same(want, got)


def test_high_copy_repeat_takes_the_arena_path(dev):
    # a 150 bp unit present ~1500 times: thousands of staged occurrences and candidates per strand, far beyond
    # the LDS staging capacity -> the same code runs over the global arena (and the buffers grow + re-run)
    rng = np.random.default_rng(9)
    unit = util.rand_seq(rng, 150)
    parts = [util.rand_seq(rng, 100_000)]
    for _ in range(1500):
        parts.append(util.mutate(rng, unit, int(rng.integers(0, 2))))
        parts.append(util.rand_seq(rng, int(rng.integers(10, 50))))
    seqs = [b"".join(parts), util.rand_seq(rng, 50_000)]
    reads = [util.mutate(rng, unit[s:s + 103], int(rng.integers(0, 4)))[:100] for s in rng.integers(0, 45, size=24)]
    reads = [r if rng.random() < 0.5 else util.revcomp(r) for r in reads]
    reads += util.make_reads(rng, seqs, 40, 100, 3)
    want, got = run_both(dev, seqs, reads, e=3)
    assert np.diff(want.cand_off.astype(np.int64)).max() > 600
    assert want.pre.max() > 3 * 512
    assert_same(want, got)


def test_short_tandem_repeats_survive_in_several_phase_groups(dev):
    # units of 4, 5 and 7 bases: a read inside such a stretch matches at shifts that are no multiple of the step, so
    # its survivors sit in two or three phase groups, a few bases apart — few occurrences (a "small" read for the
    # batched path of the fast kernel) but more than one candidate to merge: those reads are handed to the generic kernel
    rng = np.random.default_rng(12)
    parts = []
    for unit_len in (4, 5, 7, 4, 5, 7, 11):
        parts.append(util.rand_seq(rng, int(rng.integers(3000, 6000))))
        unit = util.rand_seq(rng, unit_len)
        parts.append(unit * (260 // unit_len))
    parts.append(util.rand_seq(rng, 4000))
    seqs = [b"".join(parts)]
    starts, at = [], 0
    for i, part in enumerate(parts):
        if i % 2 == 1:
            starts.append((at, len(part)))
        at += len(part)
    reads = []
    for s, n in starts:  # reads inside, and straddling the borders of, every repeat stretch
        for off in (-60, -20, 0, 7, 31, n - 100, n - 80, n - 40):
            r = seqs[0][s + off:s + off + 100]
            reads += [r, util.revcomp(r), util.mutate(rng, r + seqs[0][s + off + 100:s + off + 103], 2)[:100]]
    reads += util.make_reads(rng, seqs, 200, 100, 3)
    want, got = run_both(dev, seqs, reads, e=3)
    assert_same(want, got)
    per_strand = np.diff(want.cand_off.astype(np.int64))
    assert (per_strand >= 2).sum() > 10, "fixture must contain reads with several candidates per strand"


@pytest.mark.parametrize("e,a,lengths", [
    (0, 1, (64, 65, 66)), (1, 1, (50, 75, 100)), (2, 0, (99, 100, 101)), (3, 1, (100,)), (3, 1, (70, 100, 130, 180, 250)),
    (4, 1, (110, 111, 112, 113)), (5, 2, (150, 151)), (6, 1, (140, 200)), (7, 2, (180, 256)), (3, 1, (240, 256, 257, 300, 90)),
])
def test_sparse_index_sweep_over_lengths_and_errors(dev, e, a, lengths):
    # sparse index (every third position of a random reference): the block-staged form of the fast seed kernel.  Reads
    # of mixed lengths in random order, so that blocks of 16 reads straddle every length (and the 256-base staging
    # limit in the last case: those blocks go to the generic kernel), with and without N.
    rng = np.random.default_rng(1000 + 31 * e + a + sum(lengths))
    seqs = [util.rand_seq(rng, 400_000), util.rand_seq(rng, 30_000)]
    reads = []
    for L in lengths:
        reads += util.make_reads(rng, seqs, 120, L, e)
        reads += util.make_reads(rng, seqs, 40, L, e, n_rate=0.01)
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    want, got = run_both(dev, seqs, reads, e=e, a=a)
    assert want.stats[1] > len(reads) // 3
    assert_same(want, got)


@pytest.mark.parametrize("n_reads", [1, 15, 16, 17, 63, 64, 65, 129])
def test_batch_sizes_around_the_block_and_pull_sizes(dev, n_reads):
    # the fast seed kernel takes reads in blocks of 16, pulled 64 at a time from a cursor: batches that end inside a
    # block, at a block's end, and inside / at the end of a pull
    rng = np.random.default_rng(7000 + n_reads)
    seqs = [util.rand_seq(rng, 150_000)]
    reads = util.make_reads(rng, seqs, n_reads, 100, 3)
    want, got = run_both(dev, seqs, reads, e=3)
    assert_same(want, got)


def test_many_short_sequences(dev):
    # 150 sequences: more than the 64 whose coordinates the dense kernel keeps in LDS (its clip then goes through the
    # block table in HBM), reads starting at the first and ending at the last base of many of them, and occurrence
    # entries within 1 024 bases of a sequence start (the remapped entries of the 32-bit table) everywhere
    rng = np.random.default_rng(150)
    seqs = [util.rand_seq(rng, int(rng.integers(400, 4000))) for _ in range(150)]
    reads = util.make_reads(rng, seqs, 600, 100, 3)
    for s in seqs[::7]:
        reads += [s[:100], s[-100:], util.revcomp(s[1:101]), s[5:105], s[-105:-5]]
    want, got = run_both(dev, seqs, reads, e=3)
    assert want.stats[1] > 500
    assert_same(want, got)


def test_long_lists_without_repeats(dev):
    # occurrence lists of 65..128 entries whose positions are unrelated (every 12-mer of twenty loci planted ~100 times at
    # random indexed positions elsewhere): the dense kernel's packed overflow chunk, its wave-reduced maximum of U and
    # the second bitmap probe, with survivors at the true locus only — a 3 Gbp reference in miniature
    rng = np.random.default_rng(128)
    ref = bytearray(util.rand_seq(rng, 20_000_000))
    starts = 1000 + 3000 * np.arange(20)
    lo_plant = (int(starts[-1]) + 5000) // 3  # the loci themselves stay untouched
    for s in starts:
        read = bytes(ref[int(s):int(s) + 100])
        for off in range(89):
            kmer = read[off:off + 12]
            for pos in rng.integers(lo_plant, (len(ref) - 20) // 3, size=int(rng.integers(80, 150))) * 3:
                ref[int(pos):int(pos) + 12] = kmer
    seqs = [bytes(ref)]
    reads = []
    for s in starts:
        for n_err in (0, 1, 3):
            r = util.mutate(rng, seqs[0][int(s):int(s) + 103], n_err)[:100]
            reads.append(r if rng.random() < 0.5 else util.revcomp(r))
    reads += util.make_reads(rng, seqs, 60, 100, 3)
    want, got = run_both(dev, seqs, reads, e=3)
    assert want.pre.max() > 1000 and np.sort(want.pre)[-40] > 400, "lists must be long"
    assert_same(want, got)
