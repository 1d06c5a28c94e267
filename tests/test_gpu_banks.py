"""References in BANKS of sequences (fem_seed_dense.hip.h): what the dense-index kernels do with a reference whose
sequences do not fit one 32-bit coordinate space.  FEM_TEST_BANK_BASES makes the library cut small references the same
way (coordinates per bank), so that the oracle can check it: the selection once, the join per bank on each bank's part of
every list, the rule "the last run keeps values <= max(U) only" (src/filter.c:85) carried across banks, a strand's
candidates of bank after bank handed over as one ascending run.  Needs a GPU: -m gpu."""
import os

import numpy as np
import pytest

from oracle import fem_oracle as fo
from tests import util

pytestmark = pytest.mark.gpu


def _device(bank_bases, bank_seqs=None):
    from fem_amd import Device
    os.environ["FEM_FORCE_DENSE"] = "1"
    os.environ["FEM_TEST_BANK_BASES"] = str(bank_bases)
    if bank_seqs:
        os.environ["FEM_TEST_BANK_SEQS"] = str(bank_seqs)
    try:
        return Device(0)
    finally:
        os.environ.pop("FEM_FORCE_DENSE")
        os.environ.pop("FEM_TEST_BANK_BASES")
        os.environ.pop("FEM_TEST_BANK_SEQS", None)


def _reference(rng, n_seq, shared):
    """Sequences that share repeat units (a list then has entries in several banks) between stretches of their own."""
    units = [util.rand_seq(rng, 260) for _ in range(5)]
    seqs = []
    for s in range(n_seq):
        parts = [util.rand_seq(rng, int(rng.integers(200, 1500)))]
        for _ in range(int(rng.integers(18, 30))):
            if shared and rng.random() < 0.6:
                parts.append(util.mutate(rng, units[int(rng.integers(0, len(units)))], int(rng.integers(0, 3))))
            parts.append(util.rand_seq(rng, int(rng.integers(300, 2500))))
        if s % 3 == 1:
            parts.append(b"N" * 40)
            parts.append(util.rand_seq(rng, 900))
        seqs.append(b"".join(parts))
    return seqs


def _compare(dev, seqs, reads, e, a, banked=True):
    ref = fo.Reference(seqs)
    idx = fo.OracleIndex(ref)
    batch = fo.ReadBatch(reads)
    want = fo.map_reads(ref, idx, batch, e=e, a=a)
    dev.upload_reference(seqs)
    dev.upload_index(12, 3, idx.lookup, idx.occ[:idx.n_occ])
    assert dev.seed_kernel(e=e, a=a) == ("seed_join_banked_kernel" if banked else "seed_join_kernel")
    got = dev.map_batch(batch.bases, batch.off, e=e, a=a)
    off, cand, ed, end = got.per_strand()
    assert np.array_equal(off, want.cand_off), "candidate counts per (read, strand)"
    assert np.array_equal(cand, want.cands), "candidate locations"
    assert np.array_equal(ed, want.v_ed), "edit distances / accept set"
    assert np.array_equal(end[ed != 0xFF], want.v_end[want.v_ed != 0xFF]), "end offsets"
    assert np.array_equal(got.stats, want.stats), (got.stats, want.stats)
    rec = dev.fetch_records()  # the tail works on (sequence, position): nothing of the banks is left in it
    assert np.array_equal(rec.rec_begin, want.rec_off) and np.array_equal(rec.tid, want.r_tid) and np.array_equal(rec.pos0, want.r_pos)
    assert np.array_equal(rec.cigar, want.cig) and np.array_equal(rec.md, want.md)
    return want


@pytest.mark.parametrize("e,a,L,bank_bases,n_seq", [(3, 1, 100, 110_000, 7), (7, 1, 150, 130_000, 9), (2, 2, 80, 90_000, 6),
                                                     (0, 1, 60, 110_000, 5), (5, 1, 120, 1_000_000, 4)])
def test_banked_reference_equals_oracle(e, a, L, bank_bases, n_seq):
    rng = np.random.default_rng(5200 + 10 * e + a)
    seqs = _reference(rng, n_seq, shared=True)
    reads = util.make_reads(rng, seqs, 900, L, e, n_rate=0.002)
    # reads at the very start and end of every sequence: the first and last sequences of a bank, the remapped entries
    for s in seqs:
        reads += [s[:L], s[-L:], fo.revcomp(s[:L]), fo.revcomp(s[-L:]), s[3:3 + L], s[-L - 5:-5]]
    dev = _device(bank_bases)
    try:
        want = _compare(dev, seqs, reads, e, a, banked=bank_bases < 1_000_000)
    finally:
        dev.close()
    assert want.stats[1] > 0.7 * 900
    per_strand = np.diff(want.cand_off.astype(np.int64))
    assert per_strand.max() >= 8, "fixture must hold strands with candidates in several places"
    # candidates of one strand in more than one sequence (hence, with these bank sizes, in more than one bank)
    spans = 0
    for i in np.nonzero(per_strand >= 2)[0][:4000]:
        c = want.cands[int(want.cand_off[i]):int(want.cand_off[i + 1])]
        spans += len(set((c >> np.uint64(32)).tolist())) > 1
    assert spans > 20


def test_banks_cut_by_the_number_of_sequences():
    # a bank also ends at 2^18 sequences (the remapped near-start entries of the 32-bit table carry the sequence's index
    # within its bank); FEM_TEST_BANK_SEQS = 2 cuts seven sequences into banks of 2 + 2 + 2 + 1.  Reads at every
    # sequence's first bases take that path.
    rng = np.random.default_rng(77)
    seqs = _reference(rng, 7, shared=True)
    L, e = 100, 3
    reads = util.make_reads(rng, seqs, 600, L, e, n_rate=0.002)
    for s in seqs:
        for at in (0, 1, 2, 5, 30, 200, 900, 1020, 1030):
            reads += [s[at:at + L], fo.revcomp(s[at:at + L])]
    dev = _device(10_000_000, bank_seqs=2)
    try:
        _compare(dev, seqs, reads, e, 1)
    finally:
        dev.close()


def test_more_banks_than_the_join_takes_falls_back():
    # nine sequences of ~40 kbp at 30 000 coordinates per bank would need nine banks: the library declines the dense
    # tables and the 64-bit form runs — same results
    from fem_amd import Device
    rng = np.random.default_rng(61)
    seqs = _reference(rng, 9, shared=True)
    reads = util.make_reads(rng, seqs, 400, 100, 3, n_rate=0.0)
    ref = fo.Reference(seqs)
    idx = fo.OracleIndex(ref)
    batch = fo.ReadBatch(reads)
    want = fo.map_reads(ref, idx, batch, e=3, a=1)
    dev = _device(30_000)
    try:
        dev.upload_reference(seqs)
        dev.upload_index(12, 3, idx.lookup, idx.occ[:idx.n_occ])
        assert dev.seed_kernel(e=3) not in ("seed_join_kernel", "seed_join_banked_kernel")
        got = dev.map_batch(batch.bases, batch.off, e=3)
        off, cand, ed, end = got.per_strand()
        assert np.array_equal(off, want.cand_off) and np.array_equal(cand, want.cands) and np.array_equal(ed, want.v_ed)
        assert np.array_equal(got.stats, want.stats)
    finally:
        dev.close()


def test_fuzz_slice_on_a_banked_reference():
    # tests/fuzz_gpu.py's random batches (every e and a, lengths 30-300, damage, N, lower case, records and SAM text)
    # on its repeat-rich reference cut into two banks: three 23 kbp sequences | one of 200 kbp
    from tests import fuzz_gpu
    os.environ["FEM_FORCE_DENSE"] = "1"
    os.environ["FEM_TEST_BANK_BASES"] = "150000"
    seen = []
    try:
        n, bad = fuzz_gpu.run(909, kinds=("repeat",), trials_per_kind=60, threads=8, max_reads=3000,
                              log=lambda *a, **k: seen.append(a[1]["kernel"]))
    finally:
        os.environ.pop("FEM_FORCE_DENSE")
        os.environ.pop("FEM_TEST_BANK_BASES")
    assert n == 60 and bad == 0
    assert seen.count("seed_join_banked_kernel") > 40, seen  # (e + 1 + a > 10 runs the generic kernel)


def test_a_higher_bank_whose_part_of_u_is_dropped_whole():
    # Round-3 review: "U has entries in bank b" must mean entries the reference KEEPS.  Here a piece of the read sits at the very
    # start of the first sequence of bank 1, so that the occurrences of its k-mers there have pos < the seed's offset in the read
    # and are dropped (src/filter.c:89,106): max(U) then lies in bank 0, and the last run — a poly-A k-mer with a tandem of
    # occurrences above max(U) in bank 0 — must be cut there (src/filter.c:85) instead of being merged whole.  Many placements of
    # the pieces, both strands, every one against the oracle.
    rng = np.random.default_rng(8585)
    L, e = 100, 3
    reads, seqs_sets = [], []
    for trial in range(12):
        core = bytearray(util.rand_seq(rng, L))
        at = int(rng.integers(30, 70))
        core[at:at + 14] = b"A" * 14                      # the read holds a poly-A k-mer: frequent, hence the last run
        core = bytes(core)
        cut0 = int(rng.integers(8, 40))
        piece = core[cut0:cut0 + int(rng.integers(20, 45))]  # its k-mers start at read offset cut0 + j, at pos j of sequence B
        a_seq = (util.rand_seq(rng, int(rng.integers(1500, 4000))) + core + util.rand_seq(rng, int(rng.integers(400, 2500))) +
                 b"A" * int(rng.integers(20, 70)) + util.rand_seq(rng, 900) + b"A" * 40 + util.rand_seq(rng, 1200))
        b_seq = piece + util.rand_seq(rng, int(rng.integers(2500, 5000)))
        seqs_sets.append([a_seq, b_seq])
        reads.append([core, fo.revcomp(core), util.mutate(rng, core, 1)[:L].ljust(L, b"C"), core[:L - 1] + b"G"])
    for seqs, rds in zip(seqs_sets, reads):
        dev = _device(len(seqs[0]) + 2048 + 100)  # bank 0 = sequence A, bank 1 = sequence B
        try:
            _compare(dev, seqs, rds + util.make_reads(rng, seqs, 60, L, e), e, 1)
        finally:
            dev.close()
