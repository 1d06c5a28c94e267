"""SAM text rendered on the device (fem_dev_fetch_sam: mapping tail + sam_len_kernel / sam_write_kernel, fem_tail.hip)
against the host formatter on the device's records (fem_records_sam) and against the text built from the oracle's records
(field rules of src/align.c:546-632, src/output_queue.c:93-116).  Needs a GPU: -m gpu."""
import numpy as np
import pytest

from fem_amd import host
from oracle import fem_oracle as fo
from tests import util
from tests.test_host import expected_sam

pytestmark = pytest.mark.gpu


def _case(seed, e, L, n_reads, repeats, odd):
    from fem_amd import Device
    rng = np.random.default_rng(seed)
    if repeats:
        seqs = util.repeat_rich_reference(rng, n_seq=3, unit_len=300, n_units=4, copies=50, spacer=200)
        seqs.append(util.rand_seq(rng, 120_000))
    else:
        seqs = [util.rand_seq(rng, 200_000), util.rand_seq(rng, 50_000)]
    long_fields = L > 200  # names, reference names and MD strings beyond one and two bytes per lane of sam_write_kernel
    names = ["chr%d_%s" % (i, "x" * (i * (70 if long_fields else 7))) for i in range(len(seqs))]
    reads = util.make_reads(rng, seqs, n_reads, L, e, n_rate=0.003)
    if odd:
        reads[3] = reads[3].lower()
        reads[5] = reads[5][:L // 2] + b"RYKM=.-*"[:min(8, L - L // 2)] + reads[5][L // 2 + 8:]
        reads[7] = reads[7][:L - 17]  # a different length
    rnames = ["read_%d/%s" % (i, "n" * (i % (150 if long_fields else 40))) for i in range(len(reads))]
    quals = ["".join(chr(33 + (11 * i + j) % 60) for j in range(len(r))) for i, r in enumerate(reads)]
    ref = fo.Reference(seqs)
    idx = fo.OracleIndex(ref)
    dev = Device(0)
    dev.upload_reference(seqs)
    dev.upload_reference_names(names)
    dev.upload_index(12, 3, idx.lookup, idx.occ[:idx.n_occ])
    return dev, ref, idx, seqs, names, reads, rnames, quals


@pytest.mark.parametrize("seed,e,L,n,repeats,odd", [(1, 3, 100, 1500, False, True), (2, 7, 150, 800, True, True),
                                                     (3, 2, 64, 3000, True, False), (4, 0, 36, 500, False, False),
                                                     (5, 7, 260, 700, False, True)])
def test_device_sam_text_equals_host_text_and_oracle_text(seed, e, L, n, repeats, odd):
    dev, ref, idx, seqs, names, reads, rnames, quals = _case(seed, e, L, n, repeats, odd)
    try:
        batch = fo.ReadBatch(reads)
        want = fo.map_reads(ref, idx, batch, e=e)
        q = np.frombuffer("".join(quals).encode(), np.uint8)
        dev.reserve_batch(len(reads), 2 * len(reads), L, e=e, slot=2)  # (slot 2 with its allocations made ahead, slot 0 as they come)
        for slot in (0, 2):
            dev.stage_reads(batch.bases, batch.off, slot=slot)
            dev.stage_text(q, rnames, slot=slot)
            dev.map_staged(e=e, slot=slot)
            text, n_records, n_asserted, stats = dev.fetch_sam(slot=slot, nowait=slot == 2)
            assert np.array_equal(stats, want.stats) and n_records == int(want.rec_off[-1])
            # the same batch with its qualities kept on the host (fem_dev_commit_names_stage): the device leaves the QUAL field of
            # every read's first record open, fem_dev_sam_quals says where, fem_sam_fill_quals fills it in — the same bytes
            dev.stage_text(q, rnames, slot=slot, quals_on_host=True)
            text_h, n_records_h, _, _ = dev.fetch_sam(slot=slot, nowait=slot == 2, quals=q, offsets=batch.off)
            assert text_h == text and n_records_h == n_records
            dev.stage_text(q, rnames, slot=slot)  # (and with them on the device again, for the records below)
            # the host formatter on the records the device tail hands out
            rec = dev.fetch_records(slot=slot)
            tref = host.TailReference(ref.text, ref.off, ref.len, names=names)
            host_text, host_asserted = host.records_sam(tref, rnames, batch.bases, batch.off, q, rec, threads=3, parts=True)
            assert text.decode("latin-1") == host_text
            assert n_asserted == host_asserted == int(np.count_nonzero(rec.flag & 0x8000))
            if n_asserted == 0:
                exp = expected_sam(names, reads, rnames, quals, want)
                # SEQ goes through the 4-bit round trip of the BAM record: IUPAC letters stay, anything else is N
                assert [l.split("\t")[:9] + l.split("\t")[10:] for l in text.decode("latin-1").splitlines()] == \
                       [l.split("\t")[:9] + l.split("\t")[10:] for l in exp.splitlines()]
        assert n_records > n // 2 or L < 40
    finally:
        dev.close()


def test_fetch_sam_needs_its_inputs():
    from fem_amd import FemError
    dev, ref, idx, seqs, names, reads, rnames, quals = _case(9, 3, 100, 50, False, False)
    try:
        batch = fo.ReadBatch(reads)
        dev.stage_reads(batch.bases, batch.off)
        dev.map_staged(e=3)
        with pytest.raises(FemError):
            dev.fetch_sam()  # no qualities / names committed for this batch
        q = np.frombuffer("".join(quals).encode(), np.uint8)
        with pytest.raises(FemError):
            dev.stage_text(q, rnames[:-1])  # one name short
        # fem_dev_reserve_batch: shapes and parameters out of range are refused, and a handle without an index
        for bad in (dict(n_reads=0, n_records=1, max_len=100), dict(n_reads=10, n_records=10, max_len=0),
                    dict(n_reads=10, n_records=10, max_len=100_000), dict(n_reads=1 << 40, n_records=10, max_len=100)):
            with pytest.raises(FemError):
                dev.reserve_batch(slot=3, **bad)
        with pytest.raises(FemError):
            dev.reserve_batch(50, 50, 100, e=9, slot=3)  # parameters out of range
        dev.reserve_batch(50, 100, 100, e=3, slot=3)
        from fem_amd import Device
        bare = Device(0)
        try:
            with pytest.raises(FemError):
                bare.reserve_batch(50, 50, 100)
        finally:
            bare.close()
        empty = fo.ReadBatch([])
        dev.stage_reads(empty.bases, empty.off, slot=1)
        dev.stage_text(np.zeros(0, np.uint8), [], slot=1)
        dev.map_staged(e=3, slot=1)
        text, n_records, n_asserted, stats = dev.fetch_sam(slot=1)
        assert text == b"" and n_records == 0
    finally:
        dev.close()
