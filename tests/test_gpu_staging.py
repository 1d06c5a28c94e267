"""The zero-copy read stream of the C ABI (fem_dev_acquire_stage / fem_dev_commit_stage, include/fem_hip.h) against the
copying form (fem_dev_stage_reads) and the oracle, and the device pipeline bench.py times (several slots in flight).
Reference shape: the reusable SequenceBatch ring of src/input_queue.c:34-79.  Needs a GPU: -m gpu."""
import os

import numpy as np
import pytest

from oracle import fem_oracle as fo
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from fem_amd import Device
    rng = np.random.default_rng(90)
    seqs = [util.rand_seq(rng, 250_000), util.rand_seq(rng, 40_000)]
    ref = fo.Reference(seqs)
    idx = fo.OracleIndex(ref)
    dev = Device(0)
    dev.upload_reference(seqs)
    dev.upload_index(12, 3, idx.lookup, idx.occ[:idx.n_occ])
    yield rng, seqs, ref, idx, dev
    dev.close()


def _fill(dev, slot, reads, extra_reads=0, extra_bases=0):
    n, nb = len(reads), sum(len(r) for r in reads)
    hb, ho = dev.acquire_stage(n + extra_reads, nb + extra_bases, slot=slot)
    assert len(hb) >= nb + 64 and len(ho) >= n + 1
    at = 0
    for i, r in enumerate(reads):
        ho[i] = at
        hb[at:at + len(r)] = np.frombuffer(r, np.uint8)
        at += len(r)
    ho[n] = at
    return n, max((len(r) for r in reads), default=0)


def test_staged_batches_equal_copied_batches_and_the_oracle(setup):
    rng, seqs, ref, idx, dev = setup
    batches = [util.make_reads(rng, seqs, n, L, 3) for n, L in ((700, 100), (33, 150), (1200, 64), (5, 100))]
    wants = [fo.map_reads(ref, idx, fo.ReadBatch(b), e=3, stages=fo.STAGE_SEED | fo.STAGE_VERIFY) for b in batches]
    # all four slots filled, committed and mapped before any is fetched: H2D, kernels and D2H of different slots overlap
    for slot, b in enumerate(batches):
        n, mx = _fill(dev, slot, b, extra_reads=slot * 7, extra_bases=slot * 1000)
        dev.commit_stage(n, mx, slot=slot)
        dev.map_staged(e=3, slot=slot)
    for slot in (2, 0, 3, 1):  # fetched out of order
        got = dev.fetch(slot=slot)
        want = wants[slot]
        off, cand, ed, end = got.per_strand()
        assert np.array_equal(got.stats, want.stats)
        assert np.array_equal(off, want.cand_off) and np.array_equal(cand, want.cands) and np.array_equal(ed, want.v_ed)
        assert np.array_equal(end[ed != 0xFF], want.v_end[want.v_ed != 0xFF])
    # the same batches through the copying entry point, slots reused
    for slot, b in enumerate(batches):
        rb = fo.ReadBatch(b)
        got = dev.map_batch(rb.bases, rb.off, e=3, slot=(slot + 1) % 4)
        assert np.array_equal(got.stats, wants[slot].stats)
        assert np.array_equal(got.per_strand()[1], wants[slot].cands)
    # a slot's staging survives its batch: committing it again maps the same reads again (what bench.py's steps do)
    n, mx = _fill(dev, 1, batches[0])
    for _ in range(3):
        dev.commit_stage(n, mx, slot=1)
        dev.map_staged(e=3, slot=1)
        assert np.array_equal(dev.fetch_stats(slot=1), wants[0].stats)
    # reads of one length: the offsets are generated on the device (they are not even looked at in the staging buffer)
    hb, ho = dev.acquire_stage(n, sum(len(r) for r in batches[0]), slot=2)
    hb[:n * 100] = np.frombuffer(b"".join(batches[0]), np.uint8)
    ho[:] = 0xDEAD
    dev.commit_stage(n, 100, slot=2, uniform=True)
    dev.map_staged(e=3, slot=2)
    got = dev.fetch(slot=2)
    assert np.array_equal(got.stats, wants[0].stats) and np.array_equal(got.per_strand()[1], wants[0].cands)
    rec = dev.fetch_records(slot=1)  # the device tail reads the staged characters too
    assert rec.n_records == int(wants[0].stats[4])


def test_staging_errors_are_reported_not_fatal(setup):
    from fem_amd import FemError
    rng, seqs, ref, idx, dev = setup
    reads = util.make_reads(rng, seqs, 10, 100, 3)
    with pytest.raises(FemError):
        dev.acquire_stage(10, 1000, slot=9)  # no such slot
    n, mx = _fill(dev, 0, reads)
    with pytest.raises(FemError):
        dev.commit_stage(n + 5, mx, slot=0)  # more reads than the buffers were acquired for
    with pytest.raises(FemError):
        dev.commit_stage(n, 5000, slot=0)  # a read longer than the device path takes
    hb, ho = dev.acquire_stage(4, 400, slot=3)
    ho[0] = 8  # offsets must start at 0
    ho[1:5] = [108, 208, 308, 408]
    with pytest.raises(FemError):
        dev.commit_stage(4, 100, slot=3)
    # an empty batch is fine
    dev.acquire_stage(0, 0, slot=2)
    dev.commit_stage(0, 0, slot=2)
    dev.map_staged(e=3, slot=2)
    assert dev.fetch_stats(slot=2).tolist() == [0, 0, 0, 0, 0]
    # stage_reads still checks what a caller hands it
    bases = np.zeros(3000, np.uint8) + 65
    with pytest.raises(FemError):
        dev.stage_reads(bases, np.array([0, 100, 50], np.uint64))  # offsets go backwards
    with pytest.raises(FemError):
        dev.stage_reads(bases, np.array([0, 2000], np.uint64))  # longer than the limit
    # and the handle still works afterwards
    n, mx = _fill(dev, 0, reads)
    dev.commit_stage(n, mx, slot=0)
    dev.map_staged(e=3, slot=0)
    want = fo.map_reads(ref, idx, fo.ReadBatch(reads), e=3, stages=fo.STAGE_SEED | fo.STAGE_VERIFY)
    assert np.array_equal(dev.fetch_stats(slot=0), want.stats)


def test_host_placement_next_to_the_gpu():
    # fem_device_numa / fem_bind_thread_near_device: the calling thread ends up on a subset of its CPUs, all of them on
    # the GPU's node; FEM_NUMA_BIND=0 turns it off.  (In a child process: the binding is inherited by later threads.)
    import subprocess
    import sys
    code = r'''
import os, sys
sys.path.insert(0, %r)
from fem_amd import device
before = os.sched_getaffinity(0)
node, cpus = device.device_numa(0)
os.environ["FEM_NUMA_BIND"] = "0"
assert device.bind_near_device(0) is False and os.sched_getaffinity(0) == before
del os.environ["FEM_NUMA_BIND"]
bound = device.bind_near_device(0)
after = os.sched_getaffinity(0)
assert after <= before
if node >= 0 and cpus:
    want = set()
    for part in cpus.split(","):
        lo, _, hi = part.partition("-")
        want |= set(range(int(lo), int(hi or lo) + 1))
    if want & before:
        assert bound and after == want & before, (node, cpus, sorted(after)[:4])
    else:
        assert not bound and after == before
else:
    assert not bound and after == before
print("ok", node, cpus, len(before), len(after))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith(b"ok"), r.stderr.decode()


def test_copying_pipeline_with_changing_batches(setup):
    # fem_dev_stage_reads over all four slots, three batches in flight, with what the packed transfer and the results that
    # come home behind the kernels are sensitive to: batch sizes going up and down, equal and mixed read lengths, lengths
    # that are no multiple of four, unpackable bytes, empty batches — every batch against the oracle
    rng, seqs, ref, idx, dev = setup
    shapes = []
    for i in range(36):
        n = int(rng.choice([0, 1, 7, 300, 2500, 9000]))
        L = int(rng.choice([100, 150, 101, 64]))
        shapes.append((n, L, rng.random() < 0.3, rng.random() < 0.3))
    batches, wants = [], []
    for n, L, mixed, odd in shapes:
        reads = util.make_reads(rng, seqs, n, L, 3)
        if mixed and n > 3:
            reads[1] = reads[1][:L - 5]
        if odd and n:
            for j in range(0, n, 5):
                r = bytearray(reads[j])
                r[int(rng.integers(0, len(r)))] = ord("N")
                reads[j] = bytes(r) if j % 10 else bytes(r).lower()
        b = fo.ReadBatch(reads)
        batches.append(b)
        wants.append(fo.map_reads(ref, idx, b, e=3, stages=fo.STAGE_SEED | fo.STAGE_VERIFY))
    depth, n_slots = 3, 4

    def check(i):
        got = dev.fetch(slot=i % n_slots, copy=False)
        want = wants[i]
        off, cand, ed, end = got.per_strand()
        assert np.array_equal(got.stats, want.stats), i
        assert np.array_equal(off, want.cand_off) and np.array_equal(cand, want.cands) and np.array_equal(ed, want.v_ed), i
        assert np.array_equal(end[ed != 0xFF], want.v_end[want.v_ed != 0xFF]), i

    n_packed = 0
    for i, b in enumerate(batches):
        if i >= depth:
            check(i - depth)
        dev.stage_reads(b.bases, b.off, slot=i % n_slots)
        n_packed += dev.stage_info(i % n_slots)[1]
        dev.map_staged(e=3, slot=i % n_slots)
    for i in range(len(batches) - depth, len(batches)):
        check(i)
    assert 8 < n_packed < len(batches)


def test_two_handles_from_two_threads(setup):
    # handles are independent (include/fem_hip.h): two of them on one GPU, each driven by its own host thread through the
    # copying entry point (own staging threads, own streams), give what one gives
    import threading
    from fem_amd import Device
    rng, seqs, ref, idx, dev = setup
    other = Device(0)
    other.upload_reference(seqs)
    other.upload_index(12, 3, idx.lookup, idx.occ[:idx.n_occ])
    batches = [fo.ReadBatch(util.make_reads(rng, seqs, 20_000, 100, 3)) for _ in range(2)]
    wants = [fo.map_reads(ref, idx, b, e=3, stages=fo.STAGE_SEED | fo.STAGE_VERIFY) for b in batches]
    errors = []

    def work(d, b, want):
        try:
            for rep in range(6):
                got = d.map_batch(b.bases, b.off, e=3, slot=rep % 4)
                assert np.array_equal(got.stats, want.stats) and np.array_equal(got.per_strand()[1], want.cands)
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    threads = [threading.Thread(target=work, args=(d, b, w)) for d, b, w in zip((dev, other), batches, wants)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    other.close()
    assert not errors, errors
