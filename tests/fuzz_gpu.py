"""Randomised differential run: device path vs oracle over references of every index density, random parameters, read
lengths, damage — candidates, edit distances, end offsets, counters and (on the small references) the tail's records and
the SAM text.  By the clock:  FUZZ_SECONDS=420 FUZZ_SEED=11 python tests/fuzz_gpu.py  (minutes of oracle time, not
collected by pytest); a bounded slice with fixed seeds and trial counts runs under `pytest -m gpu`
(tests/test_gpu_fuzz.py)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from fem_amd import Device, host
from oracle import fem_oracle as fo
from tests import util

KINDS = ("repeat", "sparse", "mid", "dense")


def reference(rng, kind, threads=16):
    if kind == "dense":
        text, off, lens = host.synth_reference(int(rng.integers(1, 1 << 30)), [72_000_000] * 3, threads=threads)
        return [text[int(o):int(o) + int(l)].tobytes() for o, l in zip(off, lens)]
    if kind == "mid":
        text, off, lens = host.synth_reference(int(rng.integers(1, 1 << 30)), [30_000_000, 5_000_000, 1000, 13], threads=threads)
        return [text[int(o):int(o) + int(l)].tobytes() for o, l in zip(off, lens)]
    if kind == "sparse":
        text, off, lens = host.synth_reference(int(rng.integers(1, 1 << 30)), [3_000_000, 2_000_000], threads=threads)
        return [text[int(o):int(o) + int(l)].tobytes() for o, l in zip(off, lens)]
    seqs = util.repeat_rich_reference(rng, n_seq=3, unit_len=300, n_units=6, copies=60, spacer=150)
    seqs.append(util.rand_seq(rng, 200_000))
    return seqs


def trial(rng, dev, ref, idx, tref, seqs, kind, threads=16, max_reads=15000, log=print):
    """One random batch through oracle and device; True if everything compared is identical."""
    e = int(rng.integers(0, 8)); a = int(rng.choice([1, 1, 1, 2]))
    L = int(rng.integers(30, 301)); n = int(rng.integers(500, min(6000, max_reads) if kind == "repeat" else max_reads))
    reads = util.make_reads(rng, seqs, n, L, min(e + 2, 9), n_rate=float(rng.choice([0, 0, 0.002, 0.02])))
    if rng.random() < 0.4:  # mixed lengths
        for j in range(0, n, 3):
            reads[j] = reads[j][:int(rng.integers(max(1, L // 3), L + 1))]
    if rng.random() < 0.3:
        for j in range(0, n, 11):
            reads[j] = reads[j].lower()
    for j in range(0, n, 97):
        reads[j] = util.rand_seq(rng, len(reads[j]))
    b = fo.ReadBatch(reads)
    full = kind != "dense" and n < 8000
    want = fo.map_reads(ref, idx, b, e=e, a=a, threads=threads, stages=(fo.STAGE_SEED | fo.STAGE_VERIFY | (fo.STAGE_ALIGN if full else 0)))
    slot = int(rng.integers(0, 4))
    dev.stage_reads(b.bases, b.off, slot=slot)
    if full:  # qualities and names for the device's SAM text
        quals = rng.integers(33, 100, size=len(b.bases)).astype(np.uint8)
        rnames = ["q%d_%s" % (j, "z" * int(rng.integers(0, 70))) for j in range(n)]
        dev.stage_text(quals, rnames, slot=slot)
    dev.map_staged(e=e, a=a, slot=slot)
    got = dev.fetch(slot=slot)
    o, cand, ed, end = got.per_strand()
    ok = (np.array_equal(o, want.cand_off) and np.array_equal(cand, want.cands) and np.array_equal(ed, want.v_ed)
          and np.array_equal(got.stats, want.stats) and np.array_equal(end[ed != 255], want.v_end[want.v_ed != 255]))
    if ok and full:
        rec = dev.fetch_records(slot=slot)
        ok = (np.array_equal(rec.rec_begin, want.rec_off) and np.array_equal(rec.flag & 0x7FFF, want.r_flag & 0x7FFF) and np.array_equal(rec.pos0, want.r_pos)
              and np.array_equal(rec.nm, want.r_nm) and np.array_equal(rec.cigar_off, want.cig_off) and np.array_equal(rec.cigar, want.cig)
              and np.array_equal(rec.md_off, want.md_off) and np.array_equal(rec.md, want.md))
    if ok and full:  # the text rendered on the device against the host formatter on the same records
        text, n_rec, n_assert, st = dev.fetch_sam(slot=slot, nowait=bool(rng.integers(0, 2)))
        host_text, host_assert = host.records_sam(tref, rnames, b.bases, b.off, quals, rec, threads=4, parts=True)
        ok = text.decode("latin-1") == host_text and n_assert == host_assert and n_rec == rec.n_records
        if ok and rng.integers(0, 2):  # ... and with the qualities kept on the host (the device leaves their field open): the same bytes
            dev.stage_text(quals, rnames, slot=slot, quals_on_host=True)
            text_h, n_rec_h, _, _ = dev.fetch_sam(slot=slot, nowait=bool(rng.integers(0, 2)), quals=quals, offsets=b.off)
            ok = text_h == text and n_rec_h == n_rec
    log(kind, dict(e=e, a=a, L=L, n=n, packed=dev.stage_info(slot)[1], full=full, kernel=dev.seed_kernel(e=e, a=a)), "ok" if ok else "MISMATCH",
        [int(x) for x in got.stats], flush=True)
    return ok


def run(seed, kinds=KINDS, seconds=None, trials_per_kind=None, threads=16, max_reads=15000, log=print):
    """Either `seconds` of trials split over the kinds (the later kinds get what is left), or a fixed number per kind."""
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds if seconds else None
    n_trials = bad = 0
    share = {"repeat": 4, "sparse": 3, "mid": 2, "dense": 1}
    for kind in kinds:
        if t_end and time.time() > t_end:
            break
        seqs = reference(rng, kind, threads)
        ref = fo.Reference(seqs); idx = fo.OracleIndex(ref, threads=threads)
        dev = Device(0); dev.upload_reference(seqs); dev.upload_index(12, 3, idx.lookup, idx.occ[:idx.n_occ])
        ref_names = ["ref%d%s" % (i, "_" * (i % 5)) for i in range(len(seqs))]
        dev.upload_reference_names(ref_names)
        tref = host.TailReference(ref.text, ref.off, ref.len, names=ref_names)
        try:
            if t_end:
                t_kind = time.time() + (t_end - time.time()) / share[kind]
                while time.time() < t_kind:
                    n_trials += 1
                    bad += not trial(rng, dev, ref, idx, tref, seqs, kind, threads, max_reads, log)
            else:
                for _ in range(trials_per_kind[kind] if isinstance(trials_per_kind, dict) else trials_per_kind):
                    n_trials += 1
                    bad += not trial(rng, dev, ref, idx, tref, seqs, kind, threads, max_reads, log)
        finally:
            dev.close()
    return n_trials, bad


if __name__ == "__main__":
    n_trials, bad = run(int(os.environ.get("FUZZ_SEED", "1")), seconds=float(os.environ.get("FUZZ_SECONDS", "240")))
    print("trials", n_trials, "bad", bad)
    sys.exit(1 if bad else 0)
