#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

The reference (haowenz/FEM) cannot be built in this image (htslib is an un-vendored submodule) and ships no golden
vectors, so these fixtures are produced by the CPU ORACLE (oracle/fem_oracle.c), not by the reference itself: they
freeze the oracle + the seeded generator so that later changes to either, or to the device path, are caught.
Inputs are not stored: they are a pure function of (seed, sizes) through libfemhost's generator.

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from fem_amd import host  # noqa: E402
from oracle import fem_oracle as fo  # noqa: E402

CASES = {
    # BASELINE.json configs[0]: 1k synthetic 100 bp reads, e=3, 1 Mbp random reference, k=12 step=3
    "c1_seed1": dict(seed=1, seq_lens=[1_000_000], n_reads=1000, L=100, e=3, a=1),
    # multi-sequence, longer reads, maximum error threshold
    "multi_e7": dict(seed=7, seq_lens=[300_000, 150_000, 5_000], n_reads=600, L=150, e=7, a=1),
    # repeat-rich, three sequences with N runs — hundreds of candidates per strand (full groups of 8: the
    # 16-bit Myers lanes), reads with more than 64 records (klib's radix sort instead of the insertion sort), secondary
    # flags, indels.  Its inputs come from numpy's generator, so they are STORED in the fixture (reference + reads).
    "repeat_rich": dict(kind="repeat", seed=31, n_reads=500, L=100, e=3, a=1, n_rate=0.003),
}
HERE = os.path.dirname(os.path.abspath(__file__))


def make_repeat_inputs(case):
    """The repeat-rich case's inputs from scratch (what the committed fixture stores)."""
    from tests import util
    rng = np.random.default_rng(case["seed"])
    seqs = util.repeat_rich_reference(rng, n_seq=3, unit_len=300, n_units=3, copies=90, spacer=200)
    reads = util.make_reads(rng, seqs, case["n_reads"], case["L"], case["e"], n_rate=case["n_rate"])
    lens = np.array([len(s) for s in seqs], np.uint32)
    off = np.concatenate([[0], np.cumsum(lens.astype(np.uint64))]).astype(np.uint64)[:-1]
    text = np.frombuffer(b"".join(seqs), np.uint8)
    rlen = np.array([len(r) for r in reads], np.uint64)
    offs = np.concatenate([[0], np.cumsum(rlen)]).astype(np.uint64)
    bases = np.frombuffer(b"".join(reads) + b"\0" * 64, np.uint8)
    return text, off, lens, bases, offs


def inputs(case, name=None):
    if case.get("kind") == "repeat":
        path = os.path.join(HERE, (name or "repeat_rich") + ".npz")
        if os.path.exists(path):
            z = np.load(path)
            return z["in_text"], z["in_off"], z["in_lens"], z["in_bases"], z["in_offs"]
        return make_repeat_inputs(case)
    text, off, lens = host.synth_reference(case["seed"], case["seq_lens"], threads=4)
    bases, offs = host.synth_reads(case["seed"], text, off, lens, case["n_reads"], case["L"], case["e"], threads=4)
    return text, off, lens, bases, offs


def oracle_outputs(case):
    text, off, lens, bases, offs = inputs(case)
    ref = fo.Reference([text[int(o):int(o) + int(l)].tobytes() for o, l in zip(off, lens)])
    idx = fo.OracleIndex(ref)
    res = fo.map_reads(ref, idx, fo.ReadBatch.from_arrays(bases, offs), e=case["e"], a=case["a"])
    sam = "".join("%d\t%d\t%d\t%d\t%s\t%d\t%s\n" % (r, int(res.r_flag[j]), int(res.r_tid[j]), int(res.r_pos[j]) + 1,
                                                     res.cigar_str(j), int(res.r_nm[j]), res.md_str(j))
                  for r in range(case["n_reads"]) for j in range(int(res.rec_off[r]), int(res.rec_off[r + 1])))
    extra = {}
    if case.get("kind") == "repeat":
        per_read = np.diff(res.rec_off.astype(np.int64))
        per_strand = np.diff(res.cand_off.astype(np.int64))
        extra = dict(in_text=text, in_off=off, in_lens=lens, in_bases=bases, in_offs=offs,
                     max_records_per_read=np.array([per_read.max()], np.uint64), reads_over_64_records=np.array([(per_read > 64).sum()], np.uint64),
                     strands_with_full_groups=np.array([(per_strand >= 8).sum()], np.uint64))
    return dict(stats=res.stats, cand_off=res.cand_off, cands=res.cands, v_ed=res.v_ed, v_end=res.v_end, **extra,
                index_sha256=np.frombuffer(hashlib.sha256(idx.lookup.tobytes() + idx.occ[:idx.n_occ].tobytes()).digest(), np.uint8),
                records_sha256=np.frombuffer(hashlib.sha256(sam.encode()).digest(), np.uint8),
                n_records=np.array([len(res.r_flag)], np.uint64))


if __name__ == "__main__":
    for name, case in CASES.items():
        if case.get("kind") == "repeat" and os.path.exists(os.path.join(HERE, name + ".npz")) and "--regenerate-inputs" in sys.argv:
            os.unlink(os.path.join(HERE, name + ".npz"))
        out = oracle_outputs(case)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "stats", out["stats"].tolist(), "records", int(out["n_records"][0]),
              {k: int(out[k][0]) for k in ("max_records_per_read", "reads_over_64_records", "strands_with_full_groups") if k in out})
