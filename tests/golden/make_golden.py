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
}


def inputs(case):
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
    return dict(stats=res.stats, cand_off=res.cand_off, cands=res.cands, v_ed=res.v_ed, v_end=res.v_end,
                index_sha256=np.frombuffer(hashlib.sha256(idx.lookup.tobytes() + idx.occ[:idx.n_occ].tobytes()).digest(), np.uint8),
                records_sha256=np.frombuffer(hashlib.sha256(sam.encode()).digest(), np.uint8),
                n_records=np.array([len(res.r_flag)], np.uint64))


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for name, case in CASES.items():
        out = oracle_outputs(case)
        np.savez_compressed(os.path.join(here, name + ".npz"), **out)
        print(name, "stats", out["stats"].tolist(), "records", int(out["n_records"][0]))
