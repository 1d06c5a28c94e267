"""The dense path's cliffs: what share of the reads the selection / join kernels hand to the generic
seed_filter_kernel, and what that does to the rate, on (a) a repeat-rich 3 Gbp-class reference (5 % of it 300-bp units in
1000 copies each), 100 bp reads, e = 3; (b) the plain C3 reference with 300 bp and 250 bp reads, e = 3."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fem_amd import Device, host

def rate(dev, text, off, lens, seed, n, L, e, reps=3):
    bases, offs = host.synth_reads(seed, text, off, lens, n, L, e, threads=16)
    dev.set_timing(True)
    out = None
    for r in range(reps):
        dev.reset_timing()
        dev.stage_reads(bases, offs, slot=0)
        t0 = time.perf_counter()
        dev.map_staged(e=e, slot=0)
        st = dev.fetch_stats(slot=0)
        dt = time.perf_counter() - t0
        times = {k: dev.kernel_time(i)[0] for k, i in (("join", 0), ("verify", 1), ("generic", 2), ("select", 8))}
        out = {"Mreads_per_s": round(n / dt / 1e6, 2), "ms": round(dt * 1e3, 2), "kernel_ms": {k: round(v, 3) for k, v in times.items()},
               "stats": [int(x) for x in st], "cand_per_read": round(int(st[3]) / n, 2)}
    dev.set_timing(False)
    return out

which = sys.argv[1:] or ["repeat", "long"]
res = {}
text, off, lens = host.synth_reference(3, [125_000_000] * 24, threads=16)
if "long" in which:
    dev = Device(0)
    dev.upload_reference([text[int(o):int(o) + int(l)] for o, l in zip(off, lens)])
    dev.build_index(12, 3, fetch=False)
    for L in (100, 200, 250, 300):
        res["c3_L%d_e3" % L] = rate(dev, text, off, lens, 33, 1_000_000, L, 3)
        print("c3 reference, L=%d e=3:" % L, json.dumps(res["c3_L%d_e3" % L]), flush=True)
    dev.close()
if "repeat" in which:
    rng = np.random.default_rng(5)
    total = int(lens.astype(np.uint64).sum())
    acgt = np.frombuffer(b"ACGT", np.uint8)
    n_units, copies, ulen = 500, 1000, 300
    for u in range(n_units):
        unit = acgt[rng.integers(0, 4, ulen)]
        seq = rng.integers(0, len(lens), copies)
        at = rng.integers(2000, int(lens[0]) - 2000 - ulen, copies)
        for s_, a_ in zip(seq, at):
            p = int(off[s_]) + int(a_)
            text[p:p + ulen] = unit
    dev = Device(0)
    dev.upload_reference([text[int(o):int(o) + int(l)] for o, l in zip(off, lens)])
    dev.build_index(12, 3, fetch=False)
    res["repeat_L100_e3"] = rate(dev, text, off, lens, 44, 1_000_000, 100, 3)
    print("repeat-rich 3 Gbp (5 %% in 300-bp units x 1000), L=100 e=3:", json.dumps(res["repeat_L100_e3"]), flush=True)
    dev.close()
print(json.dumps(res))
