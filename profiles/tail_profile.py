"""Times the device mapping tail (fem_dev_fetch_records) on the C2 workload: 10 M reads mapped, then the tail three times.
profiles/r03_c2_tail_kernel_stats.csv = rocprofv3 --kernel-trace --stats --output-format csv -- python3 profiles/tail_profile.py
(run from the repo root).  `subst` as the first argument: reads whose errors are substitutions only (0..3 per read, both
strands) instead of the generator's 60 % substitutions / 20 % insertions / 20 % deletions — what a short-read sequencer's
output looks like to the traceback: nearly every record ends in trace_ident_kernel."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from fem_amd import Device, host
dev = Device(0)
text, off, lens = host.synth_reference(2, [5_000_000], threads=16)
dev.upload_reference([text[:5_000_000]]); dev.build_index(12, 3, fetch=False)
n=10_000_000
if len(sys.argv) > 1 and sys.argv[1] == "subst":
    rng = np.random.default_rng(5)
    L = 100
    b = np.empty(n * L + 64, np.uint8)
    b[n * L:] = 0
    comp = np.zeros(256, np.uint8); comp[list(b"ACGT")] = list(b"TGCA")
    nxt = np.zeros(256, np.uint8); nxt[list(b"ACGT")] = list(b"CGTA")
    for lo in range(0, n, 1_000_000):  # a million reads at a time: 100 MB of indexes
        m = min(1_000_000, n - lo)
        pos = rng.integers(0, 5_000_000 - L, m)
        r = text[pos[:, None] + np.arange(L)[None, :]]
        k = rng.integers(0, 4, m)
        for j in range(3):  # the j-th substitution of the reads that have more than j
            rows = np.nonzero(k > j)[0]
            cols = rng.integers(0, L, len(rows))
            r[rows, cols] = nxt[r[rows, cols]]
        rev = rng.integers(0, 2, m).astype(bool)
        r[rev] = comp[r[rev][:, ::-1]]
        b[lo * L:(lo + m) * L] = r.reshape(-1)
    o = (np.arange(n + 1, dtype=np.uint64) * L)
else:
    b,o = host.synth_reads(2, text, off, lens, n, 100, 3, threads=16)
dev.stage_reads(b,o,slot=0)
for rep in range(3):
    dev.map_staged(e=3,slot=0); dev.sync(0)
    dev.set_timing(True); dev.reset_timing()
    r = dev.fetch_records(slot=0)
    dev.set_timing(False)
    print("records", r.n_records, {k: round(dev.kernel_time(k)[0],3) for k in (0,1,3,4,5)}, flush=True)
