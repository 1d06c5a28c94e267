"""Times the device mapping tail (fem_dev_fetch_records) on the C2 workload: 10 M reads mapped, then the tail three times.
profiles/r02_c2_tail_kernel_stats.csv = rocprofv3 --kernel-trace --stats -- python3 profiles/tail_profile.py (run from the repo root)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from fem_amd import Device, host
dev = Device(0)
text, off, lens = host.synth_reference(2, [5_000_000], threads=16)
dev.upload_reference([text[:5_000_000]]); dev.build_index(12, 3, fetch=False)
n=10_000_000
b,o = host.synth_reads(2, text, off, lens, n, 100, 3, threads=16)
dev.stage_reads(b,o,slot=0)
for rep in range(3):
    dev.map_staged(e=3,slot=0); dev.sync(0)
    dev.set_timing(True); dev.reset_timing()
    r = dev.fetch_records(slot=0)
    dev.set_timing(False)
    print("records", r.n_records, {k: round(dev.kernel_time(k)[0],3) for k in (0,1,3,4,5)}, flush=True)
