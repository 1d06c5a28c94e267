"""Times the device SAM text (fem_dev_fetch_sam: tail + sam_len_kernel / scan / sam_write_kernel) on the C2 workload: 2.5 M
reads mapped, then tail + text three times.  profiles/r02_c2_sam_kernel_stats.csv =
rocprofv3 --kernel-trace --stats -- python3 profiles/sam_profile.py (run from the repo root)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from fem_amd import Device, host
from fem_amd.device import _BatchSam
dev = Device(0)
text, off, lens = host.synth_reference(2, [5_000_000], threads=16)
dev.upload_reference([text[:5_000_000]]); dev.build_index(12, 3, fetch=False)
dev.upload_reference_names(["chr1"])
n = 2_500_000
b, o = host.synth_reads(2, text, off, lens, n, 100, 3, threads=16)
q = np.full(n * 100, ord("I"), np.uint8)
names = ["r%d" % i for i in range(n)]
for rep in range(3):
    dev.stage_reads(b, o, slot=0); dev.stage_text(q, names, slot=0)
    dev.map_staged(e=3, slot=0); dev.sync(0)
    dev.set_timing(True); dev.reset_timing()
    r = _BatchSam()
    dev._check(dev._L.fem_dev_fetch_sam(dev._h, 0, C.byref(r)))  # (no copy of the text into Python)
    dev.set_timing(False)
    print("records", int(r.n_records), "text MB", round(r.len / 1e6, 1), {k: round(dev.kernel_time(k)[0], 3) for k in (3, 4, 5, 7)}, flush=True)
