"""Where does the C2 pipeline's time go: H2D, kernels, D2H in every combination (packed commit form)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fem_amd import Device, host
N_SLOTS, DEPTH = 4, 4
batch, L, e = 2_500_000, 100, 3
text, off, lens = host.synth_reference(2, [5_000_000], threads=16)
dev = Device(0)
dev.upload_reference([text[:5_000_000]])
dev.build_index(12, 3, fetch=False)
for s in range(N_SLOTS):
    hb, _ = dev.acquire_stage(batch, batch * L, slot=s)
    host.synth_reads_packed(2, text, off, lens, batch, L, e, hb, first_read=s * batch, threads=16)

def run(n, send, fetch):
    def submit(i):
        s = i % N_SLOTS
        if send:
            dev.commit_stage_packed(batch, L, 0, slot=s)
        dev.map_staged(e=e, slot=s)
    def retire(i):
        s = i % N_SLOTS
        if fetch == "full":
            dev.fetch(slot=s, copy=False)
        else:
            dev.fetch_stats(slot=s)
    for i in range(n):
        if i >= DEPTH:
            retire(i - DEPTH)
        submit(i)
    for i in range(max(0, n - DEPTH), n):
        retire(i)

out = {}
for name, send, fetch in (("h2d+k+d2h", True, "full"), ("h2d+k", True, "stats"), ("k+d2h", False, "full"), ("k", False, "stats"),
                          ("h2d+k+d2h again", True, "full")):
    run(8, send, fetch)
    t0 = time.perf_counter()
    run(40, send, fetch)
    dt = time.perf_counter() - t0
    out[name] = round(batch * 40 / dt / 1e6, 1)
    print(name, out[name], "Mreads/s", round(dt / 40 * 1e3, 3), "ms/step", flush=True)
print(json.dumps(out))
