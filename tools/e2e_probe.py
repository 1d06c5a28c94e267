"""tools/e2e_probe.py <c2|c3> <n_reads> [ENV=VAL ...] : FEM map to /dev/null on generated files, stage times + (FEM_TIMELINE) device timeline"""
import os, sys, time, subprocess, tempfile, shutil, re
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from fem_amd import host
key, n_reads = sys.argv[1], int(sys.argv[2])
runs = [a for a in sys.argv[3:]]
w = bench.WORKLOADS[key]
text, off, lens = host.synth_reference(w["seed"] if key == "c2" else 3, w["seq_lens"], threads=16)
d = tempfile.mkdtemp(prefix="fem_e2e_", dir="/dev/shm")
exe = os.path.join(os.getcwd(), "fem_amd", "csrc", "FEM")
try:
    fa, fq, ix = (os.path.join(d, n) for n in ("ref.fa", "reads.fq", "ref.idx"))
    host.write_fasta(fa, text, off, lens)
    bases, _ = host.synth_reads(w["seed"], text, off, lens, n_reads, w["L"], w["e"], first_read=0, threads=16)
    host.write_fastq(fq, bases, w["L"], n_reads)
    r = subprocess.run([exe, "index", "12", "3", fa, ix], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-300:]
    for spec in runs or [""]:
        env = dict(os.environ, FEM_STAGE_TIMES="1")
        for kv in spec.split(","):
            if "=" in kv:
                k, v = kv.split("=", 1); env[k] = v
        out = env.pop("OUT", "/dev/null")
        nthr = env.pop("T", "16")
        t0 = time.perf_counter()
        r = subprocess.run([exe, "map", "-e", str(w["e"]), "-t", nthr, "--ref", fa, "--index", ix, "--read1", fq, "-o", out], capture_output=True, text=True, timeout=600, env=env)
        m = re.search(r"Time: ([0-9.]+)s", r.stderr)
        print("==== %s %d reads [%s]: rc %d Time %s -> %.1f Mreads/s (wall %.2f)" % (key, n_reads, spec, r.returncode, m.group(1) if m else None, n_reads / float(m.group(1)) / 1e6 if m else 0, time.perf_counter() - t0))
        for l in r.stderr.split("\n"):
            if l.startswith("[FEM]") or l.startswith("TL") or l.startswith("[fetch_sam]") or l.startswith("[tail]"):
                print(l)
finally:
    shutil.rmtree(d, ignore_errors=True)
