"""tools/e2e_files.py <c2|c3> <n_reads> <dir>: reference FASTA, reads FASTQ and index for FEM map runs, left in <dir>"""
import os, sys, subprocess
sys.path.insert(0, os.getcwd())
import bench
from fem_amd import host
key, n_reads, d = sys.argv[1], int(sys.argv[2]), sys.argv[3]
w = bench.WORKLOADS[key]
os.makedirs(d, exist_ok=True)
text, off, lens = host.synth_reference(w["seed"] if key == "c2" else 3, w["seq_lens"], threads=16)
fa, fq, ix = (os.path.join(d, n) for n in ("ref.fa", "reads.fq", "ref.idx"))
host.write_fasta(fa, text, off, lens)
bases, _ = host.synth_reads(w["seed"], text, off, lens, n_reads, w["L"], w["e"], first_read=0, threads=16)
host.write_fastq(fq, bases, w["L"], n_reads)
r = subprocess.run([os.path.join(os.getcwd(), "fem_amd", "csrc", "FEM"), "index", "12", "3", fa, ix], capture_output=True, text=True, timeout=600)
assert r.returncode == 0, r.stderr[-300:]
print("files in", d, "e =", w["e"])
