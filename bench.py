#!/usr/bin/env python3
"""bench.py — mapped Mreads/s of the FEM hot path (seeding + candidate filter + banded Myers) on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: bench.py starts its N ranks itself (children made before anything touches the GPU), or runs as one rank when
it is already under torch.distributed.run (RANK / WORLD_SIZE in the environment).  One rank per GPU.

One step = one batch of `--batch` (2.5 M) synthetic reads through the DEVICE PIPELINE of libfemhip.so:
    the batch at two bits per base in the slot's pinned staging (what the FASTQ parser of `FEM map` writes there:
    fem_seqfile_fill_packed; here the generator) --fem_dev_commit_stage_packed: H2D, expanded--> seed/filter kernel(s) +
    verify kernel --D2H--> fem_batch_result
with four batches in flight on four slots and a different batch in every slot (fresh H2D and D2H every step).  No host
work per base happens inside a step, so N ranks need no host cores to speak of.  The same pipeline fed by
fem_dev_map_batch_submit (a caller-owned batch of characters, packed by the library's host threads) and by the zero-copy
character form are measured next to it at N = 1 (`stage_reads_mreads`, `zero_copy_ascii_mreads`).
After the timed steps the pipeline checks itself (`config.pipeline_check`): every repeat of a slot's batch gave the same
five counters, and — at N = 1, where the oracle's index exists — one slot's candidates / edit distances / end offsets of
a 100 k-read prefix, fetched out of the running four-deep pipeline, equal the oracle's.
`value`, `ms_per_step` and `roofline` all come from the SAME workload, the headline one: C3 (100 bp, e=3, 3 Gbp
reference: the HBM-resident configuration SURVEY.md 8d calls bandwidth-relevant; 20 steps x 2.5 M = BASELINE's 50 M
reads).  `value` is the MEDIAN of `--reps` (3) timed runs of exactly K steps each, `ms_per_step` and the kernel times are
that run's; `spread` has all of them (`value_is_repetition`: which one).  C2 (5 Mbp: the index is cache-resident, the pipeline host- and link-balanced) and C5 (150 bp,
e=7) are measured next to it at N = 1 (`pipeline_by_workload`, `roofline_by_workload`).

`roofline` is the dominant kernel of the headline workload — seed_join_kernel — with ITS algorithmic bytes (8 P: the
occurrence entries of the selected seeds) over ITS mean launch time inside the timed pipeline (HIP events on its stream;
it runs beside the next batch's seed_select_kernel there).  `roofline_step` is the figure nothing hides in: all
algorithmic bytes of a step (SURVEY.md 8d's formula) over the driver-visible ms_per_step.  `roofline_by_kernel` has
seed_select_kernel with its own bytes (N L + 16 S N) over its time when it has the chip to itself.
`config.kernel_only_mreads`: the kernels alone on batches already resident in HBM, four slots in rotation (so that,
as in the pipeline, one batch's selection runs beside the previous batch's join).  `e2e_cli`: FASTQ -> SAM.

Workloads (BASELINE.json `configs`):
    c3   24 x 125 Mbp reference, 100 bp, e=3       the headline: `value`, all N
    c2   5 Mbp reference, 100 bp reads, e=3        N = 1 runs it too
    c5   same reference as c3, 150 bp, e=7         N = 1 runs it too (20 steps x 2.5 M = BASELINE's 50 M)
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PROFILE_ROUND = "r05"   # profiles/<round>_<workload>_hbm_traffic.json: the committed PMC passes `roofline.traffic` comes from


def kernel_sources_sha16():
    """Hash of the device sources: a committed profile only speaks for the build it was taken on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "fem_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "fem_amd", "csrc", "*.hip.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md (a device copy reaches ~6.3 TB/s)
N_SLOTS = 4  # batch slots of the library
# batches in flight: all four slots (a dense index has the device work on two batches at once — the selection of one beside
# the join of the previous — so three in flight left the host one batch short now and then: 225-250 against 248-258 Mreads/s)
DEPTH = int(os.environ.get("FEM_BENCH_DEPTH", "4"))
CHECK_READS = 100_000  # prefix of one slot's batch whose full pipeline result is compared with the oracle's (N = 1)
PRIME_TO = 12          # untimed batches a workload has seen before its timed steps (warm-up included), at least ...
# ... and the PROCESS this many before its first timed step: the HIP runtime stalls single calls and waits for 20-50 ms while it
# grows its pools over a process's first ~80 batches (round 5's last day: with 12, one or two of the first workload's three
# timed runs carried such a stall — rep by rep 7.73 7.76 8.36 7.72 7.71 / 7.66 7.72 9.74 7.73 7.65 ms per step —, with 100 none:
# 7.71 7.63 7.66 7.67 7.68 / 7.72 7.64 7.64 7.64 7.66; the later workloads of the same process never showed one)
PRIME_PROCESS_TO = int(os.environ.get("FEM_BENCH_PRIME", "100"))
_batches_seen = [0]    # by this process, any workload

WORKLOADS = {
    "c2": dict(seed=2, seq_lens=[5_000_000], L=100, e=3,
               name="C2: synthetic 100 bp reads, e=3, 5 Mbp random reference, k=12 step=3"),
    "c3": dict(seed=3, seq_lens=[125_000_000] * 24, L=100, e=3,
               name="C3: synthetic 100 bp reads, e=3, 24x125 Mbp random reference, k=12 step=3"),
    "c5": dict(seed=5, seq_lens=[125_000_000] * 24, L=150, e=7,
               name="C5: synthetic 150 bp reads, e=7, 24x125 Mbp random reference, k=12 step=3"),
    # not a BASELINE configuration: C3's reads on a reference with a real genome's kind of k-mer spectrum — 5 % of it 500
    # units of 300 bp in 1 000 copies each (lists of a thousand entries: the generic kernel's share)
    "c3r": dict(seed=3, seq_lens=[125_000_000] * 24, L=100, e=3, repeats=dict(units=500, copies=1000, unit_len=300, rng=5),
                name="C3 reads on a repeat-rich 3 Gbp reference (5 % in 500 units of 300 bp x 1000 copies), k=12 step=3"),
}


def plant_repeats(text, off, lens, units, copies, unit_len, rng):
    """Overwrites `copies` places per unit with the unit (tools/cliff_probe.py's reference): in place."""
    import numpy as np
    g = np.random.default_rng(rng)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for _ in range(units):
        unit = acgt[g.integers(0, 4, unit_len)]
        seq = g.integers(0, len(lens), copies)
        at = g.integers(2000, int(lens[0]) - 2000 - unit_len, copies)
        for s_, a_ in zip(seq, at):
            p_ = int(off[s_]) + int(a_)
            text[p_:p_ + unit_len] = unit
KERNEL_IDS = {"seed": 0, "verify_kernel": 1, "seed_filter_kernel": 2, "seed_select_kernel": 8}


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def self_launch(args):
    """--gpus N without a launcher: start the N ranks as children of a process that has not touched the GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def usable_cores():
    """Host cores this process may really use: the affinity mask, cut down to the cgroup CPU quota when there is one
    (a one-GPU box shows all of the host's CPUs in the mask but schedules only its share)."""
    n_aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    n = n_aff if quota is None else max(1, min(n_aff, int(quota + 0.5)))
    return n, n_aff, quota


class Rank:
    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # FEM_BENCH_FORCE_DIST=1: a process group even for one rank, so that the RCCL reduction of the counters (the path's
        # one exchange) runs on a one-GPU box too (tests/test_gpu_bench.py)
        self.dist = self.world > 1 or os.environ.get("FEM_BENCH_FORCE_DIST") == "1"


def run_workload(key, dev, data, rk, steps, warmup, batch, torch, dist, red_dev, threads, reps=1, compare_forms=False):
    """Pipeline measurement + resident-kernel replay of one workload on this rank's device.  `data` = (text, off, lens)."""
    import numpy as np
    from fem_amd import host
    w = WORKLOADS[key]
    L, e, a, k, step = w["L"], w["e"], 1, 12, 3
    text, off, lens = data
    # a different batch for every slot, in ordinary (pageable) host memory: what a caller of the C ABI owns.  The job is
    # world x N_SLOTS batches of read indices; this rank maps its contiguous range of them (fem_amd/shard.py)
    from fem_amd.shard import reduce_stats, shard_range
    t0 = time.time()
    lo, hi = shard_range(rk.rank, rk.world, rk.world * N_SLOTS * batch)
    assert hi - lo == N_SLOTS * batch
    # the slots' pinned staging, lent by the library, filled ONCE with this rank's four batches at two bits per base (the
    # generator writes the form directly; the FASTQ parser of `FEM map` does the same): a step only commits it again
    for s in range(N_SLOTS):
        hb, _ = dev.acquire_stage(batch, batch * L, slot=s)
        host.synth_reads_packed(w["seed"], text, off, lens, batch, L, e, hb, first_read=lo + s * batch, threads=threads)
    log("rank %d %s: %d x %d reads generated (2-bit, into the pinned staging) in %.1fs" % (rk.rank, key, N_SLOTS, batch, time.time() - t0))
    batches = []  # the same batches as characters in ordinary host memory: made later, for the N = 1 comparison forms

    def fence():
        if rk.dist:
            dist.barrier()
        torch.cuda.synchronize()

    d2h_bytes = [0]
    # fem_dev_fetch_packed (11.4 instead of 30.8 bytes per read home at C2) where the index is sparse: C2 1 015-1 029 -> 1 078-1 092
    # Mreads/s.  On a dense index its packing kernel has to sit in the chain of the batches' kernels (0.07 ms, and the join
    # behind it starts 0.1 ms later): C3 322-327 -> 317-318, so C3 / C5 fetch the plain form.  FEM_BENCH_FETCH=plain|packed forces one.
    packed_fetch = {"plain": False, "packed": True}.get(os.environ.get("FEM_BENCH_FETCH", ""), key == "c2")
    form = ["packed_commit"]
    host_s = [0.0, 0.0]  # seconds this rank's thread spent in the staging call (enqueueing; packing in the stage_reads form) / waiting in the fetch
    seen = [[] for _ in range(N_SLOTS)]  # the five counters of every batch retired, per slot
    keep = {}                            # slot -> True: keep a copy of that slot's next full result (the self-check)
    kept = {}

    def submit(i):
        s = i % N_SLOTS
        t_in = time.perf_counter()
        if form[0] == "packed_commit":
            # fem_dev_commit_stage_packed (include/fem_hip.h): the staging holds the batch at two bits per base; sent,
            # expanded on the device; everything asynchronous, no host work per base
            dev.commit_stage_packed(batch, L, 0, slot=s)
        elif form[0] == "stage_reads":
            # fem_dev_map_batch_submit: the caller's batch of characters is packed to two bits per base into the slot's
            # pinned staging by the library's host threads, then as above
            dev.stage_reads(batches[s][0], batches[s][1], slot=s)
        elif form[0] == "acquire_commit":
            # zero-copy form: the batch already sits in the slot's pinned staging as characters (a parser wrote it there)
            dev.commit_stage(batch, L, slot=s, uniform=True)
        # ("resident": the slot's batch is in HBM already, nothing is sent)
        dev.map_staged(e=e, a=a, k=k, step=step, slot=s)
        host_s[0] += time.perf_counter() - t_in

    def retire(i):
        sl = i % N_SLOTS
        if form[0] == "resident":
            st = dev.fetch_stats(slot=sl)  # nothing but the counters comes back
            seen[sl].append(st.copy())
            return st
        t_in = time.perf_counter()
        if form[0] == "packed_commit" and packed_fetch:
            # fem_dev_fetch_packed: one byte per strand, one offset per 256 strands, 11 bytes per candidate and none per padding
            # slot (include/fem_hip.h) — packed on the device and sent home behind the batch's kernels
            r = dev.fetch_packed(slot=sl, copy=False)
            host_s[1] += time.perf_counter() - t_in
            d2h_bytes[0] = r.d2h_bytes
            seen[sl].append(r.stats.copy())
            if keep.pop(sl, False):
                n_chk = min(batch, CHECK_READS)
                kept[sl] = ("packed", r.count[:2 * n_chk].copy(), r.seg_begin[:(2 * n_chk + 255) // 256].copy(), r.cand.copy(), r.ed.copy(), r.end.copy())
            return r.stats
        r = dev.fetch(slot=sl, copy=False)  # waits; the per-candidate outcome is (or comes) in pinned host memory
        host_s[1] += time.perf_counter() - t_in
        d2h_bytes[0] = 16 * r.n_reads + 11 * r.n_candidates
        seen[sl].append(r.stats.copy())
        if keep.pop(sl, False):
            n_chk = min(batch, CHECK_READS)
            kept[sl] = ("plain", r.cand_begin[:2 * n_chk].copy(), r.cand_count[:2 * n_chk].copy(), r.cand.copy(), r.ed.copy(), r.end.copy())
        return r.stats

    trace = []  # FEM_BENCH_TRACE=1: (what, step, seconds) of every submit / retire of the timed region (fill and drain made visible)

    def pipeline(n, traced=False):
        _batches_seen[0] += n
        tot = np.zeros(5, dtype=np.uint64)
        last = None
        for i in range(n):
            if i >= DEPTH:
                last = retire(i - DEPTH)
                if traced:
                    trace.append(("retire", i - DEPTH, time.perf_counter()))
                tot += last
            submit(i)
            if traced:
                trace.append(("submit", i, time.perf_counter()))
        for i in range(max(0, n - DEPTH), n):
            last = retire(i)
            if traced:
                trace.append(("retire", i, time.perf_counter()))
            tot += last
        return tot, last

    dev.set_timing(True)  # (on during the warm-up as well: the timed steps then differ from it in nothing)
    # The HIP runtime grows its command / signal pools over the first batches of a process's life (two enqueue calls of ~8 ms
    # each around the 8th-10th batch; single stalls of 20-50 ms up to the ~80th: PRIME_PROCESS_TO above): with fewer warm-up
    # steps than that asked for, the difference is run first, untimed and reported as config.priming_steps — it is setup, like
    # the buffer allocations.
    priming = max(0, max(PRIME_TO, PRIME_PROCESS_TO - _batches_seen[0]) - max(warmup, 1))
    if priming:
        pipeline(priming)
    pipeline(max(warmup, 1))
    h2d_bytes, sent_packed = dev.stage_info(0)
    runs = []  # (elapsed seconds, kernel times) of every timed repetition of exactly `steps` steps; the first is `value`
    job = last_stats = None
    host_reps = []
    for rep in range(max(1, reps)):
        dev.reset_timing()
        fence()
        host_s[0] = host_s[1] = 0.0
        t_start = time.perf_counter()
        job_r, last_r = pipeline(steps, traced=(os.environ.get("FEM_BENCH_TRACE") == "1" and rep == 0) or os.environ.get("FEM_BENCH_TRACE") == "2")
        if trace:
            t_ret = [t - t_start for w_, _, t in trace if w_ == "retire"]
            log("%s trace: retire times (ms) %s" % (key, " ".join("%.2f" % (1e3 * t) for t in t_ret)))
            log("%s trace: submit times (ms) %s" % (key, " ".join("%.2f" % (1e3 * (t - t_start)) for w_, _, t in trace if w_ == "submit")))
            log("%s trace: steady step %.3f ms (retires %d..%d)" % (key, 1e3 * (t_ret[-DEPTH - 1] - t_ret[DEPTH]) / max(1, len(t_ret) - 2 * DEPTH - 1), DEPTH, len(t_ret) - DEPTH - 1))
            trace.clear()
        if rk.dist:  # MappingStats reduction (src/FEM_map.c:200-212): the path's one exchange, 40 bytes over RCCL
            job_r = reduce_stats(job_r, device=red_dev)
        fence()
        elapsed = time.perf_counter() - t_start
        if rk.dist:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        runs.append((elapsed, {name: dev.kernel_time(kid) for name, kid in KERNEL_IDS.items()}))
        if reps > 1 and rk.rank == 0:  # (a repetition that falls out of line: was it the kernels or the host?)
            kt_r = runs[-1][1]
            try:  # (... or the job's CPU quota: periods in which the cgroup was stopped for having used it up)
                thr = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().splitlines())
                log("%s rep %d: cgroup cpu.stat nr_throttled %s throttled_usec %s usage_usec %s" % (key, rep, thr.get("nr_throttled"), thr.get("throttled_usec"), thr.get("usage_usec")))
            except Exception:
                pass
            log("%s rep %d: %.2f ms per step; host stage %.3f + fetch %.3f ms per step; kernels %s" % (
                key, rep, 1e3 * elapsed / steps, host_s[0] * 1e3 / steps, host_s[1] * 1e3 / steps,
                " ".join("%s %.3f" % (n_[5:9], t_[0] / max(1, t_[1])) for n_, t_ in kt_r.items() if t_[1])))
        host_reps.append((host_s[0] * 1e3 / steps, host_s[1] * 1e3 / steps))
        if rep == 0:
            job, last_stats = job_r, last_r
            d2h_timed = d2h_bytes[0]  # (of the timed steps' form; the comparison forms below fetch the plain arrays)
    dev.set_timing(False)
    # `value` is the MEDIAN of the timed repetitions (each exactly `steps` steps between fences; all of them in spread.values).
    # Until round 5's last day it was the first one; since then the GPU boxes show one stall of 20-50 ms every half second or so
    # (a host call or a wait that returns late: kernel times unchanged, no cgroup throttling, the previous day's library shows
    # it too), and whichever repetition it lands in comes out 8-20 % low.
    mid = sorted(range(len(runs)), key=lambda i_: runs[i_][0])[len(runs) // 2]
    elapsed, kt = runs[mid]
    host_first = host_reps[mid]

    # ---- the pipeline checks itself (untimed): the same four-deep pipeline once more round the slots, with a copy of slot
    #      0's full result taken out of it; every repeat of a slot's batch, timed steps included, must have given the same
    #      five counters ----
    keep[0] = True
    pipeline(2 * N_SLOTS)
    repeats_identical = all(all(np.array_equal(x, sl[0]) for x in sl) for sl in seen if sl)
    n_repeats = [len(sl) for sl in seen]
    pipe_check = {"repeats_identical": bool(repeats_identical), "batches_retired_per_slot": n_repeats,
                  "what": "every retired batch of a slot (priming, warm-up, timed steps, this check) gave the same five counters"}
    if rk.dist:  # every rank's verdict
        ok_t = torch.tensor([1 if repeats_identical else 0], dtype=torch.int64, device=red_dev)
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        pipe_check["repeats_identical"] = bool(ok_t.item())
    pipe_check["_kept"] = kept.get(0)
    pipe_check["_first_read"] = lo
    pipe_check["_slot0_stats"] = seen[0][0] if seen[0] else None

    stage_reads_rate = zero_copy = None
    h2d_zero_copy = 0
    host_stage_reads = None
    if compare_forms:
        # the same pipeline fed by fem_dev_map_batch_submit: caller-owned batches of characters in ordinary host memory, packed
        # by the library's host threads (round 3's `value`; measured at every N since round 5: whole job, slowest rank's clock)
        for s in range(N_SLOTS):
            batches.append(host.synth_reads(w["seed"], text, off, lens, batch, L, e, first_read=lo + s * batch, threads=threads))
        form[0] = "stage_reads"
        pipeline(N_SLOTS + 2)
        fence()
        host_s[0] = host_s[1] = 0.0
        tz = time.perf_counter()
        n_z = max(4, min(steps, 10))
        pipeline(n_z)
        fence()
        dt_z = time.perf_counter() - tz
        if rk.dist:
            tz_max = torch.tensor([dt_z], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tz_max, op=dist.ReduceOp.MAX)
            dt_z = float(tz_max.item())
        stage_reads_rate = rk.world * batch * n_z / dt_z / 1e6
        host_stage_reads = (host_s[0] * 1e3 / n_z, host_s[1] * 1e3 / n_z)
    if compare_forms == "all":
        # the zero-copy character form (the staging buffers held the packed batches until now)
        form[0] = "acquire_commit"
        for s in range(N_SLOTS):
            hb, ho = dev.acquire_stage(batch, batch * L + 8, slot=s)
            hb[:batch * L] = batches[s][0][:batch * L]
            ho[:batch + 1] = batches[s][1]
        pipeline(N_SLOTS + 2)  # (every slot's buffers exist after this)
        fence()
        tz = time.perf_counter()
        pipeline(n_z)
        fence()
        zero_copy = batch * n_z / (time.perf_counter() - tz) / 1e6
        h2d_zero_copy = dev.stage_info(0)[0]
    batches.clear()

    # the kernels alone on batches already resident in HBM (every slot as staged by the pipeline steps above), the four
    # slots in rotation: as in the pipeline, a batch's seed selection runs beside the previous batch's join
    form[0] = "resident"
    n_rep = max(8, min(steps, 12))
    pipeline(N_SLOTS)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    pipeline(n_rep)
    torch.cuda.synchronize()
    kernel_only = batch * n_rep / (time.perf_counter() - t1) / 1e6
    # ... and one batch at a time (nothing beside anything): every kernel's time when it has the chip to itself
    dev.set_timing(True)
    dev.reset_timing()
    for _ in range(3):
        dev.map_staged(e=e, a=a, k=k, step=step, slot=0)
        dev.fetch_stats(slot=0)
    torch.cuda.synchronize()
    alone = {name: dev.kernel_time(kid) for name, kid in KERNEL_IDS.items()}
    dev.set_timing(False)
    # (the comparison forms and the resident replay mapped the same four batches again: their counters count too)
    local_ok = all(all(np.array_equal(x, sl[0]) for x in sl) for sl in seen if sl)
    pipe_check["repeats_identical"] = bool(pipe_check["repeats_identical"] and local_ok)
    pipe_check["batches_retired_per_slot"] = [len(sl) for sl in seen]

    seed_name = dev.seed_kernel(e=e, a=a, k=k, step=step)

    def per_launch_of(times):
        out = {}
        for name, (ms, n) in times.items():
            if n:
                out[seed_name.split("<")[0] if name == "seed" else name] = (ms / n, n)
        return out

    per_launch = {n_: (v[0], v[1] / steps) for n_, v in per_launch_of(kt).items()}
    alone_ms = {n_: v[0] for n_, v in per_launch_of(alone).items()}
    # algorithmic bytes: B = N*L + 16*(L-k+1)*N + 8*P + (L+2e)*C + 16*M, from the path's own counters
    N, P, Cn, M = batch, int(last_stats[2]), int(last_stats[3]), int(last_stats[4])
    S = L - k + 1
    select_bytes = N * L + 16 * S * N            # read bases + one 8-byte lookup pair per seed and strand
    join_bytes = 8 * P                           # occurrence entries of the selected seeds
    seed_bytes = select_bytes + join_bytes
    verify_bytes = (L + 2 * e) * Cn + 16 * M     # reference window per verification + result record
    split = "seed_select_kernel" in per_launch   # dense index: the seed path is two kernels
    bytes_of = {"seed_select_kernel": select_bytes, "seed_join_kernel": join_bytes if split else seed_bytes,
                "seed_fast_kernel": seed_bytes, "seed_filter_kernel": seed_bytes, "verify_kernel": verify_bytes}
    cand = [n_ for n_ in per_launch if n_ in bytes_of and n_ != "seed_select_kernel"]
    dominant = max(cand, key=lambda n_: per_launch[n_][0] * per_launch[n_][1])
    dom_ms, dom_launches = per_launch[dominant]

    def roof_of(name, ms, nbytes, what):
        ach = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"bound": "hbm", "kernel": name, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None, "wire_frac": None, "traffic_source": None,
                "algorithmic_bytes_per_launch": int(nbytes), "avg_launch_ms": round(ms, 4), "time_is": what}

    roof = roof_of(dominant, dom_ms, bytes_of[dominant] / max(dom_launches, 1.0),
                   "mean launch duration inside the timed pipeline (HIP events on the kernel's stream)")
    roof["bytes_are"] = ("8 P: the occurrence entries of the selected seeds (SURVEY.md 8d's third term)" if split and dominant == "seed_join_kernel"
                         else "SURVEY.md 8d's terms for this kernel")
    if split and dominant == "seed_join_kernel":
        # what this implementation needs to move for the same work: 4-byte entries of the 32-bit occurrence table + the
        # selection's hand-over (8 bytes per selected seed, 6 R per read) — NOT what `achieved` / `frac` are computed from
        R = e + 1 + a
        impl = 4 * P + 8 * 6 * R * N + 8 * N
        roof["implementation_bytes_per_launch"] = int(impl / max(dom_launches, 1.0))
        roof["implementation_gbs"] = round(impl / max(dom_launches, 1.0) / (dom_ms * 1e-3) / 1e9, 2) if dom_ms > 0 else 0.0
    by_kernel = {dominant: roof}
    for n_ in per_launch:
        if n_ in bytes_of and n_ != dominant and n_ in alone_ms and n_ != "seed_filter_kernel":
            by_kernel[n_] = roof_of(n_, alone_ms[n_], bytes_of[n_], "launch duration with the chip to itself (resident replay, one batch at a time)")
    ms_step = elapsed * 1e3 / steps
    step_roof = roof_of("whole step: every kernel of one batch", ms_step * rk.world, (seed_bytes + verify_bytes) * rk.world,
                        "driver-visible ms_per_step (copies, gaps, fill and drain included)")
    step_roof["achieved"] = round((seed_bytes + verify_bytes) / (ms_step * 1e-3) / 1e9, 2)  # per GPU
    step_roof["frac"] = round(step_roof["achieved"] / HBM_PEAK_GBS, 5)
    step_roof["algorithmic_bytes_per_launch"] = int(seed_bytes + verify_bytes)
    step_roof["avg_launch_ms"] = round(ms_step, 4)
    # HBM-side bytes of the dominant kernel: NOT measured in this run.  They come from the committed rocprofv3 --pmc passes
    # of this same command (profiles/r03_<wl>_hbm_traffic.json: FETCH_SIZE + WRITE_SIZE, corrected by the calibration of
    # profiles/r03_fetch_calibration.json for this kernel's access pattern), scaled to this batch size.
    tname = "%s_%s_hbm_traffic.json" % (PROFILE_ROUND, key)
    tpath = os.path.join(ROOT, "profiles", tname)
    if os.path.exists(tpath):
        try:
            prof = json.load(open(tpath))
            if prof.get("kernel_sources_sha16") != kernel_sources_sha16():
                # the kernels changed since the PMC passes were taken: their bytes say nothing about this build
                roof["traffic_source"] = ("profiles/%s was collected on other kernel sources (%s, now %s): run profiles/make_profiles.py"
                                          % (tname, prof.get("kernel_sources_sha16"), kernel_sources_sha16()))
            else:
                for kname, ctr in prof["kernels"].items():
                    if dominant in kname and "hbm_bytes_per_launch" in ctr:
                        roof["traffic"] = int(ctr["hbm_bytes_per_launch"] * (batch / max(dom_launches, 1.0)) / prof["reads_per_launch"])
                        # what the counters saw on the wires over the same launch time, against the same peak
                        roof["wire_frac"] = round(roof["traffic"] / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if dom_ms > 0 else None
                        roof["traffic_source"] = ("committed profile profiles/%s of these kernel sources (FETCH_SIZE x %.2f + WRITE_SIZE: "
                                                  "calibrated on this kernel's access pattern), rescaled (not measured in this run)"
                                                  % (tname, ctr.get("fetch_factor", 1.0)))
        except Exception as ex:  # a malformed profile file must not break the measurement
            log("could not read %s: %s" % (tpath, ex))
    values = [rk.world * batch * steps / r[0] / 1e6 for r in runs]
    value = values[mid]
    return {
        "workload": w["name"], "value": round(value, 3), "ms_per_step": round(ms_step, 3), "steps": steps,
        "spread": {"reps": len(values), "min": round(min(values), 3), "median": round(sorted(values)[len(values) // 2], 3),
                   "max": round(max(values), 3), "values": [round(v, 3) for v in values], "value_is_repetition": mid},
        "reads_per_step_per_gpu": batch, "read_len": L, "e": e, "a": a, "k": k, "step": step,
        "kernel_only_mreads": round(kernel_only, 3), "seed_kernel": seed_name, "index_tables": dev.index_info(),
        "h2d_bytes_per_step": int(h2d_bytes), "h2d_packed": bool(sent_packed), "d2h_bytes_per_step": d2h_timed,
        "fetch_form": "fem_dev_fetch_packed" if packed_fetch else "fem_dev_fetch",
        "stage_reads_mreads": None if stage_reads_rate is None else round(stage_reads_rate, 3),
        "zero_copy_ascii_mreads": None if zero_copy is None else round(zero_copy, 3), "zero_copy_h2d_bytes_per_step": int(h2d_zero_copy),
        "priming_steps": priming, "pipeline_check": pipe_check,
        "distinct_reads": "%d distinct reads per GPU (%d slots x %d), each slot's batch mapped again every %d steps"
                          % (N_SLOTS * batch, N_SLOTS, batch, N_SLOTS),
        # what this rank's host thread does per step: the staging call (2-bit packing of the batch on the library's threads,
        # enqueueing the copies and kernels) and the wait inside the fetch (device not done yet, or results on their way);
        # packing reads batch x L bytes of host memory and writes a quarter of that
        "host_ms_per_step": {"stage_call": round(host_first[0], 3), "fetch_wait": round(host_first[1], 3),
                             "form": "fem_dev_commit_stage_packed + fem_dev_map_staged: enqueues only, no host work per base",
                             "stage_reads_form": None if host_stage_reads is None else
                             {"stage_call": round(host_stage_reads[0], 3), "fetch_wait": round(host_stage_reads[1], 3),
                              "stage_threads": int(os.environ.get("FEM_STAGE_THREADS", "12"))}},
        "counters": {"reads": int(job[0]), "mapped_reads": int(job[1]), "pre_filter": int(job[2]),
                     "candidates": int(job[3]), "mappings": int(job[4])},
        "counters_last_step_per_gpu": [int(x) for x in last_stats],
        "algorithmic_bytes_per_step_per_gpu": seed_bytes + verify_bytes,
        "kernel_ms_per_launch": {n_: round(v[0], 4) for n_, v in per_launch.items()},
        "kernel_launches_per_step": {n_: round(v[1], 2) for n_, v in per_launch.items()},
        "kernel_ms_alone": {n_: round(v, 4) for n_, v in alone_ms.items()},
        "roofline": roof, "roofline_step": step_roof, "roofline_by_kernel": by_kernel,
    }


def profile_replay(key, dev, data, n, batch, torch, threads):
    """What `rocprofv3 --kernel-trace` should see when the kernel times of the bench line are checked: the four slots'
    batches staged once, then nothing but kernels (and the 40-byte counter copies) for n steps."""
    from fem_amd import host
    w = WORKLOADS[key]
    L, e = w["L"], w["e"]
    text, off, lens = data
    for s in range(N_SLOTS):
        b, o = host.synth_reads(w["seed"], text, off, lens, batch, L, e, first_read=s * batch, threads=threads)
        dev.stage_reads(b, o, slot=s)
        dev.map_staged(e=e, a=1, k=12, step=3, slot=s)
    for s in range(N_SLOTS):
        dev.fetch_stats(slot=s)

    def replay(m):
        for i in range(m):
            if i >= DEPTH:
                dev.fetch_stats(slot=(i - DEPTH) % N_SLOTS)
            dev.map_staged(e=e, a=1, k=12, step=3, slot=i % N_SLOTS)
        for i in range(max(0, m - DEPTH), m):
            dev.fetch_stats(slot=i % N_SLOTS)

    replay(2 * N_SLOTS)  # warm-up launches: `warmup_launches_per_kernel` of every kernel's trace rows
    torch.cuda.synchronize()
    dev.set_timing(True)
    dev.reset_timing()
    t0 = time.perf_counter()
    replay(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dev.set_timing(False)
    seed_name = dev.seed_kernel(e=e, a=1, k=12, step=3).split("<")[0]
    times = {}
    for name, kid in KERNEL_IDS.items():
        ms, cnt = dev.kernel_time(kid)
        if cnt:
            times[seed_name if name == "seed" else name] = {"launches": int(cnt), "mean_ms": round(ms / cnt, 4)}
    return {"mode": "resident replay, slots in rotation, %d batches in flight" % DEPTH, "workload": key, "steps": n,
            "reads_per_launch": batch, "warmup_launches_per_kernel": 3 * N_SLOTS, "mreads_per_s": round(batch * n / dt / 1e6, 2),
            "event_times": times}


def check_prefix_against_oracle(fo, ref, idx, w, data, pipe_check, threads):
    """One slot's result out of the running pipeline (run_workload kept a copy) against the oracle on a prefix of that
    slot's batch: candidates, edit distances, end offsets per (read, strand), array for array."""
    import numpy as np
    from fem_amd import host
    kept = pipe_check.get("_kept")
    if kept is None:
        return None
    text, off, lens = data
    if kept[0] == "packed":  # fem_batch_packed: counts per strand, where each segment of 256 strands starts
        _, count8, seg_begin, cand, ed, end = kept
        assert not np.any(count8 == 255)
        n_seg = len(seg_begin)
        pad = np.zeros(n_seg * 256, dtype=np.int64)
        pad[:len(count8)] = count8
        within = np.cumsum(pad.reshape(n_seg, 256), axis=1) - pad.reshape(n_seg, 256)
        cand_begin = (seg_begin.astype(np.int64)[:, None] + within).reshape(-1)[:len(count8)]
        cand_count = count8.astype(np.uint32)
    else:
        _, cand_begin, cand_count, cand, ed, end = kept
    n_chk = len(cand_begin) // 2
    bases, offsets = host.synth_reads(w["seed"], text, off, lens, n_chk, w["L"], w["e"], first_read=pipe_check["_first_read"], threads=threads)
    want = fo.map_reads(ref, idx, fo.ReadBatch.from_arrays(bases, offsets), e=w["e"], a=1, threads=threads, stages=fo.STAGE_SEED | fo.STAGE_VERIFY)
    cnt = cand_count.astype(np.int64)
    o = np.zeros(len(cnt) + 1, np.uint64)
    o[1:] = np.cumsum(cnt)
    sel = np.arange(int(o[-1]), dtype=np.int64) + np.repeat(cand_begin.astype(np.int64) - o[:-1].astype(np.int64), cnt)
    ok = (np.array_equal(o, want.cand_off) and np.array_equal(cand[sel], want.cands) and np.array_equal(ed[sel], want.v_ed)
          and np.array_equal(end[sel][ed[sel] != 0xFF], want.v_end[want.v_ed != 0xFF]))
    return {"reads": n_chk, "candidates": int(o[-1]), "mappings": int(np.count_nonzero(ed[sel] != 0xFF)), "equal_to_oracle": bool(ok)}


def cpu_baseline(w, data, n_sample, dev, threads, label="C2", pipe_check=None, also_check=()):
    """The oracle (CPU restatement of the reference, 'port') timed on this box's host cores on a bounded sample of the
    same workload, same stages as the device path (seeding + filter + verification).  Checker, never shipped."""
    import numpy as np
    from fem_amd import host
    from oracle import fem_oracle as fo
    text, off, lens = data
    L, e = w["L"], w["e"]
    bases, offsets = host.synth_reads(w["seed"], text, off, lens, n_sample, L, e, first_read=0, threads=threads)
    ref = fo.Reference([text[int(o):int(o) + int(l)].tobytes() for o, l in zip(off, lens)])
    t0 = time.time()
    idx = fo.OracleIndex(ref, threads=threads if len(text) > 500_000_000 else 1)  # (threaded build: same arrays, BASELINE-sized references)
    t_index = time.time() - t0
    sample = fo.ReadBatch.from_arrays(bases, offsets)
    t0 = time.perf_counter()
    h = fo.map_reads(ref, idx, sample, e=e, a=1, threads=threads, stages=fo.STAGE_SEED | fo.STAGE_VERIFY, keep_handle=True)
    dt = time.perf_counter() - t0
    st = np.zeros(5, np.uint64)
    fo.lib().fo_result_stats(h, st.ctypes.data)
    fo.free_result(h)
    got = dev.map_batch(bases, offsets, e=e, a=1, slot=1).stats  # the same sample through the device path
    if pipe_check is not None:
        pipe_check["prefix_vs_oracle"] = check_prefix_against_oracle(fo, ref, idx, w, data, pipe_check, threads)
    for w_other, pc_other in also_check:  # other workloads on the same reference (C5 beside C3): the oracle's index is at hand
        if pc_other is not None:
            pc_other["prefix_vs_oracle"] = check_prefix_against_oracle(fo, ref, idx, w_other, data, pc_other, threads)
    return {"value": round(n_sample / dt / 1e6, 4), "unit": "Mreads/s", "cores": threads, "kind": "port",
            "sample": "%d reads of the %s workload, seeding+filter+verification, %d threads (every core this process may use)" % (n_sample, label, threads),
            "seconds": round(dt, 3), "index_build_seconds": round(t_index, 2),
            "counters_match_device": bool(np.array_equal(got, st))}


def e2e_cli(w, data, n_reads, threads, label="C2"):
    """`FEM index` + `FEM map` on generated FASTA / FASTQ files: the mapping-phase time the reference prints itself
    ("Time:", src/FEM_map.c:172,219), FASTQ -> SAM."""
    import re
    from fem_amd import host
    text, off, lens = data
    L, e = w["L"], w["e"]
    exe = os.path.join(ROOT, "fem_amd", "csrc", "FEM")
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    need = int(len(text) * 1.02) + 9 * (len(text) // 3) + n_reads * (2 * L + 16)  # FASTA + index file + FASTQ
    if base and shutil.disk_usage(base).free < need + (4 << 30):
        base = None  # (inputs then sit behind the page cache of the ordinary temporary directory)
    if shutil.disk_usage(base or tempfile.gettempdir()).free < need + (2 << 30):
        return {"error": "not enough room for %.1f GB of input files" % (need / 1e9)}
    d = tempfile.mkdtemp(prefix="fem_e2e_", dir=base)
    # the SAM file goes to the ordinary temporary directory (a disk file system behind the page cache: what a user writes
    # to), the inputs sit in memory.  (tmpfs takes 6 GB/s from one writer, the page cache of the GPU box's /tmp 11 GB/s:
    # 24-26 against 37-38 Mreads/s for this run, which the writer bounds in both cases.)
    d_out = tempfile.mkdtemp(prefix="fem_e2e_out_")
    try:
        fa, fq, ix = (os.path.join(d, n) for n in ("ref.fa", "reads.fq", "ref.idx"))
        sam = os.path.join(d_out, "out.sam")
        host.write_fasta(fa, text, off, lens)
        bases, _ = host.synth_reads(w["seed"], text, off, lens, n_reads, L, e, first_read=0, threads=threads)
        host.write_fastq(fq, bases, L, n_reads)
        t_ix = time.perf_counter()
        r = subprocess.run([exe, "index", "12", "3", fa, ix], capture_output=True, text=True, timeout=600)
        t_ix = time.perf_counter() - t_ix
        if r.returncode != 0:
            return {"error": "FEM index failed: " + r.stderr[-300:]}
        env = dict(os.environ, FEM_STAGE_TIMES="1")
        load = [None]
        cores = [None]  # host cores busy on average during the mapping phase (user + system time / its wall time)

        def run_map(out_path):
            t0 = time.perf_counter()
            r = subprocess.run([exe, "map", "-e", str(e), "-t", str(threads), "--ref", fa, "--index", ix, "--read1", fq, "-o", out_path],
                               capture_output=True, text=True, timeout=600, env=env)
            wall = time.perf_counter() - t0
            if r.returncode != 0:
                return None, wall, "FEM map failed: " + r.stderr[-300:]
            m = re.search(r"Time: ([0-9.]+)s", r.stderr)
            st = re.search(r"stage busy seconds: (.*)", r.stderr)
            ld = re.search(r"resident on \d+ GPUs? in ([0-9.]+)s", r.stderr)
            load[0] = float(ld.group(1)) if ld else None
            cpu = re.search(r"host CPU in the mapping phase: .* = ([0-9.]+) cores", r.stderr)
            cores[0] = float(cpu.group(1)) if cpu else None
            return (float(m.group(1)) if m else None), wall, (st.group(1) if st else None)

        secs, wall, busy = run_map(sam)
        if secs is None:
            return {"error": busy}
        file_cores = cores[0]
        sam_bytes = os.path.getsize(sam)
        os.unlink(sam)
        # what the output directory's file system takes from plain write() calls — one writer (what FEM map has, like the
        # reference's output queue, src/output_queue.c:60-91) and four — so that the to-file figure can be read against it
        def write_rate(n_writers, one_file, total=min(sam_bytes, 2 << 30)):
            import threading
            buf = bytes(8 << 20)
            paths = [os.path.join(d_out, "ceiling_%d.bin" % (0 if one_file else i)) for i in range(n_writers)]
            fds = [os.open(p_, os.O_WRONLY | os.O_CREAT, 0o666) for p_ in paths]
            share = total // n_writers
            def one(fd, at, nbytes):
                done = 0
                while done < nbytes:
                    done += os.pwrite(fd, buf[:min(len(buf), nbytes - done)], at + done)
            t0 = time.perf_counter()
            ts = [threading.Thread(target=one, args=(fds[i], i * share if one_file else 0, share)) for i in range(n_writers)]
            [t.start() for t in ts]
            [t.join() for t in ts]
            dt = time.perf_counter() - t0
            [os.close(fd) for fd in fds]
            for p_ in set(paths):
                os.unlink(p_)
            return total / dt / 1e9
        fs_gbs = {"one_writer_gbs": round(write_rate(1, True), 2), "four_writers_one_file_gbs": round(write_rate(4, True), 2),
                  "four_writers_four_files_gbs": round(write_rate(4, False), 2)}
        fs_gbs["one_writer_as_mreads"] = round(fs_gbs["one_writer_gbs"] * 1e9 / (sam_bytes / n_reads) / 1e6, 1)
        fs_gbs["what"] = ("pwrite() of zeros into the SAM file's directory, 8 MiB at a time: one writer, four into ONE file (what a SAM file is: "
                          "buffered writes take the inode's lock), four into four files; as_mreads = one writer at this run's SAM bytes per read")
        # the same run with the SAM text discarded: what the host stages do when no file system is in the way
        null_secs, _, null_busy = run_map("/dev/null")
        null_cores = cores[0]
        null_again, _, _ = run_map("/dev/null")
        return {"value": round(n_reads / secs / 1e6, 3), "unit": "Mreads/s",
                "what": "FEM map mapping phase (its own 'Time:' line, src/FEM_map.c:172,219): FASTQ parse -> device -> SAM text -> file, "
                        "%d reads of %s, -t %d, inputs in %s, SAM file in %s" % (n_reads, label, threads, base or "tmp", d_out),
                "seconds": secs, "wall_seconds_incl_load": round(wall, 3), "sam_bytes": sam_bytes, "stage_busy": busy,
                "fem_index_wall_seconds": round(t_ix, 2), "reference_and_index_resident_seconds": load[0],
                "file_system_write_ceiling": fs_gbs,
                "host_cores_busy": file_cores,
                "to_dev_null": {"value": round(n_reads / null_secs / 1e6, 3) if null_secs else None, "seconds": null_secs,
                                "stage_busy": null_busy, "host_cores_busy": null_cores,
                                "second_run": round(n_reads / null_again / 1e6, 3) if null_again else None}}
    finally:
        shutil.rmtree(d, ignore_errors=True)
        shutil.rmtree(d_out, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS), help="the workload `value` is measured on")
    ap.add_argument("--reps", type=int, default=3, help="timed runs of exactly --steps steps each (value = their median; spread = all)")
    ap.add_argument("--batch", type=int, default=2_500_000, help="reads per step and GPU")
    ap.add_argument("--extra", default="auto", help="comma list of further workloads measured on rank 0 when N = 1 "
                                                    "(auto = c5,c3r,c2 next to c3; none)")
    ap.add_argument("--extra-steps", type=int, default=20)
    ap.add_argument("--cpu-sample", type=int, default=4_000_000, help="reads of the C2 workload timed on the host cores")
    ap.add_argument("--cpu-sample-c3", type=int, default=2_000_000, help="reads of the C3 workload timed on the host cores (0 = skip; "
                                                                         "the oracle's 3 Gbp index takes ~30 s to build)")
    ap.add_argument("--e2e-reads", type=int, default=32_000_000, help="reads of the end-to-end FEM map run on C2 (0 = skip)")
    ap.add_argument("--e2e-reads-c3", type=int, default=32_000_000, help="reads of the end-to-end FEM map run on the headline configuration "
                                                                        "(3 GB FASTA + 8 GB index file + 7 GB FASTQ in /dev/shm; 0 = skip)")
    ap.add_argument("--profile-replay", type=int, default=0, help="profiling aid: only the resident replay of --workload (kernels on batches "
                                                                     "already in HBM, slots in rotation, no copies in flight), this many steps; "
                                                                     "prints the HIP-event means of exactly those launches")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-compare", action="store_true", help="skip the stage_reads / zero-copy comparison forms (N = 1 runs them by default)")
    ap.add_argument("--no-e2e", action="store_true")
    args = ap.parse_args()
    if args.steps < 1 or args.warmup < 0 or args.batch < 1:
        sys.exit("bench.py: --steps >= 1, --warmup >= 0, --batch >= 1")
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    if args.profile_replay > 0:
        # the profiled replay launches every batch whole: a batch that meets an idle device is mapped in two parts otherwise
        # (fem_hip.hip launch_batch), and a profile's "average duration per launch" would mix halves with wholes
        os.environ["FEM_TESTING"], os.environ["FEM_NO_PARTS"] = "1", "1"

    rk = Rank()
    if rk.world != args.gpus:
        args.gpus = rk.world
    import numpy as np
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the hot path has no CPU fallback")
    # host placement: this rank's threads and pinned buffers next to its GPU, before any thread pool exists (after torch
    # has loaded its HIP runtime: the other order leaves torch without a device)
    from fem_amd import device as _fem_device
    numa_bound = _fem_device.bind_near_device(0 if os.environ.get("FEM_BENCH_ONE_GPU") == "1" else rk.local_rank)
    # Rehearsal hooks for a one-GPU box (never set by the driver): FEM_BENCH_ONE_GPU=1 puts every rank on GPU 0,
    # FEM_BENCH_BACKEND=gloo reduces the counters over gloo instead of RCCL (RCCL refuses two ranks on one GPU).
    backend = os.environ.get("FEM_BENCH_BACKEND", "nccl")
    local_rank = 0 if os.environ.get("FEM_BENCH_ONE_GPU") == "1" else rk.local_rank
    torch.cuda.set_device(local_rank)
    if rk.dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rk.rank, world_size=rk.world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rk.rank, world_size=rk.world)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    from fem_amd import Device, host
    cores, n_aff, quota = usable_cores()
    if "FEM_BENCH_THREADS" in os.environ:
        cores = max(1, int(os.environ["FEM_BENCH_THREADS"]))
    threads = max(1, cores // (rk.world if os.environ.get("FEM_BENCH_ONE_GPU") == "1" else 1))
    gen_threads = max(1, min(threads, 32))
    if rk.world > 1 and "FEM_STAGE_THREADS" not in os.environ:
        # the library packs a batch on 12 threads by default (what a 16-core share feeds); N ranks on one node share the
        # node's cores: each takes its part of them, so that eight ranks do not put 96 threads on a 16-core quota
        os.environ["FEM_STAGE_THREADS"] = str(max(2, min(12, cores // rk.world)))
    log("rank %d: %d host threads (affinity mask %d CPUs%s, cgroup quota %s)" %
        (rk.rank, threads, n_aff, ", bound to the GPU's NUMA node" if numa_bound else "", quota))

    extras = []
    if rk.world == 1 and args.extra != "none":
        extras = [x for x in (["c5", "c3r", "c2"] if args.extra == "auto" else args.extra.split(",")) if x in WORKLOADS and x != args.workload]
        if args.extra == "auto" and args.workload != "c3":
            extras = []

    results, data_cache = {}, {}
    dev = None
    bw = {}
    cpu_c3 = [None]
    e2e_c3 = [None]

    def cpu_c3_now():
        """The headline configuration on the host cores, while the 3 Gbp reference is at hand (host text + device)."""
        w3 = WORKLOADS["c3"]
        ref_key3 = (3, tuple(w3["seq_lens"]), False)
        if cpu_c3[0] is None and not args.no_cpu and rk.world == 1 and "c3" in results and args.cpu_sample_c3 > 0 and ref_key3 in data_cache:
            cpu_c3[0] = cpu_baseline(w3, data_cache[ref_key3][:3], args.cpu_sample_c3, dev, threads, label="C3",
                                     pipe_check=results["c3"]["pipeline_check"],
                                     also_check=[(WORKLOADS["c5"], results["c5"]["pipeline_check"])] if "c5" in results else ())
            cpu_c3[0].update(affinity_cpus=n_aff, cgroup_cpu_quota=quota,
                             device_pipeline_over_cpu=round(results["c3"]["value"] / max(cpu_c3[0]["value"], 1e-9), 1))
        # ... and `FEM index` + `FEM map` end to end on the headline configuration (its own process and handle)
        if e2e_c3[0] is None and not args.no_e2e and rk.world == 1 and "c3" in results and args.e2e_reads_c3 > 0 and ref_key3 in data_cache:
            try:
                e2e_c3[0] = e2e_cli(w3, data_cache[ref_key3][:3], args.e2e_reads_c3, threads, label="C3")
            except Exception as ex:
                e2e_c3[0] = {"error": repr(ex)}

    for key in [args.workload] + extras:
        w = WORKLOADS[key]
        ref_key = (w["seed"] if key == "c2" else 3, tuple(w["seq_lens"]), "repeats" in w)  # c3 and c5 share one reference (seed 3)
        if ref_key not in data_cache:
            cpu_c3_now()
            t0 = time.time()
            data_cache.clear()
            if dev is not None:
                dev.close()
            text, off, lens = host.synth_reference(ref_key[0], w["seq_lens"], threads=gen_threads)
            if "repeats" in w:
                plant_repeats(text, off, lens, **w["repeats"])
            dev = Device(local_rank)
            dev.upload_reference([text[int(o):int(o) + int(l)] for o, l in zip(off, lens)])
            n_occ, _, _ = dev.build_index(12, 3, fetch=False)
            data_cache[ref_key] = (text, off, lens, n_occ)
            log("rank %d %s: reference %d bp in %d sequences, index %d entries built on device in %.1fs"
                % (rk.rank, key, int(lens.astype(np.uint64).sum()), len(lens), n_occ, time.time() - t0))
            if not bw:
                bw = {"device_copy_gbs": round(dev.copy_bandwidth(1 << 30, 10), 1), "pinned_h2d_gbs": round(dev.h2d_bandwidth(1 << 28, 8), 1)}
        text, off, lens, n_occ = data_cache[ref_key]
        steps, warmup = (args.steps, args.warmup) if key == args.workload else (args.extra_steps, N_SLOTS + 1)  # (every slot warm)
        if "repeats" in w and key != args.workload:
            steps = min(steps, 8)  # (a step of c3r is ~60 ms and sends 1.1 GB of candidates home)
        if args.profile_replay > 0:
            print(json.dumps(profile_replay(key, dev, (text, off, lens), args.profile_replay, args.batch, torch, gen_threads)), flush=True)
            dev.close()
            return
        res = run_workload(key, dev, (text, off, lens), rk, steps, warmup, args.batch, torch, dist, red_dev, gen_threads,
                           reps=1 if "repeats" in w else args.reps,
                           compare_forms=None if args.no_compare or "repeats" in w else ("all" if rk.world == 1 else "stage_reads"))
        res["index_entries"] = n_occ
        results[key] = res
        log("rank %d %s: pipeline %.1f Mreads/s, kernels only %.1f, %s" % (rk.rank, key, res["value"], res["kernel_only_mreads"], res["kernel_ms_per_launch"]))

    cpu_c3_now()
    if rk.rank != 0:
        if rk.dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    head = results[args.workload]
    out = {
        "metric": "mapped Mreads/s (100 bp, e=3) at 1/2/4/8 MI355X + achieved HBM GB/s vs roofline",
        "value": head["value"], "unit": "Mreads/s", "n_gpus": rk.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        # (round 4 changed what `value` is fed by — see config.value_is; the round-3 form, the library packing the caller's
        #  characters on host threads, is pipeline_by_workload.*.stage_reads_mreads, measured at every N)
        "value_form": "packed_commit", "value_stage_reads_form": head.get("stage_reads_mreads"),
        "dtype": "u32/u64 integer + 32-bit Myers bit-vectors", "data": "synthetic", "spread": head["spread"],
        "config": dict({k_: v_ for k_, v_ in head.items() if k_ not in ("value", "ms_per_step", "steps", "roofline", "roofline_step",
                                                                         "roofline_by_kernel", "spread")},
                       value_is="device pipeline: the batch at two bits per base in the slot's pinned staging (as the FASTQ parser of FEM map "
                                "writes it) -> fem_dev_commit_stage_packed (H2D, expansion) -> kernels -> D2H of fem_batch_result, %d batches "
                                "in flight, a different batch per slot, fresh H2D and D2H every step; no host work per base; the median of %d timed "
                                "runs of exactly %d steps each (spread.values)" % (DEPTH, max(1, args.reps), args.steps),
                       parallelism="reads sharded x%d, index replicated" % rk.world, bandwidths=bw),
        "roofline": dict(head["roofline"], workload=args.workload),
        "roofline_step": dict(head["roofline_step"], workload=args.workload),
        "roofline_by_kernel": head["roofline_by_kernel"],
        "roofline_by_workload": {k_: {"dominant_kernel": v_["roofline"], "step": v_["roofline_step"]} for k_, v_ in results.items()},
        "pipeline_by_workload": {k_: {x: v_[x] for x in ("value", "ms_per_step", "steps", "kernel_only_mreads", "seed_kernel", "index_tables",
                                                           "reads_per_step_per_gpu", "kernel_ms_per_launch", "kernel_ms_alone", "counters_last_step_per_gpu",
                                                           "algorithmic_bytes_per_step_per_gpu", "h2d_bytes_per_step", "h2d_packed", "d2h_bytes_per_step", "fetch_form",
                                                           "stage_reads_mreads", "zero_copy_ascii_mreads", "zero_copy_h2d_bytes_per_step", "priming_steps",
                                                           "host_ms_per_step", "spread", "pipeline_check")}
                                 for k_, v_ in results.items()},
    }
    if cpu_c3[0] is not None:
        out["cpu_baseline"] = cpu_c3[0]
    c2_data = None
    if (not args.no_cpu or not args.no_e2e) and rk.world == 1:
        w2 = WORKLOADS["c2"]
        if (w2["seed"], tuple(w2["seq_lens"]), False) in data_cache:
            c2_data = data_cache[(w2["seed"], tuple(w2["seq_lens"]), False)][:3]
        else:
            dev.close()
            t2 = host.synth_reference(w2["seed"], w2["seq_lens"], threads=gen_threads)
            dev = Device(local_rank)
            dev.upload_reference([t2[0][int(o):int(o) + int(l)] for o, l in zip(t2[1], t2[2])])
            dev.build_index(12, 3, fetch=False)
            c2_data = t2
    if not args.no_cpu and rk.world == 1:
        key_cb = "cpu_baseline_c2" if "cpu_baseline" in out else "cpu_baseline"
        out[key_cb] = cpu_baseline(WORKLOADS["c2"], c2_data, args.cpu_sample, dev, threads,
                                   pipe_check=results["c2"]["pipeline_check"] if "c2" in results else None)
        out[key_cb].update(affinity_cpus=n_aff, cgroup_cpu_quota=quota)
    dev.close()
    if not args.no_e2e and rk.world == 1 and args.e2e_reads > 0:
        try:
            out["e2e_cli"] = e2e_cli(WORKLOADS["c2"], c2_data, args.e2e_reads, threads)
        except Exception as ex:
            out["e2e_cli"] = {"error": repr(ex)}
    for v_ in results.values():
        for k_ in [k_ for k_ in v_["pipeline_check"] if k_.startswith("_")]:
            del v_["pipeline_check"][k_]
    pc = head["pipeline_check"]
    out["counters_match_pipeline"] = bool(pc["repeats_identical"] and (pc.get("prefix_vs_oracle") or {"equal_to_oracle": True})["equal_to_oracle"])
    if e2e_c3[0] is not None:
        out["e2e_cli_c3"] = e2e_c3[0]
    out["config"]["counter_reduction"] = ("torch.distributed all_reduce, backend %s, %d rank(s)" % (backend, rk.world)) if rk.dist else "single process: none"
    print(json.dumps(out), flush=True)
    if rk.dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
