#!/usr/bin/env python3
"""bench.py — mapped Mreads/s of the FEM hot path (seeding + candidate filter + banded Myers) on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

One step = one pass of the device hot path (libfemhip.so: seed/filter kernel + verify kernel) over one batch of
synthetic reads that is already resident in HBM, plus the reduction of the five MappingStats counters over ranks
(RCCL through torch.distributed when N > 1).  Reads shard over ranks (weak scaling: every rank maps its own
`reads_per_gpu` reads, global read index = rank * reads_per_gpu + i); the index and reference are replicated.

Workloads (BASELINE.json `configs`):
    c2 (default)  5 Mbp reference, 10 M x 100 bp reads, e=3   — the configuration the metric is quoted on
    c3            24 x 125 Mbp reference, 100 bp reads, e=3   — HBM-resident index (opt-in: --workload c3)
    c5            24 x 125 Mbp reference, 150 bp reads, e=7   — (opt-in: --workload c5)
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md (6.3 TB/s achievable by a copy)

WORKLOADS = {
    "c2": dict(seed=2, seq_lens=[5_000_000], reads=10_000_000, L=100, e=3,
               name="C2: 10M synthetic 100 bp reads, e=3, 5 Mbp random reference, k=12 step=3"),
    "c3": dict(seed=3, seq_lens=[125_000_000] * 24, reads=20_000_000, L=100, e=3,
               name="C3: synthetic 100 bp reads, e=3, 24x125 Mbp random reference, k=12 step=3"),
    "c5": dict(seed=5, seq_lens=[125_000_000] * 24, reads=10_000_000, L=150, e=7,
               name="C5: synthetic 150 bp reads, e=7, 24x125 Mbp random reference, k=12 step=3"),
}


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: the workload's)")
    ap.add_argument("--cpu-sample", type=int, default=4_000_000, help="reads of the same workload timed on the host cores")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the hot path has no CPU fallback")
    # Rehearsal hooks for a one-GPU box (never set by the driver): FEM_BENCH_ONE_GPU=1 puts every rank on GPU 0,
    # FEM_BENCH_BACKEND=gloo reduces the counters over gloo instead of RCCL (RCCL refuses two ranks on one GPU).
    backend = os.environ.get("FEM_BENCH_BACKEND", "nccl")
    if os.environ.get("FEM_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    from fem_amd import Device, host

    w = WORKLOADS[args.workload]
    n_reads = args.reads or w["reads"]
    L, e, a, k, step = w["L"], w["e"], 1, 12, 3
    threads = max(1, min(len(os.sched_getaffinity(0)), 16))

    t0 = time.time()
    text, off, lens = host.synth_reference(w["seed"], w["seq_lens"], threads=threads)
    dev = Device(local_rank)
    dev.upload_reference([text[int(o):int(o) + int(l)] for o, l in zip(off, lens)])
    n_occ, _, _ = dev.build_index(k, step, fetch=False)
    log("rank %d: reference %d bp in %d sequences, index %d entries built on device in %.1fs"
        % (rank, int(lens.astype(np.uint64).sum()), len(lens), n_occ, time.time() - t0))
    t0 = time.time()
    bases, offsets = host.synth_reads(w["seed"], text, off, lens, n_reads, L, e, first_read=rank * n_reads, threads=threads)
    log("rank %d: %d reads generated in %.1fs" % (rank, n_reads, time.time() - t0))
    t0 = time.time()
    dev.stage_reads(bases, offsets, slot=0)
    h2d_s = time.time() - t0

    stats_dev = torch.zeros(5, dtype=torch.int64, device=red_dev)

    def step_once():
        dev.map_staged(e=e, a=a, k=k, step=step, slot=0)
        return dev.fetch_stats(slot=0)  # waits for the kernels (and re-runs the batch if a scratch buffer had to grow)

    def reduce_stats(st):
        # MappingStats reduction (reference src/FEM_map.c:200-212: once per job, after the last batch) = one 40-byte
        # RCCL all-reduce
        if world == 1:
            return st
        stats_dev.copy_(torch.from_numpy(st.astype(np.int64)))
        dist.all_reduce(stats_dev)
        return stats_dev.cpu().numpy().astype(np.uint64)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step_once()
    dev.set_timing(True)
    dev.reset_timing()
    fence()
    t_start = time.perf_counter()
    job_stats = np.zeros(5, dtype=np.uint64)
    for _ in range(args.steps):
        local_stats = step_once()
        job_stats += local_stats.astype(np.uint64)
    total_stats = reduce_stats(job_stats)  # (inside the timed region: it is the path's one exchange)
    fence()
    elapsed = time.perf_counter() - t_start
    dev.set_timing(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # per-kernel HIP-event totals over the timed region
    fast_ms, fast_n = dev.kernel_time(0)   # seed_fast_kernel<R, ...>
    ver_ms, ver_n = dev.kernel_time(1)     # verify_kernel
    gen_ms, gen_n = dev.kernel_time(2)     # seed_filter_kernel (generic form: queued reads)
    cnt_ms, cnt_n = dev.kernel_time(6)     # count_mappings_kernel (per-read counts + counters)
    ms_per_step = elapsed * 1e3 / args.steps
    value = world * n_reads * args.steps / elapsed / 1e6

    # algorithmic bytes: B = N*L + 16*(L-k+1)*N + 8*P + (L+2e)*C + 16*M, from the path's own counters
    N, P, Cn, M = n_reads, int(local_stats[2]), int(local_stats[3]), int(local_stats[4])
    S = L - k + 1
    seed_bytes = N * L + 16 * S * N + 8 * P      # read bases + one 8-byte lookup pair per seed and strand + occurrences
    verify_bytes = (L + 2 * e) * Cn + 16 * M     # reference window per verification + result record
    step_ms = {"seed_fast_kernel": fast_ms / args.steps, "seed_filter_kernel": gen_ms / args.steps,
               "verify_kernel": ver_ms / args.steps, "count_mappings_kernel": cnt_ms / args.steps}
    launches = {"seed_fast_kernel": fast_n / args.steps, "seed_filter_kernel": gen_n / args.steps,
                "verify_kernel": ver_n / args.steps, "count_mappings_kernel": cnt_n / args.steps}
    dominant = max(step_ms, key=step_ms.get)
    # the two seed kernels split the same reads: the dominant one is charged the seeding bytes of the whole batch
    dom_bytes_step = verify_bytes if dominant == "verify_kernel" else seed_bytes
    n_launch = max(launches[dominant], 1.0)
    dom_ms = step_ms[dominant] / n_launch          # mean duration of one launch
    dom_bytes = dom_bytes_step / n_launch          # algorithmic bytes of one launch
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    per_launch = {k_: (step_ms[k_] / max(launches[k_], 1.0)) for k_ in step_ms}

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # HBM-side traffic of the dominant kernel comes from separate rocprofv3 --pmc passes of this same command
    # (FETCH_SIZE, WRITE_SIZE; summaries committed under profiles/), scaled to this run's reads per launch.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_%s_hbm_traffic.json" % args.workload)
    if os.path.exists(tpath):
        try:
            prof = json.load(open(tpath))
            for kname, ctr in prof["kernels"].items():
                if dominant.split("_kernel")[0] in kname and "FETCH_SIZE" in ctr and ctr["FETCH_SIZE"]["mean_per_launch"] > 1e3:
                    kib = ctr["FETCH_SIZE"]["mean_per_launch"] + ctr.get("WRITE_SIZE", {}).get("mean_per_launch", 0.0)
                    traffic = int(kib * 1024 * (n_reads / n_launch) / prof["reads_per_launch"])
        except Exception as ex:  # a malformed profile file must not break the measurement
            log("could not read %s: %s" % (tpath, ex))

    out = {
        "metric": "mapped Mreads/s (100 bp, e=3) at 1/2/4/8 MI355X + achieved HBM GB/s vs roofline",
        "value": round(value, 3), "unit": "Mreads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64/u32 integer + 32-bit Myers bit-vectors", "data": "synthetic",
        "config": {"workload": w["name"], "reads_per_gpu": n_reads, "read_len": L, "e": e, "a": a, "k": k, "step": step,
                   "index_entries": n_occ, "parallelism": "reads sharded x%d, index replicated" % world,
                   # MappingStats of the whole job (all ranks, all timed steps), as the reference prints them at its end
                   "counters": {"reads": int(total_stats[0]), "mapped_reads": int(total_stats[1]),
                                "pre_filter": int(total_stats[2]), "candidates": int(total_stats[3]),
                                "mappings": int(total_stats[4])},
                   "counters_per_step_per_gpu": [int(x) for x in local_stats],
                   "algorithmic_bytes_per_step_per_gpu": seed_bytes + verify_bytes,
                   "kernel_ms_per_launch": {k_: round(v_, 4) for k_, v_ in per_launch.items()},
                   "kernel_launches_per_step": {k_: round(v_, 2) for k_, v_ in launches.items()},
                   "h2d_stage_s": round(h2d_s, 3)},
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(dom_ms, 4)},
    }

    if not args.no_cpu and world == 1:
        out["cpu_baseline"] = cpu_baseline(w, text, off, lens, min(args.cpu_sample, n_reads), bases, offsets, dev, e, a)
    print(json.dumps(out), flush=True)
    dev.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(w, text, off, lens, n_sample, bases, offsets, dev, e, a):
    """The oracle (CPU restatement of the reference, 'port') timed on this box's host cores on a bounded sample of
    the same workload, same stages as the device path (seeding + filter + verification).  Checker, never shipped."""
    from oracle import fem_oracle as fo
    cores = min(len(os.sched_getaffinity(0)), 16)  # the CPU share of a one-GPU box
    L = w["L"]
    ref = fo.Reference([text[int(o):int(o) + int(l)].tobytes() for o, l in zip(off, lens)])
    t0 = time.time()
    idx = fo.OracleIndex(ref)
    t_index = time.time() - t0
    sample = fo.ReadBatch.from_arrays(bases[:n_sample * L + 8], offsets[:n_sample + 1])
    t0 = time.perf_counter()
    h = fo.map_reads(ref, idx, sample, e=e, a=a, threads=cores, stages=fo.STAGE_SEED | fo.STAGE_VERIFY, keep_handle=True)
    dt = time.perf_counter() - t0
    st = np.zeros(5, np.uint64)
    fo.lib().fo_result_stats(h, st.ctypes.data)
    fo.free_result(h)
    # the same sample through the device path must give the same five counters
    got = dev.map_batch(bases[:n_sample * L + 8], offsets[:n_sample + 1], e=e, a=a, slot=1).stats
    return {"value": round(n_sample / dt / 1e6, 4), "unit": "Mreads/s", "cores": cores, "kind": "port",
            "sample": "%d reads of the same workload, seeding+filter+verification, %d threads" % (n_sample, cores),
            "seconds": round(dt, 3), "index_build_seconds": round(t_index, 2),
            "counters_match_device": bool(np.array_equal(got, st))}


if __name__ == "__main__":
    main()
