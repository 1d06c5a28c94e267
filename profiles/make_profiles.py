"""Regenerates the rocprofv3 summaries under profiles/ for one bench workload (run on the GPU box, from the repo root):

    python3 profiles/make_profiles.py c2        # -> profiles/r02_c2_kernel_stats.csv, profiles/r02_c2_hbm_traffic.json

One `--kernel-trace --stats` run for the per-kernel durations, then one `--pmc` run per counter group (PMC runs never
carry trace options).  Every run profiles the same command: python3 bench.py --no-cpu --no-e2e --extra none --workload <wl>.
This script itself never touches the GPU; rocprofv3 gets the python interpreter directly after `--`.
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC_GROUPS = [  # the derived TCC counters each fill the hardware's counter slots: one per pass
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["TCC_HIT_sum"],
    ["TCC_MISS_sum"],
    ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"],
    ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY"],
]
KERNELS = ("seed_dense_kernel", "seed_fast_kernel", "seed_filter_kernel", "verify_kernel", "count_mappings_kernel",
           "bucket_summary_kernel", "trace_fast_kernel", "trace_kernel", "gather_kernel", "sort_kernel", "compact_kernel")
ROUND = os.environ.get("FEM_PROFILE_ROUND", "r02")


def run(cmd, log):
    print("running:", " ".join(cmd), flush=True)
    with open(log, "w") as f:
        try:
            rc = subprocess.call(cmd, stdout=f, stderr=subprocess.STDOUT, cwd=ROOT, timeout=300)
        except subprocess.TimeoutExpired:
            raise SystemExit("timed out: %s  -- see %s" % (" ".join(cmd), log))
    if rc != 0:
        raise SystemExit("failed (%d): %s  -- see %s" % (rc, " ".join(cmd), log))


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    if not files:
        raise SystemExit("no file matches " + pattern)
    return max(files, key=os.path.getmtime)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
    out = os.path.join(ROOT, "gpurun_out", "profiles_" + wl)
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    bench = [sys.executable, "bench.py", "--no-cpu", "--no-e2e", "--extra", "none", "--workload", wl]
    env_note = "rocprofv3 ... -- python3 bench.py --no-cpu --no-e2e --extra none --workload " + wl

    # 1. durations
    d = os.path.join(out, "trace")
    run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] + bench, os.path.join(out, "trace.log"))
    shutil.copy(newest(os.path.join(d, "**", "*kernel_stats.csv")), os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (ROUND, wl)))
    bench_line = [l for l in open(os.path.join(out, "trace.log")) if l.startswith("{")]
    reads_per_launch = None
    if bench_line:
        b = json.loads(bench_line[-1])
        lps = b["config"]["kernel_launches_per_step"]
        launches = max(lps.get("seed_dense_kernel", lps.get("seed_fast_kernel", 1.0)), 1.0)
        reads_per_launch = int(b["config"]["reads_per_step_per_gpu"] / launches)

    # 2. counters, one pass per group
    kernels = {}
    for gi, group in enumerate(PMC_GROUPS):
        d = os.path.join(out, "pmc%d" % gi)
        run(["rocprofv3", "--pmc"] + group + ["--output-format", "csv", "-d", d, "--"] + bench + ["--steps", "2", "--warmup", "1"],
            os.path.join(out, "pmc%d.log" % gi))
        path = newest(os.path.join(d, "**", "*counter_collection.csv"))
        acc = {}
        for row in csv.DictReader(open(path)):
            name = row.get("Kernel_Name", "")
            short = next((k for k in KERNELS if k in name), None)
            if not short:
                continue
            key = (name.split("(")[0].replace("void ", "").strip(), row["Counter_Name"])
            v = float(row["Counter_Value"])
            n, s = acc.get(key, (0, 0.0))
            acc[key] = (n + 1, s + v)
        for (kname, cname), (n, s) in acc.items():
            kernels.setdefault(kname, {})[cname] = {"launches": n, "mean_per_launch": s / n}

    summary = {
        "command": env_note + "  (one rocprofv3 --pmc pass per counter group; the durations come from a separate "
                   "--kernel-trace --stats pass, profiles/%s_%s_kernel_stats.csv)" % (ROUND, wl),
        "units": "FETCH_SIZE / WRITE_SIZE in KiB per launch as reported by rocprofv3 (gfx950 caveat: FETCH_SIZE "
                 "under-reports wide coalesced reads by up to 2x, MI355X_MICROARCH.md section HBM; Infinity-Cache hits are "
                 "counted); SQ_*_CYCLES summed over the chip's shader engines",
        "reads_per_launch": reads_per_launch,
        "counter_groups": PMC_GROUPS,
        "kernels": kernels,
    }
    with open(os.path.join(ROOT, "profiles", "%s_%s_hbm_traffic.json" % (ROUND, wl)), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    print("wrote profiles/%s_%s_kernel_stats.csv and profiles/%s_%s_hbm_traffic.json" % (ROUND, wl, ROUND, wl))


if __name__ == "__main__":
    main()
