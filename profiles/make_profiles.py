"""Regenerates the rocprofv3 summaries under profiles/ (run on the GPU box, from the repo root):

    python3 profiles/make_profiles.py calib       # -> profiles/r05_fetch_calibration.json
    python3 profiles/make_profiles.py c3          # -> profiles/r05_c3_kernel_stats.csv, r05_c3_kernel_times.json, r05_c3_hbm_traffic.json

Every file carries `git_head` and `kernel_sources_sha16` (bench.kernel_sources_sha16: a hash of fem_amd/csrc/*.hip*): bench.py
takes `roofline.traffic` from a profile only if the hash is that of the sources it runs on.

The command profiled for a workload is the RESIDENT REPLAY of bench.py (`--profile-replay 20`): the four slots' batches
are staged once, then nothing but kernels run (no copies in flight: under rocprofv3 every copy is a
`__amd_rocclr_copyBuffer` kernel that takes CUs from the seed kernels), with the selection of one batch beside the join
of the previous one exactly as in the bench line's pipeline.  bench.py prints the HIP-event means of exactly those
launches; `r03_<wl>_kernel_times.json` sets them beside rocprofv3's own kernel-trace durations of the same launches
(warm-up launches excluded by their position in the trace) and beside the `--stats` averages (warm-ups included).
One `--kernel-trace --stats` run, then one `--pmc` run per counter group (PMC runs never carry trace options).
This script itself never touches the GPU; rocprofv3 gets the program directly after `--`.
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("FEM_PROFILE_ROUND", "r05")
PMC_GROUPS = [  # the derived TCC counters each fill the hardware's counter slots: one per pass
    ["FETCH_SIZE"],
    ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"],
    ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"],
    ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"],
]
KERNELS = ("seed_select_kernel", "seed_join_kernel", "seed_fast_kernel", "seed_filter_kernel", "verify_kernel",
           "calib_runs", "calib_gather", "calib_pairs")


def stamp():
    sys.path.insert(0, ROOT)
    import bench
    try:
        head = subprocess.check_output(["git", "rev-parse", "HEAD"], cwd=ROOT, stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        head = os.environ.get("FEM_GIT_HEAD", "unknown (the GPU box gets a snapshot without .git; FEM_GIT_HEAD names it)")
    return {"git_head": head, "kernel_sources_sha16": bench.kernel_sources_sha16()}


def run(cmd, log, timeout=600):
    print("running:", " ".join(cmd), flush=True)
    with open(log, "w") as f:
        try:
            rc = subprocess.call(cmd, stdout=f, stderr=subprocess.STDOUT, cwd=ROOT, timeout=timeout)
        except subprocess.TimeoutExpired:
            raise SystemExit("timed out: %s  -- see %s" % (" ".join(cmd), log))
    if rc != 0:
        raise SystemExit("failed (%d): %s  -- see %s" % (rc, " ".join(cmd), log))


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    if not files:
        raise SystemExit("no file matches " + pattern)
    return max(files, key=os.path.getmtime)


def short_name(name):
    return next((k for k in KERNELS if k in name), None)


def counters(path):
    """{kernel: {counter: [value per launch, in launch order]}}"""
    acc = {}
    for row in csv.DictReader(open(path)):
        k = short_name(row.get("Kernel_Name", ""))
        if k:
            acc.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return acc


def calibrate():
    out = os.path.join(ROOT, "gpurun_out", "profiles_calib")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    exe = os.path.join(ROOT, "gpurun_out", "fetch_calibration")
    run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", os.path.join(ROOT, "profiles", "fetch_calibration.hip"), "-o", exe],
        os.path.join(out, "build.log"))
    got = {}
    for gi, group in enumerate([["FETCH_SIZE"], ["TCC_MISS_sum", "TCC_HIT_sum"], ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"]]):
        d = os.path.join(out, "pmc%d" % gi)
        try:
            run(["rocprofv3", "--pmc"] + group + ["--output-format", "csv", "-d", d, "--", exe], os.path.join(out, "pmc%d.log" % gi))
        except SystemExit as ex:  # (a counter this build of rocprofv3 does not know: the others still count)
            print("skipped", group, ex)
            continue
        for k, cs in counters(newest(os.path.join(d, "**", "*counter_collection.csv"))).items():
            for c, v in cs.items():
                got.setdefault(k, {})[c] = v[-1]  # the second launch (caches warm for the tables)
    known = json.loads("".join(l for l in open(os.path.join(out, "pmc0.log")) if l.lstrip().startswith(("{", '"'))))
    res = dict(stamp(), what="FETCH_SIZE of rocprofv3 (KiB) against known byte counts, profiles/fetch_calibration.hip; second launch of each kernel",
               known=known, measured=got, ratios={})
    if "calib_runs" in got and "FETCH_SIZE" in got["calib_runs"]:
        f = got["calib_runs"]["FETCH_SIZE"] * 1024.0
        r = known["runs"]
        res["ratios"]["runs"] = {"fetch_over_asked": f / r["asked_bytes"], "fetch_over_sector64": f / r["sector64_bytes"],
                                 "fetch_over_line128": f / r["line128_bytes"],
                                 "misses_per_run": got["calib_runs"].get("TCC_MISS_sum", 0.0) / r["n"]}
    for k, kn in (("calib_gather", "gather"), ("calib_pairs", "pairs")):
        if k in got and "FETCH_SIZE" in got[k]:
            f = got[k]["FETCH_SIZE"] * 1024.0
            miss = got[k].get("TCC_MISS_sum", 0.0)
            res["ratios"][kn] = {"fetch_over_asked": f / known[kn]["asked_bytes"], "fetch_bytes_per_l2_miss": f / miss if miss else None,
                                 "l2_misses_per_lane_load": miss / known[kn]["lane_loads"] if miss else None}
    with open(os.path.join(ROOT, "profiles", "%s_fetch_calibration.json" % ROUND), "w") as fo:
        json.dump(res, fo, indent=1, sort_keys=True)
    print(json.dumps(res["ratios"], indent=1))


def workload(wl):
    out = os.path.join(ROOT, "gpurun_out", "profiles_" + wl)
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    bench = [sys.executable, "bench.py", "--workload", wl, "--profile-replay", "20"]
    note = "rocprofv3 ... -- python3 bench.py --workload %s --profile-replay 20" % wl

    # 1. durations
    d = os.path.join(out, "trace")
    run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] + bench, os.path.join(out, "trace.log"))
    shutil.copy(newest(os.path.join(d, "**", "*kernel_stats.csv")), os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (ROUND, wl)))
    line = [l for l in open(os.path.join(out, "trace.log")) if l.startswith("{")]
    bench_out = json.loads(line[-1]) if line else {}
    warm = int(bench_out.get("warmup_launches_per_kernel", 0))
    rows = sorted(csv.DictReader(open(newest(os.path.join(d, "**", "*kernel_trace.csv")))), key=lambda r: int(r["Start_Timestamp"]))
    per = {}
    for r in rows:
        k = short_name(r["Kernel_Name"])
        if k:
            per.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    times = dict(stamp(), command=note, bench=bench_out, kernels={})
    for k, v in per.items():
        timed = v[warm:] if len(v) > warm else v
        ev = bench_out.get("event_times", {}).get(k, {})
        times["kernels"][k] = {"trace_launches": len(v), "trace_mean_ms_all": sum(v) / len(v), "timed_launches": len(timed),
                               "trace_mean_ms_timed": sum(timed) / len(timed), "trace_min_ms_timed": min(timed), "trace_max_ms_timed": max(timed),
                               "event_mean_ms": ev.get("mean_ms"), "event_launches": ev.get("launches"),
                               "trace_over_event": (sum(timed) / len(timed)) / ev["mean_ms"] if ev.get("mean_ms") else None}
    with open(os.path.join(ROOT, "profiles", "%s_%s_kernel_times.json" % (ROUND, wl)), "w") as f:
        json.dump(times, f, indent=1, sort_keys=True)

    # 2. counters, one pass per group
    kernels = {}
    for gi, group in enumerate(PMC_GROUPS):
        d = os.path.join(out, "pmc%d" % gi)
        run(["rocprofv3", "--pmc"] + group + ["--output-format", "csv", "-d", d, "--"] + bench[:-1] + ["6"], os.path.join(out, "pmc%d.log" % gi))
        for k, cs in counters(newest(os.path.join(d, "**", "*counter_collection.csv"))).items():
            for c, v in cs.items():
                timed = v[warm:] if len(v) > warm else v
                kernels.setdefault(k, {})[c] = {"launches": len(timed), "mean_per_launch": sum(timed) / len(timed)}
    # HBM-side bytes per launch: FETCH_SIZE corrected by the calibration of this access pattern, + WRITE_SIZE
    cal_path = os.path.join(ROOT, "profiles", "%s_fetch_calibration.json" % ROUND)
    cal = json.load(open(cal_path))["ratios"] if os.path.exists(cal_path) else {}
    for k, cs in kernels.items():
        if "FETCH_SIZE" not in cs:
            continue
        factor, why = 1.0, "uncalibrated"
        if k == "seed_join_kernel" and "runs" in cal:  # nearly all of its fetches are list reads
            factor, why = 1.0 / cal["runs"]["fetch_over_line128"], "calib_runs: the 128-byte lines its runs touch / FETCH_SIZE"
        elif k == "seed_select_kernel" and "gather" in cal and cal["gather"].get("fetch_bytes_per_l2_miss"):
            factor, why = 64.0 / cal["gather"]["fetch_bytes_per_l2_miss"], "calib_gather: 64 bytes per L2 miss / FETCH_SIZE per miss"
        cs["fetch_factor"] = factor
        cs["fetch_factor_from"] = why
        cs["hbm_bytes_per_launch"] = cs["FETCH_SIZE"]["mean_per_launch"] * 1024.0 * factor + cs.get("WRITE_SIZE", {}).get("mean_per_launch", 0.0) * 1024.0
    summary = dict(stamp())
    summary.update({
        "command": note + "  (one rocprofv3 --pmc pass per counter group with --profile-replay 6; the durations come from a separate "
                   "--kernel-trace --stats pass, profiles/%s_%s_kernel_stats.csv and _kernel_times.json)" % (ROUND, wl),
        "units": "FETCH_SIZE / WRITE_SIZE in KiB per launch as rocprofv3 reports them; hbm_bytes_per_launch = FETCH_SIZE x fetch_factor "
                 "(profiles/%s_fetch_calibration.json) + WRITE_SIZE, in bytes; Infinity-Cache hits are counted by FETCH_SIZE; SQ_*_CYCLES "
                 "summed over the chip's shader engines (quad-cycles)" % ROUND,
        "reads_per_launch": int(bench_out.get("reads_per_launch", 0)),
        "counter_groups": PMC_GROUPS,
        "kernels": kernels,
    })
    with open(os.path.join(ROOT, "profiles", "%s_%s_hbm_traffic.json" % (ROUND, wl)), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    print(json.dumps({k: {"trace_ms": round(v["trace_mean_ms_timed"], 3), "event_ms": v["event_mean_ms"]} for k, v in times["kernels"].items()}))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "c3"
    if what == "calib":
        calibrate()
    else:
        workload(what)
