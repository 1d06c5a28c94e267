"""bench.py end to end at toy sizes: the one-line JSON contract, and the N > 1 path (self-launching, one process per rank,
counters all-reduced) rehearsed as two gloo ranks on one GPU.  Needs a GPU: -m gpu."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-1000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    d = _run(["--workload", "c2", "--steps", "3", "--warmup", "2", "--batch", "200000", "--extra", "none", "--cpu-sample", "50000", "--e2e-reads", "100000",
              "--reps", "2"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "roofline_step", "spread", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "Mreads/s" and d["vs_baseline"] is None and d["value"] > 0
    assert abs(d["value"] - 200000 * 3 / (d["ms_per_step"] * 3e-3) / 1e6) < 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    # value, ms_per_step and roofline come from one workload: the dominant kernel's launches fit into the step
    assert r["workload"] == "c2" and r["avg_launch_ms"] <= d["ms_per_step"]
    # `value` is the median of the timed repetitions (of two: the later one in order of time), and says which one it is
    sp = d["spread"]
    assert sp["reps"] == 2 and sp["values"][sp["value_is_repetition"]] == d["value"] and sp["min"] <= d["value"] <= sp["max"]
    assert d["value"] == sorted(sp["values"], reverse=True)[len(sp["values"]) // 2]
    assert d["roofline_step"]["avg_launch_ms"] == pytest.approx(d["ms_per_step"], rel=1e-3)
    assert d["config"]["counters"]["reads"] == 3 * 200000 and d["config"]["kernel_only_mreads"] > 0
    assert d["cpu_baseline"]["counters_match_device"] is True and d["cpu_baseline"]["kind"] == "port"
    assert d["e2e_cli"].get("value", 0) > 0 and d["e2e_cli"]["to_dev_null"]["value"] > 0, d["e2e_cli"]
    c = d["config"]
    # the timed steps + the W warm-up steps + the disclosed priming steps are all the batches the workload saw before / in them
    assert c["priming_steps"] == 100 - 2 and c["h2d_packed"] is True and c["h2d_bytes_per_step"] == 200000 * 25
    assert c["zero_copy_ascii_mreads"] > 0 and c["zero_copy_h2d_bytes_per_step"] == 200000 * 100 and c["stage_reads_mreads"] > 0
    # the pipeline checked itself: every repeat of a slot's batch gave the same counters, and slot 0's result, taken out of the
    # running four-deep pipeline, equals the oracle's on its prefix
    pc = c["pipeline_check"]
    assert d["counters_match_pipeline"] is True and pc["repeats_identical"] is True and min(pc["batches_retired_per_slot"]) >= 5
    assert pc["prefix_vs_oracle"]["equal_to_oracle"] is True and pc["prefix_vs_oracle"]["reads"] == 100000 and pc["prefix_vs_oracle"]["mappings"] > 50000
    assert c["host_ms_per_step"]["stage_call"] >= 0 and "stage_reads_form" in c["host_ms_per_step"]


def test_two_ranks_start_by_themselves_and_reduce_their_counters():
    one = _run(["--workload", "c2", "--reps", "1", "--steps", "2", "--warmup", "1", "--batch", "100000", "--extra", "none", "--no-cpu", "--no-e2e"])
    two = _run(["--workload", "c2", "--reps", "1", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "100000", "--extra", "none", "--no-cpu", "--no-e2e"],
               env={"FEM_BENCH_ONE_GPU": "1", "FEM_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert two["counters_match_pipeline"] is True and one["counters_match_pipeline"] is True
    assert two["config"]["counters"]["reads"] == 2 * one["config"]["counters"]["reads"]
    # rank 0 maps the same read indices as the single-rank run; rank 1 different reads of the same distribution
    assert two["config"]["counters_last_step_per_gpu"] == one["config"]["counters_last_step_per_gpu"]
    assert 1.8 < two["config"]["counters"]["mapped_reads"] / one["config"]["counters"]["mapped_reads"] < 2.2


def test_one_rank_reduces_its_counters_over_rccl():
    # the RCCL all-reduce of the counters (src/FEM_map.c:200-212; bench.py's N > 1 branch) on a one-GPU box: a one-rank
    # "nccl" process group, the same calls as with eight ranks
    plain = _run(["--workload", "c2", "--reps", "1", "--steps", "2", "--warmup", "1", "--batch", "100000", "--extra", "none", "--no-cpu", "--no-e2e"])
    rccl = _run(["--workload", "c2", "--reps", "1", "--steps", "2", "--warmup", "1", "--batch", "100000", "--extra", "none", "--no-cpu", "--no-e2e"],
                env={"FEM_BENCH_FORCE_DIST": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                     "MASTER_PORT": "29533", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert rccl["n_gpus"] == 1 and "nccl" in rccl["config"]["counter_reduction"] and plain["config"]["counter_reduction"].startswith("single")
    assert rccl["config"]["counters"] == plain["config"]["counters"]


def test_headline_workload_is_c3_with_two_seed_kernels():
    # the default line at a toy batch: C3's 3 Gbp reference, seed_select_kernel + seed_join_kernel, one workload behind
    # value / ms_per_step / roofline
    d = _run(["--steps", "3", "--warmup", "2", "--batch", "100000", "--extra", "none", "--no-cpu", "--no-e2e", "--reps", "1"])
    assert d["roofline"]["workload"] == "c3" and d["roofline"]["kernel"] == "seed_join_kernel" and d["config"]["seed_kernel"] == "seed_join_kernel"
    assert set(d["roofline_by_kernel"]) >= {"seed_join_kernel", "seed_select_kernel"}
    assert d["roofline"]["avg_launch_ms"] <= d["ms_per_step"]
    c = d["config"]["counters_last_step_per_gpu"]
    assert d["counters_match_pipeline"] is True
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 8 * c[2]
    assert d["roofline"]["implementation_bytes_per_launch"] == 4 * c[2] + (8 * 6 * 5 + 8) * 100000
    assert c[1] > 0.97 * c[0]  # 98.4 % of the reads map on the 3 Gbp reference
