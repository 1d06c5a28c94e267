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
    d = _run(["--steps", "3", "--warmup", "2", "--batch", "200000", "--extra", "none", "--cpu-sample", "50000", "--e2e-reads", "100000"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "Mreads/s" and d["vs_baseline"] is None and d["value"] > 0
    assert abs(d["value"] - 200000 * 3 / (d["ms_per_step"] * 3e-3) / 1e6) < 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert d["config"]["counters"]["reads"] == 3 * 200000 and d["config"]["kernel_only_mreads"] > 0
    assert d["cpu_baseline"]["counters_match_device"] is True and d["cpu_baseline"]["kind"] == "port"
    assert d["e2e_cli"].get("value", 0) > 0 and d["e2e_cli"]["to_dev_null"]["value"] > 0, d["e2e_cli"]
    c = d["config"]
    # the timed steps + the W warm-up steps + the disclosed priming steps are all the batches the workload saw before / in them
    assert c["priming_steps"] == 12 - 2 and c["h2d_packed"] is True and c["h2d_bytes_per_step"] == 200000 * 25
    assert c["zero_copy_ascii_mreads"] > 0 and c["zero_copy_h2d_bytes_per_step"] == 200000 * 100


def test_two_ranks_start_by_themselves_and_reduce_their_counters():
    one = _run(["--steps", "2", "--warmup", "1", "--batch", "100000", "--extra", "none", "--no-cpu", "--no-e2e"])
    two = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "100000", "--extra", "none", "--no-cpu", "--no-e2e"],
               env={"FEM_BENCH_ONE_GPU": "1", "FEM_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert two["config"]["counters"]["reads"] == 2 * one["config"]["counters"]["reads"]
    # rank 0 maps the same read indices as the single-rank run; rank 1 different reads of the same distribution
    assert two["config"]["counters_last_step_per_gpu"] == one["config"]["counters_last_step_per_gpu"]
    assert 1.8 < two["config"]["counters"]["mapped_reads"] / one["config"]["counters"]["mapped_reads"] < 2.2
