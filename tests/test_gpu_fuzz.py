"""A bounded slice of tests/fuzz_gpu.py under `pytest -m gpu`: fixed seeds, fixed trial counts (580 batches, about a minute), so
that the driver's GPU run exercises the randomised differential comparison too — device path against the oracle on
repeat-rich, sparse, mid-density and dense (216 Mbp: seed_select_kernel + seed_join_kernel) references, e in 0..7, a in {1, 2}, read lengths 30..300 (equal or mixed), damaged
reads, lower case, N rates; candidates, edit distances, end offsets, counters, the tail's records and the device's SAM
text (reference path: src/map.c:27-55, src/filter.c:146-223, src/align.c:4-147,279-544)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,kinds,trials", [(101, ("repeat",), 200), (202, ("sparse",), 200), (303, ("mid",), 120), (404, ("dense",), 60)])
def test_fuzz_slice(seed, kinds, trials):
    from tests import fuzz_gpu
    lines = []
    n, bad = fuzz_gpu.run(seed, kinds=kinds, trials_per_kind=trials, threads=8, max_reads=6000, log=lambda *a, **k: lines.append(" ".join(str(x) for x in a)))
    assert n == trials and bad == 0, "\n".join(lines)
