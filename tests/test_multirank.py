"""The N > 1 path on CPU: two gloo ranks each map their shard (with the oracle standing in for the device) and
all-reduce the MappingStats counters; the result must equal the single-process run and the shards must tile the batch."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from fem_amd.shard import shard_range


def test_shards_tile_the_batch():
    for n in (0, 1, 7, 1000, 10_000_019):
        for world in (1, 2, 3, 8):
            edges = [shard_range(r, world, n) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, seed, out_dir):
    import torch.distributed as dist
    from fem_amd.shard import reduce_stats, shard_range
    from oracle import fem_oracle as fo
    from tests import util
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(seed)  # same data on every rank (index + reference replicated)
    seqs = [util.rand_seq(rng, 120_000), util.rand_seq(rng, 60_000)]
    reads = util.make_reads(rng, seqs, 301, 100, 3)
    ref = fo.Reference(seqs)
    idx = fo.OracleIndex(ref)
    lo, hi = shard_range(rank, world, len(reads))
    part = fo.map_reads(ref, idx, fo.ReadBatch(reads[lo:hi]), e=3, stages=fo.STAGE_SEED | fo.STAGE_VERIFY)
    total = reduce_stats(part.stats)
    np.save(os.path.join(out_dir, "stats_%d.npy" % rank), total)
    np.save(os.path.join(out_dir, "cands_%d.npy" % rank), part.cands)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reduce_to_the_single_process_counters(tmp_path):
    from oracle import fem_oracle as fo
    from tests import util
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    seed, world = 1234, 2
    mp.spawn(_worker, args=(world, port, seed, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(seed)
    seqs = [util.rand_seq(rng, 120_000), util.rand_seq(rng, 60_000)]
    reads = util.make_reads(rng, seqs, 301, 100, 3)
    ref = fo.Reference(seqs)
    whole = fo.map_reads(ref, fo.OracleIndex(ref), fo.ReadBatch(reads), e=3, stages=fo.STAGE_SEED | fo.STAGE_VERIFY)
    for r in range(world):
        assert np.array_equal(np.load(str(tmp_path / ("stats_%d.npy" % r))), whole.stats)
    cands = np.concatenate([np.load(str(tmp_path / ("cands_%d.npy" % r))) for r in range(world)])
    assert np.array_equal(cands, whole.cands)  # shards in rank order == the unsharded batch
