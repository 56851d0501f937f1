"""The N > 1 path on CPU: two gloo ranks exercise the channel partition, the max-over-ranks timing
reduction and the optional bit gather that bench.py / a multi-GPU caller use (no GPU needed)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sdr_amd import shard


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nchan, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.partition(nchan, rank, world)
    # every channel's "decisions" are a function of the channel id, so the gather is checkable
    local = torch.stack([torch.full((16,), c % 251, dtype=torch.uint8) for c in range(lo, hi)]) if hi > lo \
        else torch.zeros(0, 16, dtype=torch.uint8)
    t = shard.max_over_ranks(1.0 + rank)
    full = shard.gather_bits(local, nchan, dst=0)
    cs = shard.gather_checksums(shard.row_checksums(local), nchan)  # what bench.py's RCCL epilogue verifies against
    dist.barrier()
    if rank == 0:
        ok = bool(all(int(full[c, 0]) == c % 251 for c in range(nchan))) and cs.numel() == nchan \
            and torch.equal(shard.row_checksums(full), cs)
        q.put((t, full.shape[0], ok))
    else:
        assert full is None
    q.put((rank, lo, hi))
    dist.destroy_process_group()


def test_partition_covers_every_channel_once():
    for nchan in (1, 7, 1024, 1000):
        for world in (1, 2, 3, 8):
            parts = [shard.partition(nchan, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == nchan
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_row_checksums_tell_rows_apart():
    a = torch.zeros(3, 40, dtype=torch.uint8)
    a[0, 5] = 1
    a[1, 6] = 1
    a[2, 5] = 1
    cs = shard.row_checksums(a)
    assert cs[0] == cs[2] and cs[0] != cs[1] and cs.dtype == torch.int64
    assert shard.pin_to_gpu_numa_node(0) is None or True  # no GPU here: must not raise


@pytest.mark.parametrize("world,nchan", [(2, 1023), (2, 65), (3, 65), (4, 3)])
def test_two_rank_gloo_timing_and_gather(world, nchan):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, nchan, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = [q.get(timeout=120) for _ in range(world + 1)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    summary = [g for g in got if isinstance(g[0], float)][0]
    parts = sorted(g for g in got if not isinstance(g[0], float))
    assert summary == (float(world), nchan, True)  # max over ranks of (1.0 .. world); all channels, in order, checksums agree
    assert parts == [(r,) + shard.partition(nchan, r, world) for r in range(world)]  # (4 ranks, 3 channels: rank 0 owns none)
    if (world, nchan) == (2, 1023):
        assert parts == [(0, 0, 511), (1, 511, 1023)]
    if (world, nchan) == (2, 65):
        assert parts == [(0, 0, 32), (1, 32, 65)]
