"""Channel sharding across GPUs: channels/streams are independent (private FIR
history and PSK loop state, no cross-channel term), so the data path needs NO collective.
One process per GPU; torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only for the
timing barrier, the max-over-ranks, and the optional final gather of decision bits."""
import torch
import torch.distributed as dist


def partition(nchan, rank, world):
    """Static block partition: rank g of G owns channels [g*C/G, (g+1)*C/G) (balanced to +-1)."""
    lo = (nchan * rank) // world
    hi = (nchan * (rank + 1)) // world
    return lo, hi


def max_over_ranks(seconds, device="cpu"):
    """The job's step time is the slowest rank's."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_bits(local_bits, nchan, dst=0):
    """Optional epilogue: collect every rank's (channels_owned, n_bits) uint8 decisions on `dst` in
    channel order.  Ragged partitions are padded to the largest shard for the collective."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_bits
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [partition(nchan, r, world) for r in range(world)]
    most = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(most, local_bits.shape[1], dtype=local_bits.dtype, device=local_bits.device)
    pad[: local_bits.shape[0]] = local_bits
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    if rank != dst:
        return None
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)
