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


def gather_bits(local_bits, nchan, dst=0, always=False):
    """Optional epilogue: collect every rank's (channels_owned, n_bits) uint8 decisions on `dst` in
    channel order.  Ragged partitions are padded to the largest shard for the collective.
    always: go through the collective even in a group of one rank (bench.py --rccl-selftest)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not always):
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


def row_checksums(bits):
    """(rows, n) uint8 -> (rows,) int64 on the CPU: a position-weighted sum per row, so that rows that are swapped,
    shifted or padded differently do not collide.  The gather's check: every rank sends these over the control
    backend (gloo), the gathering rank recomputes them from the rows RCCL delivered."""
    n = bits.shape[1]
    w = (torch.arange(n, device=bits.device, dtype=torch.int64) % 65521) + 1
    return (bits.to(torch.int64) * w).sum(dim=1).cpu()


def gather_checksums(local_cs, nchan, device="cpu"):
    """Every rank's row checksums in channel order on every rank (returned on the CPU).  device = where the collective's
    tensors live: "cpu" = the control backend (gloo) when the group has one, a cuda device for an RCCL-only group."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_cs
    world = dist.get_world_size()
    sizes = [partition(nchan, r, world) for r in range(world)]
    most = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(most, dtype=torch.int64, device=device)
    pad[: local_cs.numel()] = local_cs.to(device)
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([bufs[r][: hi - lo].cpu() for r, (lo, hi) in enumerate(sizes)])


def pin_to_gpu_numa_node(device_index):
    """Keep this rank's host threads on the CPUs of its GPU's NUMA node, when sysfs says which that is
    (/sys/bus/pci/devices/<bdf>/numa_node, /sys/devices/system/node/node<N>/cpulist) and the process may run there.
    -> the node, or None when nothing was changed (unreadable, node -1, no CPU of the node in the affinity mask)."""
    import os
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        allowed = cpus & set(os.sched_getaffinity(0))
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
        return node
    except Exception:
        return None
