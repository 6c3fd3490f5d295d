"""Sharding of independent units (samples / genes / reads) over ranks and the one exchange step of the path:
a gather of fixed-size per-(sample, gene) call records.  Backend agnostic: `nccl` (= RCCL over xGMI)
on the GPU node, `gloo` in the CPU tests.  No data-path collective exists anywhere else."""
import numpy as np

CALL_DTYPE = np.dtype([("sample", np.int32), ("gene", np.int32), ("allele1", np.int32), ("allele2", np.int32)])


def partition(n_units, world, rank, cost=None):
    """Units owned by `rank`.  Without costs: contiguous blocks whose sizes differ by at most one.
    With costs (e.g. reads x alleles per sample): greedy longest-processing-time assignment, deterministic."""
    if cost is None:
        base, extra = divmod(n_units, world)
        start = rank * base + min(rank, extra)
        return list(range(start, start + base + (1 if rank < extra else 0)))
    order = sorted(range(n_units), key=lambda u: (-cost[u], u))
    load = [0] * world
    owner = [0] * n_units
    for u in order:
        r = min(range(world), key=lambda x: (load[x], x))
        owner[u] = r
        load[r] += cost[u]
    return [u for u in range(n_units) if owner[u] == rank]


def gather_calls(calls, device=None, same_count=False):
    """All ranks contribute their call records; every rank receives the full, sample/gene-sorted table.
    Records are fixed size, so one all_gather of the counts and one of a padded int32 tensor suffice; with same_count (every
    rank holds the same number of records, e.g. one sample with G genes each) the counts are not exchanged: one collective,
    no host round trip for the sizes."""
    import torch
    import torch.distributed as dist
    calls = np.ascontiguousarray(calls, CALL_DTYPE)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return np.sort(calls, order=["sample", "gene"])
    world = dist.get_world_size()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    if same_count:
        counts = [torch.tensor([len(calls)], dtype=torch.int64) for _ in range(world)]
    else:
        n = torch.tensor([len(calls)], dtype=torch.int64, device=dev)
        counts = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(counts, n)
    nmax = int(max(int(c.item()) for c in counts))
    flat = np.zeros((nmax, 4), np.int32)
    if len(calls):
        flat[:len(calls)] = calls.view(np.int32).reshape(-1, 4)
    mine = torch.from_numpy(flat).to(dev)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    rows = [p.cpu().numpy()[:int(c.item())] for p, c in zip(parts, counts)]
    out = np.concatenate(rows, axis=0).astype(np.int32) if rows else np.zeros((0, 4), np.int32)
    out = np.ascontiguousarray(out).view(CALL_DTYPE).reshape(-1)
    return np.sort(out, order=["sample", "gene"])
