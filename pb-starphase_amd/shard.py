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


class TorchGroup:
    """the same gather as ffi.Group.gather (sp_gather_results) through torch.distributed: `gloo` in the CPU tests and on a one-GPU box
    (RCCL refuses two ranks on one device), `nccl` where the host prefers torch's communicator"""

    def __init__(self, device=None):
        import torch.distributed as dist
        self.n_ranks, self.rank = dist.get_world_size(), dist.get_rank()
        self.device = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")

    def gather(self, records):
        import torch
        import torch.distributed as dist
        rec = np.ascontiguousarray(records)
        mine = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(self.device)
        parts = [torch.zeros_like(mine) for _ in range(self.n_ranks)]
        dist.all_gather(parts, mine)
        out = np.stack([p.cpu().numpy() for p in parts]).view(rec.dtype)
        return out.reshape((self.n_ranks,) + rec.shape)


class SoloGroup:
    """one rank on its own inside a larger job (nothing exchanged): what a rank passes to keep a call off the process group"""
    n_ranks, rank = 1, 0

    def gather(self, records):
        return np.ascontiguousarray(records)[None]


def make_group(ctx, ffi, backend="nccl", device=None):
    """The group a rank gathers through.  `nccl`: an sp_group of the library (librccl, ncclAllGather on the context's stream); its 128-byte
    id is made by rank 0 and handed out through the KEY-VALUE STORE of the launcher's rendezvous -- not through a collective of the process group: a rank
    that never gets here then leaves nothing pending on that group (its peers wait in store.get, which times out).  Anything else: the torch group.
    Give it a context of its own (pkg.Context(device)): should the first gather never complete, it sits on that context's stream and on no stream the run needs."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return None
    if backend != "nccl":
        return TorchGroup(device)
    rank, world = dist.get_rank(), dist.get_world_size()
    store = dist.distributed_c10d._get_default_store()
    if rank == 0:
        store.set("sp_group_unique_id", bytes(np.ascontiguousarray(ffi.group_unique_id(), np.uint8)))
    uid = np.frombuffer(store.get("sp_group_unique_id"), np.uint8).copy()
    return ffi.Group(ctx, uid, rank, world)


def gather_calls(calls, device=None, same_count=False, group=None):
    """All ranks contribute their call records; every rank receives the full, sample/gene-sorted table.
    Records are fixed size: one gather of the counts and one of the padded records; with same_count (every rank holds the same number of
    records, e.g. its share of a cohort) the counts are not exchanged: one collective, no round trip for the sizes.
    group: ffi.Group (sp_gather_results: RCCL) or TorchGroup; None = the torch process group when there is one, else a single rank."""
    calls = np.ascontiguousarray(calls, CALL_DTYPE)
    if group is None:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return np.sort(calls, order=["sample", "gene"])
        group = TorchGroup(device)
    world = group.n_ranks
    counts = [len(calls)] * world if same_count else [int(c) for c in group.gather(np.array([len(calls)], np.int64)).reshape(-1)]
    nmax = max(counts) if counts else 0
    pad = np.zeros(nmax, CALL_DTYPE)
    pad[:len(calls)] = calls
    everyone = group.gather(pad)
    out = np.concatenate([everyone[r][:counts[r]] for r in range(world)]) if world else pad
    return np.sort(out, order=["sample", "gene"])
