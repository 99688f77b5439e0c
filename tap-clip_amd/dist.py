"""Data-parallel exchange step: the image batch is sharded over the GPUs of one node, every rank
encodes its shard, and the L2-normalised embeddings [B_local, E] are all-gathered before the
cosine-logit matrix (BASELINE.json north_star; nothing of this exists in the reference, which is
single-process: SURVEY.md section 8e).  One process per GPU; backend "nccl" is RCCL over xGMI on
ROCm, "gloo" on CPU for the tests.  Payload: 256 x 512 fp32 = 512 KiB per rank -- latency-bound."""
import torch
import torch.distributed as dist


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _row_counts(n_local: int, device, group=None):
    """every rank's row count, as a python list (one tiny collective)"""
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo":
        parts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(parts, torch.tensor([n_local], dtype=torch.int64), group=group)
        return [int(p) for p in parts]
    out = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, torch.tensor([n_local], dtype=torch.int64, device=device), group=group)
    return [int(v) for v in out.tolist()]


def all_gather_rows(x: torch.Tensor, group=None, force: bool = False, ragged: bool = False) -> torch.Tensor:
    """[B_local, E] on every rank -> [sum of B_local, E], rank-major rows.  Identity at world size 1 (`force=True`
    still issues the collective on an initialised group of one rank: the RCCL call path on a 1-GPU box).

    `ragged=False` (the benchmarked hot path: the global batch is sharded evenly, `shard_rows`) gathers straight into
    one tensor and REQUIRES the same row count on every rank -- unequal counts would hang RCCL.  `ragged=True` first
    exchanges the row counts (one 8-byte collective), pads every shard to the longest and drops the padding after the
    gather: the last, short batch of an evaluation loader that is sharded without padding (drop_last=False)."""
    if not (is_distributed() or (force and dist.is_available() and dist.is_initialized())):
        return x
    x = x.contiguous()
    world = dist.get_world_size(group)
    if ragged:
        counts = _row_counts(x.shape[0], x.device, group)
        longest = max(counts)
        if any(c != longest for c in counts):
            pad = torch.zeros((longest - x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            full = all_gather_rows(torch.cat([x, pad], dim=0), group=group, force=force, ragged=False)
            keep = torch.cat([torch.arange(r * longest, r * longest + c) for r, c in enumerate(counts)]).to(full.device)
            return full.index_select(0, keep)
    if dist.get_backend(group) == "gloo":
        # CPU rehearsal backend (tests; several ranks sharing one GPU): stage through host memory
        parts = [torch.empty(x.shape, dtype=x.dtype) for _ in range(world)]
        dist.all_gather(parts, x.detach().cpu(), group=group)
        return torch.cat(parts, dim=0).to(x.device)
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x, group=group)  # RCCL over xGMI
    return out


def shard_rows(n_rows: int, rank: int, world: int):
    """Contiguous row range [lo, hi) of rank `rank` (global batch 2048 -> 256 rows per rank)."""
    if n_rows % world != 0:
        raise ValueError(f"global batch {n_rows} is not divisible by world size {world}")
    per = n_rows // world
    return rank * per, (rank + 1) * per
