"""Data-parallel exchange step: the image batch is sharded over the GPUs of one node, every rank
encodes its shard, and the L2-normalised embeddings [B_local, E] are all-gathered before the
cosine-logit matrix (BASELINE.json north_star; nothing of this exists in the reference, which is
single-process: SURVEY.md section 8e).  One process per GPU; backend "nccl" is RCCL over xGMI on
ROCm, "gloo" on CPU for the tests.  Payload: 256 x 512 fp32 = 512 KiB per rank -- latency-bound."""
import torch
import torch.distributed as dist


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def all_gather_rows(x: torch.Tensor, group=None, force: bool = False) -> torch.Tensor:
    """[B_local, E] on every rank -> [world * B_local, E], rank-major rows.  Identity at world size 1 (`force=True`
    still issues the collective on an initialised group of one rank: the RCCL call path on a 1-GPU box)."""
    if not (is_distributed() or (force and dist.is_available() and dist.is_initialized())):
        return x
    x = x.contiguous()
    world = dist.get_world_size(group)
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
