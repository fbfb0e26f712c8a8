"""Multi-GPU layer of the path: contiguous ray-range sharding + ONE all-gather of pixels.

Rays are independent; the only cross-ray op is the s*s mean, local to one LR pixel.
So the LR-pixel range is cut into ``world`` contiguous blocks (an LR pixel's s*s
sub-rays stay on one GPU), weights are replicated (4.77 MB, loaded per rank), every
rank renders its block, and one all-gather (RCCL over xGMI when the backend is
``nccl``; ``gloo`` in the CPU tests) assembles the image on every rank.  This replaces
the reference's nn.DataParallel scatter/gather inside every MLP call
(models/networks.py:54-69) — see SURVEY §2.3 / §8e.  One process per GPU.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous [lo, hi) blocks of ``cap = ceil(n_items / world)`` items; only the tail is shorter (possibly empty).

    With every block but the last at the same size the gathered buffer IS the result: rank k's rows land at
    ``[k * cap, ...)`` and the padding of the short tail sits at the very end (``all_gather_pixels`` returns a view,
    no compaction copy).  The imbalance is below ``world`` items (config #4: 5,954 LR pixels on seven ranks, 5,950
    on the eighth)."""
    if world <= 0 or n_items < 0:
        raise ValueError("world must be positive and n_items non-negative")
    cap = -(-n_items // world)
    return [(min(k * cap, n_items), min((k + 1) * cap, n_items)) for k in range(world)]


def _world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def all_gather_pixels(local: torch.Tensor, n_items: int, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Gather per-rank pixel blocks (n_local, c) into the full (n_items, c) tensor on every rank.

    ONE fixed-size ``all_gather_into_tensor`` (RCCL over xGMI under the ``nccl`` backend): the payload is LR pixels
    only -- 16 B per pixel, 95 KB per rank for config #4 -- so the exchange is latency-bound (tens of microseconds
    whatever algorithm RCCL picks on the point-to-point links); what matters is that it is a single collective with
    no host round trip and no copy after it (see ``shard_bounds``).  ``out``: an optional (world * cap, ...) receive buffer
    on ``local``'s device, re-used from step to step (a per-frame loop then allocates nothing)."""
    rank, world = _world(group)
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local
    bounds = shard_bounds(n_items, world)
    lo, hi = bounds[rank]
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} rows, expected {hi - lo}")
    c = local.shape[1:]
    cap = bounds[0][1] - bounds[0][0]
    send = local
    if hi - lo < cap:       # the tail rank(s): pad up to the common block size
        send = torch.zeros((cap, *c), dtype=local.dtype, device=local.device)
        send[: hi - lo] = local
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # development path only (N > 1 ranks sharing one GPU box, NSR_DIST_BACKEND=gloo): exchange through the host
        recv_h = torch.empty((world * cap, *c), dtype=local.dtype)
        dist.all_gather_into_tensor(recv_h, send.cpu().contiguous(), group=group)
        return recv_h[:n_items].to(local.device)
    if out is not None and (tuple(out.shape) != (world * cap, *c) or out.dtype != local.dtype or out.device != local.device):
        raise ValueError(f"out must be a ({world * cap}, ...) buffer of the payload's dtype on its device")
    recv = out if out is not None else torch.empty((world * cap, *c), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    return recv[:n_items]


def render_sharded(render_block: Callable[[int, int], torch.Tensor], n_lr: int, group=None) -> torch.Tensor:
    """Run ``render_block(lo, hi) -> (hi-lo, c)`` on this rank's LR-pixel block and all-gather.

    ``render_block`` is the per-GPU hot path (rays of LR pixels [lo, hi) -> LR pixel values);
    the result is the full (n_lr, c) image on every rank, bit-identical to a 1-GPU render
    because no reduction crosses a shard boundary.
    """
    rank, world = _world(group)
    lo, hi = shard_bounds(n_lr, world)[rank]
    return all_gather_pixels(render_block(lo, hi), n_lr, group)


def all_reduce_sum_(buffers, group=None) -> None:
    """Training-side exchange: in-place SUM all-reduce of each flat gradient buffer (one per network, 2.4 MB).

    Replaces DistributedDataParallel's bucketed gradient all-reduce (models/networks.py:84).  Every rank scales its
    loss by 1 / world (``Trainer.grad_scale``) so that the SUM is the gradient of the global-batch mean loss; the
    optimiser step then runs replicated and the weights stay bit-identical across ranks.  No-op for world == 1."""
    rank, world = _world(group)
    if world == 1:
        return
    for b in buffers:
        if b.is_cuda and dist.get_backend(group) == "gloo":     # development path, see all_gather_pixels
            h = b.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            b.copy_(h)
        else:
            dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)


def all_ranks_agree(ok: bool, device=None, group=None) -> bool:
    """Logical AND of ``ok`` over the ranks of ``group`` (one tiny MIN all-reduce; plain ``ok`` for world == 1): the
    collective form of an error check, so that every rank takes the same branch."""
    rank, world = _world(group)
    if world == 1:
        return bool(ok)
    on_dev = dist.get_backend(group) != "gloo" and device is not None
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if on_dev else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()) == 1)


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise the default process group from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, local_rank, world).  No-op for world == 1."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        # one process per GPU; with fewer GPUs than ranks (a 1-GPU box exercising the N > 1 code path under
        # NSR_DIST_BACKEND=gloo -- RCCL itself refuses two ranks on one device) the ranks share devices round-robin
        local = local % torch.cuda.device_count()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("NSR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
            if backend == "nccl":
                # bind the communicator to this rank's device at creation (no lazy guess from the first collective's tensor;
                # RCCL otherwise warns that the device is unknown until then)
                kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local, world
