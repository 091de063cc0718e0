"""Multi-GPU use of the decoder hot path: one process per GPU, utterances sharded
across ranks, no communication inside the step loop.

The path shards by independent utterances (tacotron/decoder_cell.py:180-195 is row-wise
in the batch), so the only collective is the one-off broadcast of the packed weight
blob from rank 0 (RCCL over xGMI when the backend is "nccl").  The reference's stop
rule is batch-global (tacotron/decoder.py:68); each shard applies it to its own
utterances, which is what the reference's own nn.DataParallel replicas do
(tacotron/train_util.py:215).  ``global_stop_step`` offers the whole-batch semantics
as one MIN all-reduce of a scalar after decoding."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) slice of the batch owned by `rank`
    (the first n_items % world_size ranks get one extra utterance)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(n_items, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_batch(t: torch.Tensor, world_size: int, rank: int) -> torch.Tensor:
    b, e = shard_bounds(t.shape[0], world_size, rank)
    return t[b:e]


def broadcast_blob(blob: Optional[torch.Tensor], nbytes: int, device, src: int = 0, group=None) -> torch.Tensor:
    """Rank `src` passes its packed weight blob; every rank returns a tensor holding the
    same bytes.  One collective, issued once at start-up."""
    if dist.get_rank(group) == src:
        if blob is None or blob.numel() * blob.element_size() != nbytes:
            raise ValueError("source rank must supply a blob of nbytes bytes")
        buf = blob
    else:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=src, group=group)
    return buf


def broadcast_engine_weights(engine, tensors, src: int = 0, group=None) -> None:
    """Pack on rank `src`, broadcast the blob, bind it on every other rank."""
    if dist.get_rank(group) == src:
        engine.ensure_packed(tensors)
        broadcast_blob(engine.blob, engine.packed_bytes(), engine.device, src, group)
    else:
        engine.bind(broadcast_blob(None, engine.packed_bytes(), engine.device, src, group))


def global_stop_step(local_steps: int, device, group=None) -> int:
    """Whole-batch stop semantics: the number of frames the reference would have produced
    for the unsharded batch is the minimum over shards (the first shard to fire stops all)."""
    t = torch.tensor([local_steps], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item())


def gather_outputs(local: torch.Tensor, sizes: List[int], group=None) -> torch.Tensor:
    """All-gather variable-size batch shards back into one tensor (optional; outputs
    normally stay on the GPU that produced them)."""
    world = dist.get_world_size(group)
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)
