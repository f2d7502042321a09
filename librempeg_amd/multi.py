"""Frame sharding across the GPUs of a node (one process per GPU, torch.distributed / RCCL).

The path shards by independent frames: no data-path collective.  The only exchange is
a one-off broadcast of the context's table blob (filters, LUT constants, execution plan; KBs) from the
rank that ran the host-side init."""
import torch
import torch.distributed as dist

from .swscale import SwsContext


def shard_frames(nb_frames, rank, world):
    """frame i belongs to rank i % world (sws_scale_frames sharding rule)."""
    return list(range(rank, nb_frames, world))


def broadcast_context(ctx, src=0, device=None):
    """Give every rank a context identical to rank `src`'s (which must pass its SwsContext; others pass None)."""
    rank = dist.get_rank()
    dev = device or (f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else "cpu")
    blob = ctx.export_tables() if rank == src else b""
    n = torch.tensor([len(blob)], dtype=torch.int64, device=dev)
    dist.broadcast(n, src)
    buf = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    if rank == src:
        buf.copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
    dist.broadcast(buf, src)
    if rank != src:
        ctx = SwsContext(0, 0, "yuv420p", 0, 0, "yuv420p", 0, empty=True,
                         device=(torch.cuda.current_device() if dev != "cpu" else None))
        ctx.import_tables(buf.cpu().numpy().tobytes())
    return ctx
