"""Column-sharded multi-GPU plumbing (one process per GPU, torch.distributed for rendezvous only).

The data-path exchange itself runs inside libnmfx.so on RCCL: by default a reduce-scatter of X_g H_g' by row blocks (grouped
with the all-reduce of the small [H_g H_g' | rowsum(H_g) | H statistics] tail) and an all-gather of the updated row blocks of W
per outer iteration (`replicated_w` keeps round 1's single packed all-reduce).  This module only (a) decides which columns a rank owns and (b) ships rank 0's RCCL unique id to the
other ranks.  The reference has no distributed path."""
from __future__ import annotations


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous column shard [c0, c1) of `rank`; the first n % world ranks get one extra column."""
    if not (0 <= rank < world) or n < world:
        raise ValueError("need 0 <= rank < world <= n")
    base, rem = divmod(n, world)
    c0 = rank * base + min(rank, rem)
    return c0, c0 + base + (1 if rank < rem else 0)


def packed_layout(P: int, K: int) -> dict:
    """Offsets (in elements of T) of the per-iteration all-reduce payload  [ X_g H_g' | H_g H_g' | rowsum(H_g) ]."""
    return {"XHt": (0, P * K), "HHt": (P * K, P * K + K * K), "sH": (P * K + K * K, P * K + K * K + K),
            "count": P * K + K * K + K}


def broadcast_unique_id(make_id, group=None) -> bytes:
    """rank 0 calls make_id() (-> 128 bytes from nmfx_comm_get_unique_id); every rank returns the same bytes."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    box = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    uid = box[0]
    if not isinstance(uid, (bytes, bytearray)) or len(uid) != 128:
        raise RuntimeError("unique id broadcast failed")
    return bytes(uid)


def init_comm(ctx, group=None, transport="rccl"):
    """Attach a communicator spanning `group` (default: world) to a Context.
    transport = "rccl": RCCL over xGMI.  "p2p": RCCL for the bootstrap and as the fallback, the exchange itself over the ranks'
    peer-mapped windows (all xGMI links at once, device-side flags; csrc/peer.hpp).  "p2p_only": the windows alone, no RCCL
    communicator at all (also works with several ranks on ONE device, where RCCL refuses duplicate GPUs)."""
    import torch.distributed as dist
    from .api import comm_unique_id
    if transport not in ("rccl", "p2p", "p2p_only"):      # before any collective work: a typo must not leave a half-built communicator
        raise ValueError(f"unknown transport {transport!r}")
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if transport == "p2p_only":
        ctx.comm_init_p2p(rank, world)
    else:
        uid = broadcast_unique_id(comm_unique_id, group)
        ctx.comm_init(uid, rank, world)
    if transport in ("p2p", "p2p_only"):
        attach_p2p(ctx, group)


def attach_p2p(ctx, group=None):
    """Export this rank's window, all-gather the 128-byte handles through the host, map the peers' windows.  Collective and
    all-or-nothing: a rank that cannot export or map (no IPC support, no peer access) makes EVERY rank raise, so that callers can fall
    back to another transport together instead of waiting for each other."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    try:
        mine, err = ctx.comm_p2p_export(), None
    except Exception as e:  # noqa: BLE001
        mine, err = None, repr(e)
    allh = [None] * world
    dist.all_gather_object(allh, (mine, err), group=group)
    bad = [(r, e) for r, (h, e) in enumerate(allh) if h is None]
    if bad:
        raise RuntimeError(f"peer windows: export failed on rank(s) {bad}")
    try:
        ctx.comm_p2p_attach([h for h, _ in allh])
        err = None
    except Exception as e:  # noqa: BLE001
        err = repr(e)
    alle = [None] * world
    dist.all_gather_object(alle, err, group=group)
    bad = [(r, e) for r, e in enumerate(alle) if e is not None]
    if bad:
        # all-or-nothing: the ranks where mapping succeeded detach again, so that a caller who falls back to the wrapped transport
        # on the same context finds every rank routing by the same rule (windows on some ranks, RCCL on others would never pair up)
        try:
            ctx.comm_p2p_attach(None)
        except Exception:  # noqa: BLE001
            pass
        raise RuntimeError(f"peer windows: mapping failed on rank(s) {bad}; every rank is back on the wrapped transport")
