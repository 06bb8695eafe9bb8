"""One process per GPU; pairs are independent, so the only collectives on the path are a one-off
broadcast of the packed weight blob (RCCL over xGMI, backend "nccl" on ROCm) and a final gather of the
small per-pair result records.  The reference shards the same way with independent background
processes (scripts/test_kitti.sh:45-75) and has no communication at all.

Everything here works on CPU tensors with the ``gloo`` backend too, which is how the world_size-2
tests exercise it without GPUs.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str | None = None):
    """Initialise the default process group from the torchrun environment (no-op for world size 1)."""
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard(n_items: int, rank: int, world: int):
    """Static round-robin: item i belongs to rank ``i % world``."""
    return list(range(rank, n_items, world))


def broadcast_blob(blob: torch.Tensor, src: int = 0) -> torch.Tensor:
    """Broadcast the packed fp32 weight blob in place (one ~35 MB message; 7 direct xGMI links from
    the root on an 8-GPU node, so it is a sub-millisecond one-off and never on the steady-state path)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(blob, src=src)
    return blob


def _aligned_blob(n: int, device) -> torch.Tensor:
    """``n`` floats at a 256-byte aligned address (``eyoc_model_create`` requires it)."""
    raw = torch.zeros(n + 64, dtype=torch.float32, device=device)
    off = ((-raw.data_ptr()) % 256) // 4
    return raw[off:off + n]


def broadcast_model(model, device, src: int = 0):
    """Rank ``src`` packs its parameters; everyone else adopts the broadcast blob (one message).

    On a GPU the blob is packed into / received in device memory (RCCL) and every rank ends with a device-side model.
    With ``device = cpu`` (gloo; tests/test_dist_gloo.py) the same message travels between host buffers
    (``model.pack_host()`` on ``src``) and nothing is adopted - the returned tensor is what a GPU rank would adopt."""
    device = torch.device(device)
    rank = dist.get_rank() if dist.is_initialized() else 0
    on_gpu = device.type == "cuda"
    if rank == src:
        blob = model.pack(device) if on_gpu else model.pack_host()
    else:
        blob = _aligned_blob(model.blob_floats(), device)
    broadcast_blob(blob, src)
    if rank != src and on_gpu:
        model.pack(device, blob=blob, from_blob=True)
    return blob


def gather_records(records: torch.Tensor) -> torch.Tensor:
    """All-gather per-pair result rows ``f32 [n_local, R]`` (equal n_local on every rank) -> ``[world*n_local, R]``
    ordered so that global pair i sits at row i under the round-robin sharding."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return records
    world = dist.get_world_size()
    parts = [torch.empty_like(records) for _ in range(world)]
    dist.all_gather(parts, records.contiguous())
    stacked = torch.stack(parts, 1)                 # [n_local, world, R]: row-major == round-robin order
    return stacked.reshape(-1, records.shape[1])


def gather_records_ragged(records: torch.Tensor, n_total: int) -> torch.Tensor:
    """The same for a split that does not divide evenly (LoKITTI_50: 545 pairs over 8 ranks = 69 / 68 per rank):
    rank r holds the rows of pairs ``r, r + W, ...`` (``ceil`` or ``floor`` of ``n_total / W`` of them); shorter
    ranks are padded with NaN rows for the collective and the padding is cut off again -> ``[n_total, R]`` in
    global pair order on every rank."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return records[:n_total]
    world = dist.get_world_size()
    per = (n_total + world - 1) // world
    pad = per - records.shape[0]
    if pad < 0:
        raise ValueError("a rank holds more rows than ceil(n_total / world)")
    if pad:
        records = torch.cat([records, torch.full((pad, records.shape[1]), float("nan"), dtype=records.dtype,
                                                 device=records.device)])
    return gather_records(records)[:n_total]


def spawn_ranks(fn, world: int, args=()):
    """Launch ``world`` ranks of ``fn(rank, world, *args)`` on this node when no launcher did
    (``python bench.py --gpus N`` without torchrun): each child gets the RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* environment torchrun would have set, then ``init()`` forms the RCCL (or gloo) group as usual."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_spawn_entry, args=(world, port, fn, args), nprocs=world, join=True)


def _spawn_entry(rank, world, port, fn, args):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    fn(rank, world, *args)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
