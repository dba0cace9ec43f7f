"""Data-parallel exchange step (reference: DistributedDataParallel's gradient all-reduce,
distributed_image_translation.py:31-46,401-404,513-518).

Pure data parallelism: every rank holds full replicas (identical from the shared seed), draws its own
batch, keeps BatchNorm / feature-matching statistics local, and exchanges the flat gradient buffer of the
side being stepped (D_A+D_B or G_A+G_B), summed over ranks.  The division by the world size is folded into
the Adam kernel (``grad_scale``).  No BatchNorm-buffer broadcast (that is what breaks the reference's DDP
run) and no all-reduce of the side whose gradients are discarded (F5).

Two transports behind one ``ExchangeGroup``:
  * ``"capi"``  -- the library's own RCCL communicator (csrc/comm.hip: dg_dp_init / dg_dp_allreduce_sum / ...).
                  Collectives are enqueued on the CALLER's HIP stream (torch.cuda.current_stream()), so the trainer's
                  communication stream is one hop and per-bucket all-reduces are ordered by plain stream events.
                  The 128-byte RCCL unique id travels through the c10d store that torch.distributed already set up.
  * ``"c10d"``  -- torch.distributed collectives (backend ``nccl`` == RCCL too, or ``gloo`` for the CPU / one-GPU
                  rehearsal tests).
``"auto"`` = capi when the process group's backend is nccl and the tensors live on a HIP device, else c10d.
"""
from __future__ import annotations

import ctypes

import torch
import torch.distributed as dist

from . import _lib


def world_size(group=None) -> int:
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


class ExchangeGroup:
    def __init__(self, group=None, transport: str = "auto", device=None):
        self.group = group
        self.world = world_size(group)
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.device = device
        if transport == "auto":
            nccl = self.world > 1 and dist.get_backend(group) == "nccl"
            transport = "capi" if nccl else "c10d"
        if transport not in ("capi", "c10d"):
            raise ValueError(f"transport must be auto|capi|c10d, got {transport!r}")
        self.transport = transport
        self._scratch = None
        self.calls = 0
        if transport == "capi":
            self._init_capi()

    # ---- bootstrap of the library's communicator through the c10d store ---------------------------------
    def _init_capi(self):
        L = _lib.load()
        nbytes = L.dg_dp_unique_id_bytes()
        buf = ctypes.create_string_buffer(nbytes)
        if self.world > 1:
            payload = [None]
            if self.rank == 0:
                _lib.check(L.dg_dp_get_unique_id(buf, nbytes), "dg_dp_get_unique_id")
                payload = [bytes(buf.raw)]
            dist.broadcast_object_list(payload, src=0, group=self.group)     # host-side, through the process group
            raw = payload[0]
        else:
            _lib.check(L.dg_dp_get_unique_id(buf, nbytes), "dg_dp_get_unique_id")
            raw = bytes(buf.raw)
        idbuf = ctypes.create_string_buffer(raw, nbytes)
        if self.device is not None and torch.device(self.device).index is not None:
            torch.cuda.set_device(self.device)                    # ncclCommInitRank binds the CURRENT HIP device
        _lib.check(L.dg_dp_init(self.rank, self.world, idbuf, nbytes), "dg_dp_init")
        if L.dg_dp_world_size() != self.world:
            raise _lib.DiscoganHipError(f"RCCL communicator has {L.dg_dp_world_size()} ranks, expected {self.world}")
        self._scratch = torch.zeros(1, device=self.device if self.device is not None else "cuda", dtype=torch.float32)

    # ---- collectives (enqueue on the current stream; never synchronise) -----------------------------------
    def all_reduce_sum_(self, flat: torch.Tensor) -> float:
        """Sum ``flat`` (fp32, contiguous) over the ranks in place; returns the scale (1/W) that turns the sum into
        DDP's mean."""
        if self.world == 1 and self.transport != "capi":
            return 1.0
        self.calls += 1
        if self.transport == "capi":
            assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
            _lib.check(_lib.load().dg_dp_allreduce_sum(flat.data_ptr(), flat.numel(), torch.cuda.current_stream().cuda_stream),
                       "dg_dp_allreduce_sum")
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return 1.0 / self.world

    def broadcast_(self, flat: torch.Tensor, src: int = 0):
        """Initial replica sync (DDP constructor broadcast): only needed after rank 0 loaded
        checkpoints; replicas built from the shared seed are already identical.  A flat PARAMETER buffer written this way needs ``optim.Adam.refresh_derived()`` afterwards
        (bf16 shadow / f32x3 planes are derived from it)."""
        if self.world == 1 and self.transport != "capi":
            return
        if self.transport == "capi":
            _lib.check(_lib.load().dg_dp_broadcast(flat.data_ptr(), flat.numel(), src, torch.cuda.current_stream().cuda_stream),
                       "dg_dp_broadcast")
        else:
            dist.broadcast(flat, src=src, group=self.group)

    def barrier(self):
        if self.transport == "capi":
            _lib.check(_lib.load().dg_dp_barrier(self._scratch.data_ptr(), torch.cuda.current_stream().cuda_stream), "dg_dp_barrier")
            torch.cuda.current_stream().synchronize()
        elif self.world > 1:
            dist.barrier(group=self.group)

    def close(self):
        if self.transport == "capi":
            _lib.check(_lib.load().dg_dp_destroy(), "dg_dp_destroy")
            self.transport = "closed"

    def describe(self):
        if self.transport == "capi":
            return f"capi: library RCCL communicator ({_lib.load().dg_dp_world_size()} ranks), collectives on the caller's stream"
        be = dist.get_backend(self.group) if self.world > 1 else "none"
        return f"c10d: torch.distributed backend {be} ({self.world} ranks)"


def all_reduce_flat(flat: torch.Tensor, group=None, async_op: bool = False):
    """Sum ``flat`` over the ranks of ``group`` in place through torch.distributed.  Returns (scale, work)."""
    w = world_size(group)
    if w == 1:
        return 1.0, None
    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return 1.0 / w, work


def rank_data_seed(rank: int, base: int = 1000) -> int:
    """Per-rank synthetic-batch seed (seed 1000 + rank)."""
    return base + rank


def distributed_indices(n: int, world: int, rank: int, epoch: int, seed: int = 0, shuffle: bool = True):
    """The index list ``torch.utils.data.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True)``
    yields after ``set_epoch(epoch)`` (distributed_image_translation.py:203-208,451-452): ONE permutation per
    epoch from the shared seed ``seed + epoch``, padded by wrapping to a multiple of ``world``, rank r takes
    ``perm[r::world]`` -- disjoint 1/W shards, identical length on every rank."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    num = -(-n // world)
    total = num * world
    pad = total - len(idx)
    if pad > 0:
        idx += (idx * (-(-pad // len(idx))))[:pad] if pad > len(idx) else idx[:pad]
    return idx[rank:total:world]
