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

ncclCommInitRank is a blocking collective: a rank that fails before reaching it would leave the others blocked inside it.
``guarded_bootstrap`` therefore (1) lets every rank do its LOCAL preparation and publish the outcome in the c10d store,
(2) reads all outcomes on every rank -- a vote that needs no collective -- and only if all are ready (3) enters the
collective in a helper thread under a deadline (``DG_COMM_INIT_TIMEOUT_S``, default 120 s); a rank still blocked at the
deadline prints its state and exits non-zero.  A failed vote is identical on every rank: ``"auto"`` then uses c10d on all
ranks (``ExchangeGroup.note`` says so), an explicit ``"capi"`` raises on all ranks.  The unique id and the votes travel as
store keys, so the bootstrap never touches torch.distributed's own RCCL communicator (lazily created by its first device
collective): a run on the capi transport has ONE RCCL instance.
"""
from __future__ import annotations

import ctypes
import datetime
import os
import sys
import threading
import time

import torch
import torch.distributed as dist

from . import _lib

HANG_EXIT_CODE = 17          # a rank that gives up on a blocked bootstrap leaves with this status


def world_size(group=None) -> int:
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def init_timeout_s() -> float:
    """Deadline of every bootstrap phase (env DG_COMM_INIT_TIMEOUT_S, default 120 s)."""
    return float(os.environ.get("DG_COMM_INIT_TIMEOUT_S", "120"))


class BootstrapVoteFailed(_lib.DiscoganHipError):
    """At least one rank could not prepare its side of the communicator.  Raised on EVERY rank, before any rank has
    entered the collective init: the ranks are still in step and may agree on another transport."""


def _die(msg):
    """A rank blocked in (or waiting on ranks blocked in) a collective bootstrap cannot be unblocked: say what this rank
    was doing and leave non-zero.  Never re-exec: a relaunch is a fresh child of a parent that has not touched the GPU
    (bench.py::self_launch, launch.py)."""
    print(f"[dg_dp] FATAL {msg}", file=sys.stderr, flush=True)
    os._exit(HANG_EXIT_CODE)


def store_of(group=None):
    """The c10d key-value store behind a process group (TCPStore / FileStore, prefixed per group): host-side, no collective,
    no RCCL -- the bootstrap's out-of-band channel."""
    from torch.distributed import distributed_c10d as c10d
    if group is None or group is dist.group.WORLD:
        return c10d._get_default_store()
    return c10d._world.pg_map[group][1]


_TAGS = {}


def _next_tag(kind, group):
    """Every rank creates its exchange groups / host collectives in the same order, so a per-process counter names the same
    rendezvous on all of them."""
    k = (kind, id(group) if group is not None else 0)
    _TAGS[k] = _TAGS.get(k, 0) + 1
    return f"dg/{kind}/{_TAGS[k]}"


def _gather_keys(store, tag, what, rank, world, mine, timeout_s, on_hang):
    """Every rank publishes ``mine`` under its key and reads all the others'; a key that does not appear within the
    deadline means that rank is gone or blocked -> on_hang (default: exit non-zero)."""
    store.set(f"{tag}/{what}/{rank}", mine)
    keys = [f"{tag}/{what}/{r}" for r in range(world)]
    try:
        store.wait(keys, datetime.timedelta(seconds=timeout_s))
    except Exception as e:      # noqa: BLE001  (c10d raises DistStoreError / RuntimeError depending on the store)
        on_hang(f"rank {rank}/{world}: phase '{what}' of {tag}: not every rank reported within {timeout_s:.0f} s ({type(e).__name__})")
        raise
    return [bytes(store.get(k)) for k in keys]


def guarded_bootstrap(store, rank, world, prepare, init, tag, timeout_s=None, on_hang=_die, log=None):
    """Bring up a communicator whose init is a blocking collective without ever leaving ranks blocked in it.

      1. ``prepare()`` -- local work only (dlopen, device check; rank 0 also returns the unique id).  Its outcome is
         published in the store and every rank reads every outcome: the VOTE.  If any rank failed, every rank raises
         BootstrapVoteFailed -- nobody has entered the collective.
      2. ``init(unique_id)`` -- the collective -- runs in a helper thread under a deadline.  A rank still blocked at the
         deadline prints its state and exits non-zero (``on_hang``); so does a rank whose peers do not confirm in step 3.
      3. every rank publishes the outcome of its init and reads all of them; a failure anywhere raises everywhere.

    Reference: the blocking rendezvous of dist.init_process_group, distributed_image_translation.py:31-38."""
    timeout_s = init_timeout_s() if timeout_s is None else float(timeout_s)
    say = log or (lambda m: None)
    err, uid = "", None
    try:
        uid = prepare()
    except Exception as e:      # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    if rank == 0 and not err:
        store.set(f"{tag}/uid", bytes(uid))
    votes = _gather_keys(store, tag, "ready", rank, world, b"ok" if not err else b"ERR " + err.encode(), timeout_s, on_hang)
    bad = [(r, v[4:].decode("utf-8", "replace")) for r, v in enumerate(votes) if v != b"ok"]
    if bad:
        raise BootstrapVoteFailed(f"{tag}: {len(bad)} of {world} ranks not ready, no rank entered the collective init: "
                                  + "; ".join(f"rank {r}: {m}" for r, m in bad))
    uid = bytes(store.get(f"{tag}/uid"))
    say(f"{tag}: all {world} ranks ready, entering the collective init (deadline {timeout_s:.0f} s)")
    box = {}

    def run():
        try:
            box["ret"] = init(uid)
        except BaseException as e:      # noqa: BLE001
            box["err"] = f"{type(e).__name__}: {e}"

    t0 = time.time()
    th = threading.Thread(target=run, name="dg_dp_init", daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        on_hang(f"rank {rank}/{world}: {tag}: still inside the collective init after {time.time() - t0:.0f} s "
                f"(DG_COMM_INIT_TIMEOUT_S={timeout_s:.0f}); every rank had voted ready")
        raise TimeoutError(f"{tag}: collective init did not return")
    ierr = box.get("err", "")
    done = _gather_keys(store, tag, "init", rank, world, b"ok" if not ierr else b"ERR " + ierr.encode(), timeout_s, on_hang)
    bad = [(r, v[4:].decode("utf-8", "replace")) for r, v in enumerate(done) if v != b"ok"]
    if bad:
        raise _lib.DiscoganHipError(f"{tag}: collective init failed on " + "; ".join(f"rank {r}: {m}" for r, m in bad))
    return box.get("ret")


def host_allgather(value: float, group=None, timeout_s=None):
    """[value of rank 0, ..., value of rank W-1] through the c10d store: host-side, no collective, no device work
    (bench.py: max-over-ranks timing without a second RCCL communicator)."""
    w = world_size(group)
    if w == 1:
        return [float(value)]
    rank = dist.get_rank(group)
    vals = _gather_keys(store_of(group), _next_tag("host", group), "v", rank, w, repr(float(value)).encode(),
                        init_timeout_s() if timeout_s is None else timeout_s, _die)
    return [float(v.decode()) for v in vals]


def host_barrier(group=None, timeout_s=None):
    host_allgather(0.0, group, timeout_s)


class ExchangeGroup:
    def __init__(self, group=None, transport: str = "auto", device=None, log=None):
        self.group = group
        self.world = world_size(group)
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.device = device
        self.note = None             # set when "auto" had to leave the library communicator for c10d
        self.noop = os.environ.get("DG_DP_NOOP", "0") == "1"      # collectives stubbed: bench.py's exposed-time leg
        auto = transport == "auto"
        if auto:
            nccl = self.world > 1 and dist.get_backend(group) == "nccl"
            transport = "capi" if nccl else "c10d"
        if transport not in ("capi", "c10d"):
            raise ValueError(f"transport must be auto|capi|c10d, got {transport!r}")
        self.transport = transport
        self._scratch = None
        self.calls = 0
        if transport == "capi":
            try:
                self._init_capi(log)
            except BootstrapVoteFailed as e:
                # the vote is the same on every rank and nobody has entered the collective: with "auto" all ranks move to
                # torch.distributed's RCCL together, visibly; an explicit "capi" raises on every rank
                if not auto:
                    raise
                self.transport = "c10d"
                self.note = f"library RCCL communicator unavailable ({e}); all ranks use c10d"
                print(f"[dg_dp] rank {self.rank}: {self.note}", file=sys.stderr, flush=True)

    # ---- bootstrap of the library's communicator through the c10d store ---------------------------------
    def _init_capi(self, log=None):
        def prepare():
            L = _lib.load()
            if self.device is not None and torch.device(self.device).index is not None:
                torch.cuda.set_device(self.device)                # ncclCommInitRank binds the CURRENT HIP device
            dev = ctypes.c_int(-1)
            _lib.check(L.dg_dp_ready(ctypes.byref(dev)), "dg_dp_ready")
            want = torch.device(self.device).index if self.device is not None else None
            if want is not None and dev.value != want:
                raise _lib.DiscoganHipError(f"current HIP device is {dev.value}, the exchange group was built for cuda:{want}")
            fail = os.environ.get("DG_COMM_TEST_FAIL_RANK")       # test hook: this rank reports "not ready"
            if fail is not None and int(fail) == self.rank:
                raise _lib.DiscoganHipError("DG_COMM_TEST_FAIL_RANK: failing on purpose before the collective")
            if self.rank != 0:
                return None
            nbytes = L.dg_dp_unique_id_bytes()
            buf = ctypes.create_string_buffer(nbytes)
            _lib.check(L.dg_dp_get_unique_id(buf, nbytes), "dg_dp_get_unique_id")
            return bytes(buf.raw)

        def init(uid):
            L = _lib.load()
            if self.device is not None and torch.device(self.device).index is not None:
                torch.cuda.set_device(self.device)                # the helper thread has its own current device
            idbuf = ctypes.create_string_buffer(uid, len(uid))
            _lib.check(L.dg_dp_init(self.rank, self.world, idbuf, len(uid)), "dg_dp_init")
            if L.dg_dp_world_size() != self.world:
                raise _lib.DiscoganHipError(f"RCCL communicator has {L.dg_dp_world_size()} ranks, expected {self.world}")

        if self.world > 1:
            guarded_bootstrap(store_of(self.group), self.rank, self.world, prepare, init, _next_tag("xg", self.group), log=log)
        else:
            init(prepare())
        self._scratch = torch.zeros(1, device=self.device if self.device is not None else "cuda", dtype=torch.float32)

    # ---- collectives (enqueue on the current stream; never synchronise) -----------------------------------
    def all_reduce_sum_(self, flat: torch.Tensor) -> float:
        """Sum ``flat`` (fp32, contiguous) over the ranks in place; returns the scale (1/W) that turns the sum into
        DDP's mean."""
        if self.world == 1 and self.transport != "capi":
            return 1.0
        self.calls += 1
        if self.noop:
            return 1.0 / self.world
        if self.transport == "capi":
            assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
            _lib.check(_lib.load().dg_dp_allreduce_sum(flat.data_ptr(), flat.numel(), torch.cuda.current_stream().cuda_stream),
                       "dg_dp_allreduce_sum")
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return 1.0 / self.world

    def all_reduce_max_(self, flat: torch.Tensor):
        """Max over the ranks in place (timings)."""
        if self.world == 1 and self.transport != "capi":
            return flat
        if self.transport == "capi":
            assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
            _lib.check(_lib.load().dg_dp_allreduce_max(flat.data_ptr(), flat.numel(), torch.cuda.current_stream().cuda_stream),
                       "dg_dp_allreduce_max")
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.MAX, group=self.group)
        return flat

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
