"""Token-sequence parallelism for the DiT (BASELINE.json config #4).

Every op of a DiT block except self-attention is token-local, so the (f h w) token axis is cut
into ``world`` contiguous shards (one per GPU / process).  The only exchange is the per-layer
all-gather of the post-RoPE K and V shards (RCCL over xGMI through torch.distributed; the `nccl`
backend IS RCCL on ROCm).  The reference has no such path — it runs one dense sequence on one
GPU [R infinicube/inference/guidance_buffer_generation.py:759-766].
"""

from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch


# Default cap on RCCL's channel count for the K|V transports that run RCCL kernels (every channel = one resident work-group beside
# an attention launch that runs one work-group per CU).  Measured on one MI355X (profiles/r05/kv_contention.md): the first
# resident copy work-group that holds LDS already costs the shard-shape attention 6-18 %, the next seven add ~3 points; light ones are
# nearly free up to 8 but cost +15-18 % from 16 on at the sp8 shape; 8 channels move 150-190 GB/s, above what either layout needs to hide its exchange (sp8: 126 GB/s).
RCCL_MAX_CHANNELS_DEFAULT = 8


def apply_rccl_channel_cap(max_channels: int = -1, env=None) -> str:
    """Set NCCL_MAX_NCHANNELS before the communicators are created: -1 = RCCL_MAX_CHANNELS_DEFAULT unless the variable is already
    set (a user's own value wins), 0 = leave RCCL's own choice, k > 0 = k.  Returns the value in force ('' = RCCL's own)."""
    env = os.environ if env is None else env
    if max_channels > 0:
        env["NCCL_MAX_NCHANNELS"] = str(max_channels)
    elif max_channels < 0:
        env.setdefault("NCCL_MAX_NCHANNELS", str(RCCL_MAX_CHANNELS_DEFAULT))
    return env.get("NCCL_MAX_NCHANNELS", "")


@dataclass(frozen=True)
class ShardPlan:
    world: int
    rank: int
    S: int          # global token count
    tok0: int       # first global token of this shard
    n_tok: int      # tokens in this shard

    @staticmethod
    def make(S: int, world: int = 1, rank: int = 0) -> "ShardPlan":
        if world < 1 or not (0 <= rank < world):
            raise ValueError(f"bad (world, rank) = ({world}, {rank})")
        if S % world:
            raise ValueError(
                f"token count S={S} is not divisible by world={world}; choose a GPU count that divides "
                f"the sequence (93f 480p: S=37440 = 2^6*3^2*5*13)")
        n = S // world
        return ShardPlan(world, rank, S, rank * n, n)


@dataclass
class ParallelLayout:
    """How ``world`` ranks split one generation.

    ``sp``     : every rank owns S/world tokens and runs BOTH CFG forwards (cond, uncond); K/V all-gather over
                 all ranks, 2 x L gathers per step.
    ``cfg+sp`` : the ranks form two branch groups of world/2 — group 0 runs the cond forward, group 1 the uncond
                 forward — and the token sequence is sharded over the world/2 ranks of a group.  Same FLOPs per
                 rank, but only L gathers per step, over half as many peers, each hidden under an attention that
                 is twice as long; at world == 2 there is no K/V exchange at all.  The two ranks that own the same
                 token shard (rank r and r + world/2) swap their [n_tok, 64] velocity tokens once per step (a few
                 MB) and both apply the same CFG + Euler update, so their latents stay bit-identical.
    xGMI is point-to-point (7 links per GPU): fewer, larger exchanges among fewer peers is the layout that
    suits it, so ``auto`` picks ``cfg+sp`` whenever CFG is on and ``world`` is even.
    """
    world: int
    rank: int
    mode: str                 # "sp" | "cfg+sp"
    sp_world: int
    sp_rank: int
    branch: Optional[int]     # cfg+sp: 0 = cond, 1 = uncond; sp: None (both)
    sp_group: object = None   # process group of the ranks sharing this rank's branch (None = default group)
    pair_group: object = None  # cfg+sp: (cond rank, uncond rank) owning the same token shard

    @staticmethod
    def make(world: int = 1, rank: int = 0, mode: str = "auto", use_cfg: bool = True, init_groups: bool = True) -> "ParallelLayout":
        """Creating the groups is a COLLECTIVE of the whole world: every rank must call this the same number of times, in
        the same order.  A caller that builds pipelines asymmetrically (a second WanVideoGenerator on a live worker pool:
        only rank 0 gets a new pipeline object) must reuse the first layout — WanVideoPipeline keeps its layouts in the
        pool's cache for exactly that (multigpu.WorkerPool.layouts)."""
        if mode not in ("auto", "sp", "cfg+sp"):
            raise ValueError(f"parallelism must be 'auto', 'sp' or 'cfg+sp', got {mode!r}")
        if mode == "auto":
            mode = "cfg+sp" if (use_cfg and world % 2 == 0) else "sp"
        if mode == "cfg+sp" and (world % 2 or not use_cfg):
            raise ValueError("cfg+sp needs an even number of ranks and classifier-free guidance switched on")
        if mode == "sp" or world == 1:
            return ParallelLayout(world, rank, "sp", world, rank, None)
        half = world // 2
        lay = ParallelLayout(world, rank, "cfg+sp", half, rank % half, rank // half)
        if init_groups:
            import torch.distributed as dist
            # every rank must create every group, in the same order
            sp_groups = [dist.new_group(list(range(b * half, (b + 1) * half))) for b in range(2)]
            pair_groups = [dist.new_group([i, i + half]) for i in range(half)]
            lay.sp_group, lay.pair_group = sp_groups[lay.branch], pair_groups[lay.sp_rank]
        return lay

    def shard_plan(self, S: int) -> "ShardPlan":
        return ShardPlan.make(S, self.sp_world, self.sp_rank)


class BranchExchange:
    """cfg+sp: once per step the cond rank and the uncond rank of a token shard swap velocity tokens.
    ``own`` f32 [n, c] (this rank's branch) -> ``both`` f32 [2, n, c] with slot 0 = cond, slot 1 = uncond
    (the pair group lists the cond rank first, so an all-gather lands them in that order)."""

    def __init__(self, layout: ParallelLayout):
        if layout.mode != "cfg+sp":
            raise ValueError("BranchExchange is only meaningful for the cfg+sp layout")
        import torch.distributed as dist
        self.dist, self.group, self.branch = dist, layout.pair_group, layout.branch

    def __call__(self, own: torch.Tensor, both: torch.Tensor) -> None:
        assert own.is_contiguous() and both.is_contiguous() and both.shape[0] == 2 and both.shape[1:] == own.shape
        self.dist.all_gather_into_tensor(both.view(2 * own.shape[0], *own.shape[1:]), own, group=self.group)


class KVGather:
    """Chunked, asynchronous exchange of the local K|V rows among the ranks of a sequence-parallel group.

    The DiT keeps the post-RoPE keys and the values of its shard as ONE bf16 matrix [n, 2d] (row = k(d) | v(d)), so one
    collective per row-chunk moves both.  ``start(rows, out)`` enqueues the exchange of ONE row-chunk ([m, 2d] from every
    rank -> [world*m, 2d], rank-major) and returns a handle; ``wait(handle)`` makes the compute stream wait for that
    chunk only.  With the ``nccl`` backend (= RCCL over xGMI) the transfers run on RCCL's own stream, so chunk c+1 is in
    flight while attention consumes chunk c (dit.WanDiT._sp_attention).  Attention is invariant to key order, so a
    chunk does not have to be a contiguous range of global tokens.

    ``mode`` (env ICV_KV_EXCHANGE, ``bench.py --kv-exchange``):
      * ``"allgather"``: ``all_gather_into_tensor`` — RCCL picks the schedule (a ring costs (world-1) hops);
      * ``"p2p"``: one grouped ``isend``/``irecv`` pair per peer (``batch_isend_irecv`` = ncclGroupStart/End around
        ncclSend/ncclRecv) — the fully-connected schedule xGMI is built for: each of the 7 links of a GPU carries exactly
        one peer's shard, once (0.63 ms instead of 4.4 ms per 14B layer at world 8 if RCCL rings);
      * ``"native"``: ``icv_allgather_kv`` of libicvideo on this group's own RCCL communicator (``icv_comm_create``; the id
        travels through the torch.distributed group once) and a dedicated HIP stream fenced with events — the same
        transfer without torch.distributed in the per-layer path (SURVEY §8b's C export; GPU ranks only);
      * ``"ipc"``: no RCCL and no kernel MOVING ROWS in the per-layer path (csrc/ipc.hip, ``icv_ipc_*``): the local rows live
        in a symmetric heap every peer has opened through hipIpc, each rank PULLS its peers' chunks with copy-engine
        ``hipMemcpyAsync`` on one stream per peer; readiness / reuse are flag words in shared host memory
        (``hipStreamWriteValue32`` / ``hipStreamWaitValue32``: on this runtime small kernels - the wait spins on ONE wave for
        the skew between two ranks, measured with tools/probe_streamops.py).  The RCCL modes run channel kernels that stay
        resident for the whole transfer on CUs the one-work-group-per-CU attention also wants
        (profiles/r05/kv_contention.md); this one leaves the CUs to the attention while the rows travel.  GPU ranks only; the
        torch.distributed group carries the 72-byte handles once.  The rows handed to ``start`` must come from
        ``local_rows`` (the heap)."""

    MODES = ("allgather", "p2p", "native", "ipc")

    def __init__(self, plan: ShardPlan, group=None, mode: Optional[str] = None):
        self.plan = plan
        self.group = group
        self.mode = mode or os.environ.get("ICV_KV_EXCHANGE", "allgather")
        if self.mode not in self.MODES:
            raise ValueError(f"K/V exchange mode must be one of {self.MODES}, got {self.mode!r}")
        self.timing = None          # bench: a list -> (event before wait, event after wait) per chunk = exposed transfer time
        self.n_collectives = 0
        self._native = None
        self.dist, self.peers = None, [0]
        if plan.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError("KVGather with world>1 needs torch.distributed to be initialised")
            self.dist = dist
            # global ranks of the group members, in group-rank order (the order all_gather lands the shards in)
            self.peers = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
            if len(self.peers) != plan.world:
                raise ValueError(f"K/V exchange group has {len(self.peers)} ranks, the shard plan {plan.world}")
        if self.mode == "native":
            self._native = _NativeComm.for_group(self.dist, group, self.peers, plan.rank, plan.world)
        self._heap = None           # mode "ipc": the symmetric heap (made by reserve())
        self._arrival_ops = None    # enable_arrival(): the consumer gates on arrival flags inside ONE attention launch per layer
        self._flags, self._flag_seq, self._flag_stream = None, 0, None

    # ---- where the local K|V rows live ---------------------------------------------------------------------------------
    def reserve(self, nbytes: int, device) -> None:
        """Mode "ipc": create the symmetric heap of ``nbytes`` on every rank of the group, exchange the handles, open the
        peers, self-test (a COLLECTIVE of the group: every rank calls it at the same point with the same size; raises the SAME
        error on every rank when any rank fails).  Other modes: nothing to do."""
        if self.mode != "ipc":
            return
        if self._heap is not None:
            self._heap.close()
        self._heap = _IpcHeap(self.dist, self.group, self.peers, self.plan.rank, self.plan.world, int(nbytes), device)

    def local_rows(self, rows: int, cols: int, dtype, alloc):
        """The [rows, cols] matrix this rank writes its K|V rows into: carved out of the symmetric heap in mode "ipc" (so the
        peers can pull it), a plain ``alloc((rows, cols), dtype)`` otherwise."""
        if self.mode != "ipc":
            return alloc((rows, cols), dtype)
        if self._heap is None:
            raise RuntimeError("KVGather mode 'ipc': reserve() the symmetric heap before asking for local rows")
        return self._heap.carve(rows, cols, dtype)

    def acquire(self) -> None:
        """Call BEFORE the kernels that overwrite the local rows (the K|V GEMM of the next layer): the launch stream waits until
        every peer has pulled everything published so far.  A no-op for the collective modes (their sends complete in stream
        order before the next writer starts)."""
        if self._heap is not None:
            self._heap.acquire()

    def allreduce_max(self, t: torch.Tensor) -> None:
        """In-place max over the ranks of the group (the e4m3 wire format's per-head K / V abs-max: 2 x heads floats per layer).
        Stream-ordered on nccl groups (RCCL); a blocking host call on gloo groups."""
        if self.plan.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)

    def close(self) -> None:
        if self._heap is not None:
            self._heap.close()
            self._heap = None

    def check(self) -> None:
        """Raise if a device-side wait of the transport gave up (mode "ipc": a dead or stalled peer)."""
        if self._heap is not None:
            self._heap.check()

    # ---- arrival-driven consumption (dit.WanDiT._sp_attention, csrc/attn7p.hip) ----------------------------------------------
    ARRIVAL_SLOTS = 64

    def enable_arrival(self, ops) -> None:
        """From now on the consumer does not ``wait(handle)``: it asks ``arrival(handle)`` for the flag each peer's rows of that
        row-chunk raise when they have landed and gates on them inside its own kernel (``ops.attention_pieces``), then calls
        ``consumed(handle)``.  Mode "ipc": the transport's own per-peer device words (and this rank's rows are no longer copied into
        the gathered buffer: the consumer reads them where they are).  The collective modes: ONE flag per row-chunk, written with
        ``ops.flag_write`` by a side stream that waited for the chunk's work handles."""
        self._arrival_ops = ops
        if self._heap is not None:
            self._heap.configure(copy_own_rows=False)
            self._flags = self._heap.arrival_flags()
        elif self.plan.world > 1 and torch.device(ops.device).type == "cuda":
            self._flags = ops.alloc((self.ARRIVAL_SLOTS,), torch.int32)
            self._flags.zero_()
            # HIGH priority = its own class of hardware queues: a stream of the default class can share a hardware queue with the compute stream
            # (GPU_MAX_HW_QUEUES, default 4; every 4th stream of torch's pool does), and then the attention launch that waits for this
            # stream's flag keeps the flag's writer from starting - a deterministic time-out (profiles/r06/stream_queue_share_probe.txt)
            self._flag_stream = torch.cuda.Stream(device=ops.device, priority=-1)

    def arrival(self, handle):
        """(flags, [(peer index j, flag index, value), ...]) for the peers' rows of this row-chunk (own rows excluded: in place);
        flags None = the rows are there already (CPU twins: the handles were waited for on the host)."""
        peers = [j for j in range(self.plan.world) if j != self.plan.rank]
        if not handle or not peers:
            return None, [(j, -1, 0) for j in peers]
        w = handle[0]
        if isinstance(w, _IpcWork):
            return self._flags, [(j, j, w.ticket + 1) for j in peers]
        if isinstance(w, _FlaggedWork):
            return self._flags, [(j, w.slot, w.value) for j in peers]
        for x in handle:           # no flag machinery (CPU): the host waits, the rows are simply there
            x.wait()
        return None, [(j, -1, 0) for j in peers]

    def consumed(self, handle) -> None:
        for w in handle:
            if isinstance(w, _IpcWork):
                w.consumed()

    def _flagged(self, works):
        """Collective modes under enable_arrival(): a side stream waits for the chunk and raises its flag."""
        if self._flag_stream is None:
            return tuple(works)
        slot = self._flag_seq % self.ARRIVAL_SLOTS
        self._flag_seq += 1
        value = self._flag_seq
        with torch.cuda.stream(self._flag_stream):
            for w in works:
                w.wait()
            self._arrival_ops.flag_write(self._flags, slot, value)
            flagged = torch.cuda.Event()
            flagged.record()
        return (_FlaggedWork(tuple(works), slot, value, flagged),)

    def start(self, rows: torch.Tensor, out: torch.Tensor):
        assert rows.is_contiguous() and out.is_contiguous() and out.shape[0] == self.plan.world * rows.shape[0]
        self.n_collectives += 1
        if self.mode == "native":
            return self._flagged((self._native.allgather(rows, out),))
        if self.mode == "ipc":
            return (self._heap.gather(rows, out),)
        if self.plan.world == 1:          # the one-rank rehearsal of the schedule (WanDiT.prepare(force_sp=True)): a local copy
            if self._arrival_ops is None:
                out.copy_(rows)
            return ()
        if self.mode == "allgather":
            return self._flagged((self.dist.all_gather_into_tensor(out, rows, group=self.group, async_op=True),))
        m, dist = rows.shape[0], self.dist
        if rows.is_cuda and dist.get_backend(self.group) != "nccl":
            # development set-ups only (gloo ranks sharing a GPU): gloo's send reads the device buffer from the host with no
            # regard for the stream that is still producing it - RCCL's send is stream-ordered, gloo's is not
            torch.cuda.current_stream(rows.device).synchronize()
        p2p = []
        for j, peer in enumerate(self.peers):
            if j == self.plan.rank:
                if self._arrival_ops is None:
                    out[j * m:(j + 1) * m].copy_(rows)      # own rows: a local copy on the compute stream
                continue
            p2p.append(dist.P2POp(dist.isend, rows, peer, self.group))
            p2p.append(dist.P2POp(dist.irecv, out[j * m:(j + 1) * m], peer, self.group))
        return self._flagged(dist.batch_isend_irecv(p2p))

    def wait(self, handle) -> None:
        if self.timing is not None and torch.cuda.is_available():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for w in handle:
                w.wait()
            e1.record()
            self.timing.append((e0, e1))
            return
        for w in handle:
            w.wait()   # nccl: the CURRENT STREAM waits (host does not block); gloo: host blocks


class _NativeComm:
    """libicvideo's own RCCL communicator for one sequence-parallel group + a side stream (KVGather mode "native").
    One per (process group, device) for the life of THAT PROCESS GROUP: a KVGather is rebuilt on every pipeline call, and
    ncclCommInitRank is a blocking rendezvous of the whole group that must not be repeated per call.  The communicators die
    with the torch.distributed world they were built in: whoever tears that world down (multigpu.WorkerPool.close, bench.py,
    a rank worker) calls ``close_all()`` first — a communicator kept past it would pair rank 0 with dead peers when a new
    pool rendezvouses under the same (None, peers, device) key.  Each entry keeps its group object alive, so ``id(group)``
    cannot be recycled by another group while the entry exists."""

    _cache = {}

    @classmethod
    def for_group(cls, dist, group, peers, rank: int, world: int) -> "_NativeComm":
        key = (id(group) if group is not None else None, tuple(peers), torch.cuda.current_device())
        comm = cls._cache.get(key)
        if comm is None:
            comm = cls._cache[key] = cls(dist, group, peers, rank, world)
        return comm

    @classmethod
    def close_all(cls) -> int:
        """Destroy every cached communicator (side streams drained first).  Call BEFORE destroy_process_group / before the
        peers of a worker pool exit.  Returns how many were closed."""
        n_heaps = _IpcHeap.close_all()           # the copy-engine transports die with the same world
        comms = list(cls._cache.values())
        for c in comms:
            c.close()
        cls._cache.clear()
        return len(comms) + n_heaps

    def __init__(self, dist, group, peers, rank: int, world: int):
        import ctypes
        from .. import native
        self.lib, self.native = native.lib(), native
        self.group = group          # strong reference: pins id(group) for the lifetime of the cache entry
        dev = torch.device("cuda", torch.cuda.current_device())
        idbuf = ctypes.create_string_buffer(native.COMM_ID_BYTES)
        if rank == 0:
            native.check(self.lib.icv_comm_unique_id(idbuf), "icv_comm_unique_id")
        t = torch.frombuffer(bytearray(idbuf.raw), dtype=torch.uint8).clone()
        if world > 1:
            # ship the id through the existing process group (nccl groups move device tensors, gloo host tensors)
            on_dev = dist.get_backend(group) == "nccl"
            t = t.to(dev) if on_dev else t
            dist.broadcast(t, src=peers[0], group=group)
        h = ctypes.c_void_p()
        native.check(self.lib.icv_comm_create(bytes(t.cpu().tolist()), rank, world, ctypes.byref(h)), "icv_comm_create")
        self.handle = h
        self.stream = torch.cuda.Stream(device=dev)
        self._dependents = []          # weak references to engines whose C driver context holds this handle (WanDiT._native)

    def add_dependent(self, engine) -> None:
        """An engine whose icv_dit context was given this communicator's raw handle and side stream (icv_dit_set_seqpar):
        ``close()`` destroys that context first, so a later native forward rebuilds it (or fails cleanly) instead of
        dereferencing a freed communicator."""
        import weakref
        self._dependents.append(weakref.ref(engine))

    def allgather(self, rows: torch.Tensor, out: torch.Tensor):
        ready = torch.cuda.Event()
        ready.record()                                   # the rows were produced on the compute stream
        self.stream.wait_event(ready)
        self.native.check(self.lib.icv_allgather_kv(self.handle, rows.data_ptr(), out.data_ptr(), rows.shape[0],
                                                    rows.shape[1] * rows.element_size(), self.stream.cuda_stream), "icv_allgather_kv")
        done = torch.cuda.Event()
        done.record(self.stream)
        return _EventWork(done)

    def close(self):
        """Drain the side stream, then destroy the communicator (queued transfers must not outlive it)."""
        h, self.handle = getattr(self, "handle", None), None
        for ref in getattr(self, "_dependents", []):
            eng = ref()
            if eng is not None and getattr(eng, "_native_comm", None) is self:
                eng._drop_native_context()
        self._dependents = []
        if h is not None:
            try:
                self.stream.synchronize()
                self.lib.icv_comm_destroy(h)
            except Exception:  # pragma: no cover - interpreter shutdown
                pass
        for k, v in list(self._cache.items()):
            if v is self:
                del self._cache[k]

    def __del__(self):
        self.close()


class _IpcHeap:
    """Symmetric heap + flag segment + pull streams of ONE KVGather in mode "ipc" (csrc/ipc.hip).

    Creation is a collective of the sequence-parallel group (the torch.distributed group carries the segment name and the
    72-byte heap handles, as Python objects: works on nccl and gloo groups alike) and ends with a pattern self-test: every rank
    fills the head of its heap with its own pattern, the chunk is exchanged exactly as in the loop, and a wrong byte anywhere
    fails the transport on EVERY rank (hipIpc between two devices / processes is refused on some hosts, and a transport that
    moves stale bytes must never win an autotune by being fast).  Any failure on any rank raises the same RuntimeError on
    every rank, so the start-up ladder drops the candidate symmetrically."""

    _live = []            # every open heap of this process (close_all at world teardown)
    _serial = [0]
    SELF_TEST_BYTES = 1 << 20

    def __init__(self, dist, group, peers, rank: int, world: int, nbytes: int, device):
        import ctypes
        import secrets
        from .. import native
        self.lib, self.native, self.dist, self.group = native.lib(), native, dist, group
        self.rank, self.world, self.handle, self.cursor = rank, world, None, 0
        self.device = torch.device(device)
        nbytes = (max(int(nbytes), self.SELF_TEST_BYTES) + 255) // 256 * 256
        self._serial[0] += 1
        name = f"/icv_kv_{os.getpid()}_{self._serial[0]}_{secrets.token_hex(4)}"
        if world > 1:
            box = [name]
            dist.broadcast_object_list(box, src=peers[0], group=group)
            name = box[0]
        self._shm_name = name
        err, blob = "", b""
        try:
            self._require_hw_queues(world)
            # torch owns the memory (borrowed by the library); a dedicated allocation, so the exported range is this heap
            self.mem = torch.empty((nbytes,), dtype=torch.uint8, device=self.device)
            h = ctypes.c_void_p()
            native.check(self.lib.icv_ipc_create(name.encode(), rank, world, self.mem.data_ptr(), nbytes, ctypes.byref(h)), "icv_ipc_create")
            self.handle = h
            buf = ctypes.create_string_buffer(native.IPC_HANDLE_BYTES)
            native.check(self.lib.icv_ipc_export(h, buf), "icv_ipc_export")
            blob = buf.raw
        except Exception as e:      # noqa: BLE001 - reported to every rank below
            err = f"rank {rank}: {type(e).__name__}: {e}"[:300]
        got = self._agree((err, blob))
        self._raise_if_any(got, "set-up")
        err = ""
        try:
            for p in range(world):
                if p != rank:
                    native.check(self.lib.icv_ipc_open_peer(self.handle, p, got[p][1]), f"icv_ipc_open_peer({p})")
        except Exception as e:      # noqa: BLE001
            err = f"rank {rank}: {type(e).__name__}: {e}"[:300]
        self._raise_if_any(self._agree((err, b"")), "opening the peers' heaps")
        self._unlink()                                      # everyone has it mapped: leave nothing behind in /dev/shm
        self._live.append(self)
        self._raise_if_any(self._agree((self._self_test(), b"")), "self-test")

    @staticmethod
    def _require_hw_queues(world: int) -> None:
        """One pull stream per peer + the launch stream + torch's / RCCL's own, and a pull that waits for its peer's flag is a spinning
        kernel that blocks its HARDWARE queue (profiles/r05/kv_contention.md): with the runtime's default of 4 queues the launch stream
        can sit behind a pending pull.  GPU_MAX_HW_QUEUES is read when HIP initialises, so it cannot be fixed from here: a process that
        was started without it does not get this transport - a local error like any other set-up failure, so the autotune /
        the launch ladder drop the candidate on every rank.  multigpu.WorkerPool and bench.py start their ranks with 16;
        ICV_IPC_ALLOW_FEW_QUEUES=1 overrides."""
        if world <= 1 or os.environ.get("ICV_IPC_ALLOW_FEW_QUEUES") == "1":
            return
        need, v = min(16, world + 2), os.environ.get("GPU_MAX_HW_QUEUES")
        try:
            have = int(v) if v is not None else 4          # the runtime's default
        except ValueError:
            have = 0
        if have < need:
            raise RuntimeError(f"GPU_MAX_HW_QUEUES={v if v is not None else 'unset (runtime default: 4)'}: the copy-engine transport wants >= {need} hardware "
                               f"queues for {world} ranks, and the variable is read when HIP initialises - export it before the process starts")

    def _agree(self, item):
        if self.world == 1:
            return [item]
        got = [None] * self.world
        self.dist.all_gather_object(got, item, group=self.group)
        return got

    def _unlink(self):
        """The flag segment's NAME goes as soon as it is no longer needed - also on every failure path (a set-up that
        failed between shm_open and the all-mapped point left /dev/shm/icv_kv_* behind).  Idempotent; every rank may call it."""
        name, self._shm_name = getattr(self, "_shm_name", None), None
        if name:
            self.lib.icv_ipc_shm_unlink(name.encode())

    def _raise_if_any(self, got, what):
        errs = [g[0] for g in got if g[0]]
        if errs:
            self._unlink()
            self.close()
            raise RuntimeError(f"copy-engine K|V transport unusable ({what}): " + "; ".join(errs))

    def configure(self, copy_own_rows: bool) -> None:
        self.native.check(self.lib.icv_ipc_configure(self.handle, int(bool(copy_own_rows))), "icv_ipc_configure")

    def arrival_flags(self):
        """Device words flags[p] = ticket + 1 once rank p's rows of that ticket have landed (icv_ipc_arrival)."""
        import ctypes
        p = ctypes.c_void_p()
        self.native.check(self.lib.icv_ipc_arrival(self.handle, ctypes.byref(p)), "icv_ipc_arrival")
        return _DevicePointer(p.value)

    COPY_KINDS = {0: "inconclusive (the device could not be filled)", 1: "copy engine (no wave needed)", 2: "blit kernel (needs CUs)"}

    def probe_copy(self, peer: int, nbytes: int = 8 << 20):
        """(kind text, milliseconds) of one pull of ``nbytes`` from ``peer`` while every wave slot of this device is held
        (icv_ipc_probe_copy): does the transport's data movement need compute units on THIS node?"""
        import ctypes
        kind, ms = ctypes.c_int(0), ctypes.c_double(-1.0)
        nbytes = min(int(nbytes), self.mem.numel())
        self.native.check(self.lib.icv_ipc_probe_copy(self.handle, int(peer), nbytes, ctypes.byref(kind), ctypes.byref(ms)), "icv_ipc_probe_copy")
        return self.COPY_KINDS.get(kind.value, str(kind.value)), ms.value

    def check(self) -> None:
        """Raises when one of this rank's device-side waits timed out (a dead or stalled peer): reads one host word."""
        if self.handle is not None:
            self.native.check(self.lib.icv_ipc_check(self.handle), "icv_ipc_check")

    def _self_test(self) -> str:
        try:
            n = self.SELF_TEST_BYTES
            for rep in range(2):        # twice: the second round reuses the rows (acquire) and the flag words
                self.acquire()
                if os.environ.get("ICV_TEST_HOOKS") == "1" and os.environ.get("ICV_IPC_INJECT") == f"selftest:{self.rank}":
                    raise RuntimeError("injected failure (test hook)")
                self.mem[:n] = (torch.arange(n, device=self.device, dtype=torch.int32) * (2 * self.rank + 3) + 17 * rep).to(torch.uint8)
                out = torch.zeros((self.world * n,), dtype=torch.uint8, device=self.device)
                self.gather(self.mem[:n], out).wait()
                torch.cuda.synchronize(self.device)
                for p in range(self.world):
                    want = (torch.arange(n, device=self.device, dtype=torch.int32) * (2 * p + 3) + 17 * rep).to(torch.uint8)
                    if not torch.equal(out[p * n:(p + 1) * n], want):
                        return f"rank {self.rank}: the rows pulled from rank {p} are wrong (round {rep})"
            return ""
        except Exception as e:      # noqa: BLE001
            self.abort()                # the peers may already be waiting for rows this rank will never publish
            return f"rank {self.rank}: {type(e).__name__}: {e}"[:300]

    def abort(self):
        """This rank cannot go on: release every peer wait that depends on it (icv_ipc_abort), so that the failure is an error
        on every rank instead of a spinning queue on the others.  The transport is unusable afterwards."""
        if self.handle is not None:
            self.lib.icv_ipc_abort(self.handle)

    def carve(self, rows: int, cols: int, dtype) -> torch.Tensor:
        nbytes = rows * cols * torch.empty((), dtype=dtype).element_size()
        start = (self.cursor + 255) // 256 * 256
        if start + nbytes > self.mem.numel():
            raise RuntimeError(f"symmetric K|V heap exhausted: {start + nbytes} > {self.mem.numel()} bytes (reserve() more)")
        self.cursor = start + nbytes
        return self.mem[start: start + nbytes].view(dtype).view(rows, cols)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def gather(self, rows: torch.Tensor, out: torch.Tensor):
        import ctypes
        off = rows.data_ptr() - self.mem.data_ptr()
        nbytes = rows.numel() * rows.element_size()
        if off < 0 or off + nbytes > self.mem.numel():
            raise ValueError("KVGather mode 'ipc': the rows to exchange are not inside the symmetric heap (use local_rows())")
        t = ctypes.c_int64()
        self.native.check(self.lib.icv_ipc_gather_start(self.handle, off, nbytes, out.data_ptr(), self._stream(), ctypes.byref(t)), "icv_ipc_gather_start")
        return _IpcWork(self, t.value)

    def acquire(self):
        self.native.check(self.lib.icv_ipc_acquire(self.handle, self._stream()), "icv_ipc_acquire")

    def close(self):
        h, self.handle = getattr(self, "handle", None), None
        if h is not None:
            try:
                self._unlink()
                # teardown must not depend on live peers: the library gives this rank's queues a bounded time to finish and then
                # satisfies every wait word itself (icv_ipc_drain); only then is a device-wide synchronize guaranteed to return
                self.lib.icv_ipc_drain(h, int(os.environ.get("ICV_IPC_DRAIN_TIMEOUT_MS", "5000")))
                torch.cuda.synchronize(self.device)
                self.lib.icv_ipc_destroy(h)
            except Exception:  # pragma: no cover - interpreter shutdown
                pass
        if self in self._live:
            self._live.remove(self)

    @classmethod
    def close_all(cls) -> int:
        heaps = list(cls._live)
        for h in heaps:
            h.close()
        return len(heaps)

    def __del__(self):
        self.close()


class _DevicePointer:
    """A raw device address with the one method the operator set asks of a tensor argument."""

    def __init__(self, address: int):
        self.address = int(address)

    def data_ptr(self) -> int:
        return self.address


class _IpcWork:
    def __init__(self, heap, ticket):
        self.heap, self.ticket = heap, ticket

    def wait(self):
        self.heap.native.check(self.heap.lib.icv_ipc_gather_wait(self.heap.handle, self.ticket, self.heap._stream()), "icv_ipc_gather_wait")

    def consumed(self):
        """The consumer gated on the arrival flags itself: bookkeeping only (no stream waits are enqueued)."""
        self.heap.native.check(self.heap.lib.icv_ipc_gather_consumed(self.heap.handle, self.ticket), "icv_ipc_gather_consumed")


class _FlaggedWork:
    """Work handles of one collective row-chunk whose completion a side stream turns into an arrival flag (KVGather.enable_arrival).
    The side stream has ALREADY waited for the works (a gloo send / recv work must not be waited for twice: the second wait blocks
    until its time-out): ``wait()`` makes the current stream wait for the side stream's flag write instead."""

    def __init__(self, works, slot: int, value: int, flagged: "torch.cuda.Event"):
        self.works, self.slot, self.value, self.flagged = works, slot, value, flagged

    def wait(self):
        torch.cuda.current_stream().wait_event(self.flagged)


class _EventWork:
    """wait() = the current stream waits for the transfer (like an async nccl work handle)."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


def chunk_bounds(n_rows: int, chunks: int, align: int = 1):
    """Row boundaries of the K/V gather chunks of one shard (identical on every rank).  The sizes RAMP
    (weights 1, 3, 6, 6, ...): only the first chunk's transfer is exposed before attention can start,
    so it is small; later chunks are large so launches stay efficient and each transfer hides under the
    previous chunk's attention.  ``align``: interior boundaries are multiples of it (the arrival-driven attention walks
    (peer, chunk) pieces tile by tile: a boundary off the 64-key tile grid costs every peer's piece a masked partial tile)."""
    chunks = max(1, min(chunks, max(1, n_rows // max(1, align)) if align > 1 else n_rows))
    w = [1.0, 3.0][:chunks] + [6.0] * max(0, chunks - 2)
    tot, acc, out = sum(w), 0.0, [0]
    for c in range(chunks):
        acc += w[c]
        if c == chunks - 1:
            out.append(n_rows)
            continue
        b = int(round(n_rows * acc / tot))
        if align > 1:
            b = max(align, int(round(b / align)) * align)
        out.append(min(n_rows - 1, max(out[-1] + max(1, align), b)) if align > 1 else max(out[-1] + 1, b))
    if align > 1:          # the clamps above may have produced a non-increasing tail on tiny shards: fall back to fewer chunks
        out = sorted(set(out))
        if out[-1] != n_rows:
            out.append(n_rows)
    return out


def gather_latent(latent: torch.Tensor, plan: ShardPlan, grid, group=None) -> torch.Tensor:
    """After the loop each rank has stepped only its own tokens' latent patches; exchange them so
    every rank holds the full latent (once per generation, 4.8 MB at 480p)."""
    if plan.world == 1:
        return latent
    import torch.distributed as dist
    C, T, H8, W8 = latent.shape
    Hp, Wp = H8 // 2, W8 // 2
    # view latent as tokens: [T, Hp, Wp, C, 2, 2] -> [S, C*4]
    tok = latent.reshape(C, T, Hp, 2, Wp, 2).permute(1, 2, 4, 0, 3, 5).reshape(T * Hp * Wp, C * 4).contiguous()
    mine = tok[plan.tok0: plan.tok0 + plan.n_tok].contiguous()
    full = torch.empty_like(tok)
    dist.all_gather_into_tensor(full, mine, group=group)
    return full.reshape(T, Hp, Wp, C, 2, 2).permute(3, 0, 1, 4, 2, 5).reshape(C, T, H8, W8).contiguous()


AUTOTUNE_WAIT_DEADLINE_US = 5_000_000      # in-kernel deadline of the arrival-driven attention while a candidate is being tried


def autotune_kv_exchange(model, run_layers, sync, candidates=None, reps: int = 2, reduce_max=None, log=None, exchange_only=None, probe=None,
                         on_candidate=None):
    """Start-up choice of the K|V exchange (transport x chunk count) by MEASUREMENT on the ranks that will run it.

    xGMI is point-to-point and what RCCL schedules over it is not known before the first contact: ``allgather`` may ring
    (world-1 hops), ``p2p`` uses every link once, ``native`` is the same collective on libicvideo's own communicator, ``ipc``
    moves the rows with the copy engines and leaves every CU to the attention it overlaps with; more chunks hide more of the
    transfer but launch shorter attention kernels.  Each candidate runs ``run_layers()`` (a couple of real DiT layers on this
    rank's shard: exchange + chunked attention exactly as in the loop) once to warm up and ``reps`` times under a timer; the time
    a layer takes IS the compute plus whatever part of the transfer stayed exposed.  Every rank must end with the same choice:
    times are max-reduced over the ranks (``reduce_max(list) -> list``) and ties break by candidate order.

    Failure handling, in two phases.  SET-UP (``model.set_kv_exchange``: communicator / heap creation, handle exchange,
    self-test) may fail on any rank: its outcome is max-reduced BEFORE anybody runs the candidate's layers, so a transport
    that cannot be set up on one rank is dropped on every rank and nobody is left inside one of its collectives.  RUNNING a
    candidate that every rank set up is different: a rank that raises in the middle of ``run_layers()`` leaves its peers
    blocked in that candidate's exchange, and no reduce can be reached from there - such an error is FATAL for this attempt and
    propagates (the launch ladder above, multigpu / launch_guard, abandons the process group and falls back to its next plan).
    A candidate whose warm-up ends with a device-side wait that GAVE UP (``model.exchange_gave_up()``: the arrival-driven attention's
    in-kernel deadline, shortened to AUTOTUNE_WAIT_DEADLINE_US here, or the copy-engine transport's) is dropped on every rank too.
    ``exchange_only()`` (optional): one layer's exchange with nothing to hide under; its time is recorded next to the layer time
    (``exchange_ms``: the raw transfer, for the bandwidth it implies), never used for the choice.
    ``probe(mode, chunks) -> dict`` (optional): further per-candidate measurements merged into the candidate's row (bench.py: the
    self-attention under the real exchange next to the same launches served from memory; for "ipc" whether a pull needs CUs); it
    runs on every rank at the same point, after the timed layers, and may use collectives symmetrically; never used for the choice.
    ``on_candidate(mode, chunks)`` (optional) is called before each candidate (bench.py reports a supervisor phase per candidate: each
    gets its own time budget, and a hang is blamed on the candidate that caused it).
    Returns (best (mode, chunks), table of dict rows)."""
    import time
    cands = list(candidates or [(m + sfx, c) for m in ("allgather", "p2p", "native", "ipc") for sfx, c in (("+arrival", 4), ("", 4), ("", 2))])
    table = []
    for mode, chunks in cands:
        if on_candidate is not None:
            on_candidate(mode, chunks)
        err = ""
        try:
            model.set_kv_exchange(mode, chunks)
        except Exception as e:      # noqa: BLE001 - a transport that cannot be set up here is simply not a candidate
            err = f"{type(e).__name__}: {e}"[:300]
        bad = 1.0 if err else 0.0
        if reduce_max is not None:
            bad = reduce_max([bad])[0]
        if bad:
            table.append(dict(kv_exchange=mode, sp_chunks=chunks, ms=None, error=err or "set-up failed on another rank", exchange_ms=None))
            if log:
                log(f"autotune {mode:9s} chunks {chunks}: unusable ({err or 'set-up failed on another rank'})")
            continue
        # warm-up under a SHORT in-kernel deadline: a candidate whose data movement cannot make progress beside the waiting work-groups of
        # the arrival-driven attention (an RCCL channel kernel that finds no room on CUs full of spinning attention work-groups: a hazard
        # no one-GPU rehearsal can show - a 1-rank ncclAllGather is a plain device copy) must cost seconds, not the product's deadline
        long_deadline = getattr(model, "sp_timeout_us", None)
        if long_deadline is not None:
            model.sp_timeout_us = min(long_deadline, AUTOTUNE_WAIT_DEADLINE_US)
        run_layers()
        sync()
        if long_deadline is not None:
            model.sp_timeout_us = long_deadline
        gave_up = model.exchange_gave_up() if hasattr(model, "exchange_gave_up") else None
        bad = 1.0 if gave_up else 0.0
        if reduce_max is not None:
            bad = reduce_max([bad])[0]
        if bad:
            why = gave_up or "a device-side wait gave up on another rank"
            table.append(dict(kv_exchange=mode, sp_chunks=chunks, ms=None, error=("starved or stalled: " + why)[:300], exchange_ms=None))
            if log:
                log(f"autotune {mode:17s} chunks {chunks}: dropped ({why[:200]})")
            continue
        t0 = time.perf_counter()
        for _ in range(reps):
            run_layers()
        sync()
        ms, xms = 1e3 * (time.perf_counter() - t0) / reps, 0.0
        if exchange_only is not None:
            exchange_only()
            sync()
            t0 = time.perf_counter()
            for _ in range(3):
                exchange_only()
            sync()
            xms = 1e3 * (time.perf_counter() - t0) / 3
        if reduce_max is not None:
            ms, xms = reduce_max([ms, xms])
        row = dict(kv_exchange=mode, sp_chunks=chunks, ms=ms, error=None, exchange_ms=xms or None)
        if probe is not None:
            row.update(probe(mode, chunks) or {})
        table.append(row)
        if log:
            log(f"autotune {mode:17s} chunks {chunks}: {ms:.2f} ms" + (f" (exchange alone {xms:.2f} ms)" if xms else "")
                + "".join(f"; {k} {v:.3f}" if isinstance(v, float) else f"; {k} {v}" for k, v in row.items()
                          if k not in ("kv_exchange", "sp_chunks", "ms", "error", "exchange_ms")))
    usable = [(r["ms"], i) for i, r in enumerate(table) if r["ms"] is not None]
    if not usable:
        raise RuntimeError(f"K|V exchange autotune: no transport worked: {table}")
    best = cands[min(usable)[1]]
    model.set_kv_exchange(*best)
    return best, table
