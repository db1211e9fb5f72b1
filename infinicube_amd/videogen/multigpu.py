"""Multi-GPU generation behind the UNCHANGED single-process caller.

InfiniCube's stage 2 builds ONE ``WanVideoGenerator`` in its own process with the literal ``device="cuda:0"`` and never
touches ``torch.distributed`` [R infinicube/inference/guidance_buffer_generation.py:755-782]; launching that script N
times under ``torch.distributed.run`` would render the voxel world N times and is a change of the caller's contract.
So the N-GPU mode lives behind the constructor: with ``ICV_WORLD=N`` (or ``auto``) in the environment the caller's
process becomes rank 0 on GPU 0 and ``WorkerPool`` starts N-1 persistent worker processes (one per further GPU, this
module's ``__main__``), each of which builds the same generator on ``cuda:<rank>``, joins one process group
(``nccl`` = RCCL over xGMI; the K/V exchange and the velocity swap of seqpar.py run in it) and then serves
``generate`` commands.  Per call rank 0 broadcasts the request (prompt, seed, sampling settings and the two uint8
buffers, over a gloo control group on the host), every rank runs ``WanVideoPipeline.__call__`` on its token shard /
CFG branch, and only rank 0 decodes the latent, returns the frames and writes the mp4.

Not a scheduler or a serving layer: one generator, one request at a time, exactly the reference's usage.
"""
from __future__ import annotations

import atexit
import datetime
import importlib
import os
import pickle
import socket
import subprocess
import sys
import tempfile
import time
from typing import Optional

import numpy as np
import torch

# pipeline attributes rank 0 may have changed since construction; sent with every request so the ranks cannot drift
_PIPE_SETTINGS = ("num_inference_steps", "cfg_scale", "sigma_shift", "parallelism", "sp_chunks", "kv_exchange",
                  "reference_rounding", "gemm_dtype", "attn_dtype")


_ACTIVE_POOL = None   # the one WorkerPool of this process (the reference's caller keeps ONE generator per process)


def pool_for(world: int, ctor_kwargs: dict) -> "WorkerPool":
    """The process-wide pool: created on first use, reused by a later generator built with the same arguments; a
    generator with DIFFERENT arguments cannot share the workers (they hold the first one's weights) and is refused."""
    global _ACTIVE_POOL
    if _ACTIVE_POOL is not None and not _ACTIVE_POOL._closed:
        if _ACTIVE_POOL.ctor_kwargs == ctor_kwargs and _ACTIVE_POOL.world == world:
            return _ACTIVE_POOL
        raise RuntimeError("ICV_WORLD: this process already drives a worker pool built for another WanVideoGenerator "
                           f"({_ACTIVE_POOL.ctor_kwargs}); close it (generator._pool.close()) before building a different one")
    pool = WorkerPool(world, ctor_kwargs)
    if pool.world == 1:          # every multi-GPU plan failed its start-up probe: the caller's generator runs on ONE GPU
        _ACTIVE_POOL = None
        return None
    _ACTIVE_POOL = pool
    return _ACTIVE_POOL


_DEGRADED = None      # set to a reason string when a multi-GPU start was abandoned for ONE GPU with the process group possibly wedged


def degraded() -> bool:
    """True after a failed multi-GPU start whose process group could not be torn down cleanly: the pipeline then treats
    this process as a single rank whatever torch.distributed says."""
    return _DEGRADED is not None


def layout_cache():
    """The live pool's parallel-layout cache (process groups are a resource of the pool's process group; the start-up probe
    already built the plan's layout in it - on rank 0 AND in the workers), or None."""
    if _WORKER_LAYOUTS is not None:
        return _WORKER_LAYOUTS
    return _ACTIVE_POOL.layouts if (_ACTIVE_POOL is not None and not _ACTIVE_POOL._closed) else None


def requested_world() -> int:
    """ICV_WORLD = N | auto (= every visible GPU).  1 inside a worker, when unset, or when the process already is a
    rank of somebody else's job (torch.distributed initialised / launched by torch.distributed.run)."""
    v = os.environ.get("ICV_WORLD", "").strip().lower()
    if not v or os.environ.get("ICV_WORKER_RANK") is not None or degraded():
        return 1
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (_ACTIVE_POOL is None or _ACTIVE_POOL._closed):
        return 1          # somebody else's process group (e.g. torch.distributed.run): this process is already one rank of it
    if v == "auto":
        return max(1, torch.cuda.device_count())
    n = int(v)
    if n < 1:
        raise ValueError(f"ICV_WORLD must be a positive integer or 'auto', got {v!r}")
    return n


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _resolve(spec: str):
    mod, _, fn = spec.partition(":")
    return getattr(importlib.import_module(mod), fn)


class WorkerPool:
    """Rank 0's handle on the N-1 worker processes and the process group they share."""

    def __init__(self, world: int, ctor_kwargs: dict, backend: Optional[str] = None):
        import torch.distributed as dist
        self.world, self.dist, self.ctor_kwargs = world, dist, dict(ctor_kwargs)
        self._closed = True
        self._ready = False          # set by the first wait_ready(): the workers then sit in their serve loop
        self.layouts = {}            # seqpar.ParallelLayout per (world, rank, mode, cfg): shared by every pipeline of this process
        self.backend = backend or os.environ.get("ICV_DIST_BACKEND", "nccl")
        self.timeout_s = float(os.environ.get("ICV_WORLD_TIMEOUT_S", "3600"))
        self.probe_timeout_s = float(os.environ.get("ICV_WORLD_PROBE_TIMEOUT_S", "180"))
        self.plan, self.failed_plans, self.procs = None, [], []
        # Staged start (the same ladder as bench.py's launch_guard): the requested layout / K|V transport, then the same layout
        # with RCCL's plain all-gather, then `sp` on the world group (no sub-groups at all), then ONE GPU.  Each plan is PROBED
        # before any weights are loaded: every rank builds the plan's process groups and runs one small collective on each
        # plus one K|V exchange with the plan's transport, under a deadline; a raised error or a hung rank abandons that
        # process group (workers killed, group destroyed) and the next plan starts from fresh processes.
        # K|V transport: what the caller asked for, else "auto" on RCCL ranks (the pipeline's start-up autotune measures the
        # transports on its first sequence-parallel call; the probe below checks the plain all-gather for it), "allgather" elsewhere
        req = (os.environ.get("ICV_PARALLELISM", "auto"), os.environ.get("ICV_KV_EXCHANGE") or ("auto" if self.backend == "nccl" else "allgather"))
        plans = [req]
        if os.environ.get("ICV_WORLD_FALLBACK", "1") == "1":
            plans += [(req[0], "allgather"), ("sp", "allgather")]
        seen = set()
        plans = [p for p in plans if not (self._resolved(p) in seen or seen.add(self._resolved(p)))]
        for k, plan in enumerate(plans):
            err = self._start(k, self._resolved(plan))
            if err is None:
                self.plan = self._resolved(plan)
                break
            self.failed_plans.append(dict(plan=list(self._resolved(plan)), error=err))
            print(f"[icvideo] multi-GPU start with plan {self._resolved(plan)} failed: {err[:500]}", file=sys.stderr)
            self._abandon_group()
            if degraded():
                break
        if self.plan is None:
            print(f"[icvideo] no multi-GPU plan could be started ({len(self.failed_plans)} tried); continuing on ONE GPU", file=sys.stderr)
            self.world = 1
            for kk in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                os.environ.pop(kk, None)
            return
        self._closed = False
        atexit.register(self.close)
        print(f"[icvideo] {world} ranks: this process + {world - 1} workers (backend {self.backend}, layout {self.plan[0]}, K|V exchange "
              f"{self.plan[1]}; logs in {self._dir})", file=sys.stderr)

    def _resolved(self, plan):
        mode, kv = plan
        if mode == "auto":
            mode = "cfg+sp" if self.world % 2 == 0 else "sp"
        return (mode, kv)

    def _start(self, k: int, plan) -> Optional[str]:
        """Spawn the workers for ``plan``, join the process group, probe the plan on every rank.  None = all ranks ok."""
        dist, world = self.dist, self.world
        port = _free_port()
        init_method = f"tcp://127.0.0.1:{port}"
        self._dir = tempfile.mkdtemp(prefix="icv_world_")
        spec = dict(ctor=self.ctor_kwargs, backend=self.backend, init_method=init_method, timeout_s=self.timeout_s,
                    factory=os.environ.get("ICV_WORKER_FACTORY"), plan=plan, plan_index=k, probe_timeout_s=self.probe_timeout_s)
        spec_path = os.path.join(self._dir, "spec.pkl")
        with open(spec_path, "wb") as f:
            pickle.dump(spec, f)
        self.procs = []
        from .seqpar import apply_rccl_channel_cap
        apply_rccl_channel_cap(int(os.environ.get("ICV_RCCL_MAX_CHANNELS", "-1")))      # this process (rank 0) and, inherited, the workers
        # one pull stream per peer (copy-engine K|V transport) + the launch stream + torch's / RCCL's own: a pending pull blocks its
        # hardware queue (profiles/r05/kv_contention.md, "pending waits"), so nothing else may share it; read at HIP initialisation
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        for r in range(1, world):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), ICV_WORKER_RANK=str(r),
                       ICV_WORKER_SPEC=spec_path, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            env.pop("ICV_WORLD", None)
            log = open(os.path.join(self._dir, f"worker{r}.log"), "wb")
            self.procs.append((subprocess.Popen([sys.executable, "-m", "infinicube_amd.videogen.multigpu"], env=env,
                                                stdout=log, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL), log))
        # this process is rank 0 on GPU 0 ("cuda:0" already means that; LOCAL_RANK makes it explicit for resolve_device)
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        try:
            # the data-path group's timeout bounds its rendezvous and every RCCL collective (all short); the gloo control group
            # keeps the long one: workers park in its broadcast between requests
            dist.init_process_group(self.backend, init_method=init_method, rank=0, world_size=world,
                                    timeout=datetime.timedelta(seconds=min(self.timeout_s, float(os.environ.get("ICV_WORLD_INIT_TIMEOUT_S", "600")))))
            self.ctrl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=self.timeout_s))
            mine = probe_plan(plan, k, 0, world, self.backend, self.layouts, self.probe_timeout_s)
            results = [None] * world
            work = _gather_with_liveness(dist, results, mine, self.ctrl, self._check_alive, self.probe_timeout_s + 60.0)
            if work is not None:
                return work
        except Exception as e:  # noqa: BLE001
            return f"{type(e).__name__}: {e}"
        bad = [f"rank {r}: {m}" for r, m in enumerate(results) if m != "ok"]
        return "; ".join(bad) if bad else None

    def _abandon_group(self):
        """After a failed plan: workers gone (exactly the PIDs started here), this rank's process group destroyed - under a
        deadline, because a wedged communicator can block that too; if it does, the process is marked degraded."""
        global _DEGRADED
        self._kill()
        self.layouts.clear()
        import threading
        done = []

        def destroy():
            try:
                from .seqpar import _NativeComm
                _NativeComm.close_all()
                if self.dist.is_initialized():
                    self.dist.destroy_process_group()
            except Exception as e:  # noqa: BLE001
                done.append(f"{type(e).__name__}: {e}")
                return
            done.append(None)

        t = threading.Thread(target=destroy, daemon=True)
        t.start()
        t.join(timeout=30.0)
        if not done or done[0] is not None:
            _DEGRADED = f"process group of a failed multi-GPU start could not be destroyed ({done[0] if done else 'timed out'})"
            print(f"[icvideo] {_DEGRADED}", file=sys.stderr)

    # -- rank 0 side -------------------------------------------------------------------------------
    def _check_alive(self):
        for i, (p, _) in enumerate(self.procs):
            rc = p.poll()
            if rc is not None:
                tail = ""
                try:
                    with open(os.path.join(self._dir, f"worker{i + 1}.log"), "rb") as f:
                        tail = f.read()[-2000:].decode(errors="replace")
                except OSError:
                    pass
                raise RuntimeError(f"multi-GPU worker rank {i + 1} exited with code {rc}:\n{tail}")

    def _blame_dead_worker(self, grace_s: float = 10.0):
        """A transport error usually means a worker is going down: give it a moment to exit, then report ITS log."""
        deadline = time.time() + grace_s
        while time.time() < deadline:
            self._check_alive()
            time.sleep(0.2)

    def wait_ready(self):
        """Every rank has built its generator (weights resident).  A second generator built with the same arguments
        reuses the pool (pool_for): its workers are already parked in their serve loop, where a barrier would pair with a
        broadcast and hang, so only the first call synchronises."""
        self._check_alive()
        if self._ready:
            return
        try:
            work = self.dist.barrier(group=self.ctrl, async_op=True)
            while not work.is_completed():      # a worker that dies while loading must not cost the whole collective timeout
                self._check_alive()
                time.sleep(0.2)
            work.wait()
            self._ready = True
        except RuntimeError as e:
            if "multi-GPU worker rank" not in str(e):
                self._blame_dead_worker()       # prefer the dead worker's own log to the transport's error
            raise

    def generate(self, semantic: np.ndarray, coordinate: np.ndarray, call_kwargs: dict, pipe) -> None:
        """Hand one request to the workers; the caller then runs its own share through ``pipe(...)``."""
        self._check_alive()
        if call_kwargs.get("seed") is None:
            raise ValueError("WorkerPool.generate: the caller resolves seed=None to one drawn integer for all ranks (WanVideoGenerator.generate does)")
        msg = dict(cmd="generate", shape=tuple(semantic.shape), call=call_kwargs,
                   settings={k: getattr(pipe, k) for k in _PIPE_SETTINGS if hasattr(pipe, k)})
        try:
            self.dist.broadcast_object_list([msg], src=0, group=self.ctrl)
            for arr in (semantic, coordinate):
                self.dist.broadcast(torch.from_numpy(np.ascontiguousarray(arr)), src=0, group=self.ctrl)
        except Exception:
            self._blame_dead_worker()
            raise

    def close(self):
        if getattr(self, "_closed", True):
            return
        self._closed = True
        try:
            # libicvideo's own communicators (seqpar._NativeComm) were built on this pool's process groups and go first, while
            # the peers are still alive (each worker closes its own on "exit"): a communicator that survived the pool would
            # pair this rank with dead peers when the next pool rendezvouses under the same cache key
            from .seqpar import _NativeComm
            _NativeComm.close_all()
        except Exception:
            pass
        try:
            if all(p.poll() is None for p, _ in self.procs):
                self.dist.broadcast_object_list([dict(cmd="exit")], src=0, group=self.ctrl)
            deadline = time.time() + 30
            for p, _ in self.procs:
                try:
                    p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    pass
        finally:
            self._kill()
            self.layouts.clear()     # the sub-groups go with the process group
            try:
                if self.dist.is_initialized():
                    self.dist.destroy_process_group()
            except Exception:
                pass
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                os.environ.pop(k, None)

    def _kill(self):
        for p, log in self.procs:
            if p.poll() is None:
                p.kill()          # exactly the PIDs this pool started
                p.wait()
            log.close()


# -- both sides: the start-up probe of one plan ------------------------------------------------------------------------
_WORKER_LAYOUTS = None     # in a worker: the layouts its probe built (the pipeline reuses them: creating groups is collective)


def probe_plan(plan, plan_index: int, rank: int, world: int, backend: str, layouts: dict, timeout_s: float) -> str:
    """Build ``plan``'s process groups and push one small collective through each of them and one K|V exchange through the plan's
    transport, in a helper thread under a deadline.  "ok", or what went wrong ("hung ..." when the deadline passed: the thread
    is then left behind, wedged inside its collective).  ICV_TEST_POOL_INJECT=<plan_index>:<rank>:raise|hang injects a failure."""
    import threading
    mode, kv = plan
    box = []

    def body():
        try:
            import torch.distributed as dist
            from .seqpar import KVGather, ParallelLayout
            # failure injection for the fall-back tests: honoured only when the test harness ALSO sets ICV_TEST_HOOKS=1
            inject = os.environ.get("ICV_TEST_POOL_INJECT", "") if os.environ.get("ICV_TEST_HOOKS") == "1" else ""
            for item in filter(None, inject.split(",")):
                a, r, kind = item.split(":")
                if (int(a), int(r)) == (plan_index, rank):
                    if kind == "raise":
                        raise RuntimeError(f"injected failure in the probe of plan {plan_index} on rank {rank}")
                    time.sleep(10 ** 6)
            dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))) if backend == "nccl" else torch.device("cpu")
            if dev.type == "cuda":
                torch.cuda.set_device(dev)
            lay = ParallelLayout.make(world, rank, mode, use_cfg=True)
            layouts[(world, rank, mode, True)] = lay
            for g in (None, lay.sp_group, lay.pair_group):
                if g is None and lay.mode == "cfg+sp" and world == 1:
                    continue
                m = dist.get_world_size(g)
                src = torch.full((4,), float(rank), device=dev)
                dst = torch.empty((4 * m,), device=dev)
                dist.all_gather_into_tensor(dst, src, group=g)
                want = [float(r) for r in (dist.get_process_group_ranks(g) if g is not None else range(world))]
                if dst.view(m, 4)[:, 0].tolist() != want:
                    raise RuntimeError(f"group smoke test returned {dst.view(m, 4)[:, 0].tolist()}, expected {want}")
            if lay.sp_world > 1:       # the K|V transport of the plan, on 8 rows per rank
                plan_s = lay.shard_plan(8 * lay.sp_world)
                kg = KVGather(plan_s, lay.sp_group, "allgather" if kv == "auto" else kv)
                kg.reserve(1 << 16, dev)                 # the copy-engine transport keeps the rows in its symmetric heap
                rows = kg.local_rows(8, 16, torch.bfloat16, lambda shape, dt: torch.empty(shape, dtype=dt, device=dev))
                rows.fill_(float(lay.sp_rank + 1))
                out = torch.zeros((8 * lay.sp_world, 16), device=dev, dtype=torch.bfloat16)
                kg.wait(kg.start(rows, out))
                if dev.type == "cuda":
                    torch.cuda.synchronize(dev)
                got = out.view(lay.sp_world, 8, 16)[:, 0, 0].float().tolist()
                kg.close()
                if got != [float(r + 1) for r in range(lay.sp_world)]:
                    raise RuntimeError(f"K|V exchange '{kv}' returned shard order {got}")
            box.append("ok")
        except BaseException as e:  # noqa: BLE001
            box.append(f"{type(e).__name__}: {e}")

    t = threading.Thread(target=body, daemon=True)
    t.start()
    t.join(timeout=timeout_s)
    return box[0] if box else f"hung in the probe of plan {plan} (no answer within {timeout_s:.0f} s)"


def _gather_with_liveness(dist, results, mine, group, check_alive, deadline_s: float) -> Optional[str]:
    """all_gather_object on the control group, in a thread so that a worker that died (or never reports) cannot block rank 0
    for the whole collective timeout.  None = gathered; otherwise what went wrong."""
    import threading
    err = []

    def body():
        try:
            dist.all_gather_object(results, mine, group=group)
        except Exception as e:  # noqa: BLE001
            err.append(f"{type(e).__name__}: {e}")

    t = threading.Thread(target=body, daemon=True)
    t.start()
    t_end = time.time() + deadline_s
    while t.is_alive() and time.time() < t_end:
        try:
            check_alive()
        except RuntimeError as e:
            return str(e)
        t.join(timeout=0.2)
    if t.is_alive():
        return f"not every rank reported its probe within {deadline_s:.0f} s"
    return err[0] if err else None


# -- worker side ---------------------------------------------------------------------------------------
def _exit_when_parent_dies(parent_pid: int) -> None:
    """A worker blocked in a collective would outlive a crashed caller and keep its GPU: poll the parent and leave."""
    import threading

    def watch():
        while True:
            time.sleep(5.0)
            if os.getppid() != parent_pid:
                print("[worker] parent process is gone: exiting", flush=True)
                os._exit(3)

    threading.Thread(target=watch, daemon=True).start()


def worker_main() -> int:
    import torch.distributed as dist
    _exit_when_parent_dies(os.getppid())
    with open(os.environ["ICV_WORKER_SPEC"], "rb") as f:
        spec = pickle.load(f)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    to = datetime.timedelta(seconds=spec["timeout_s"])
    dist.init_process_group(spec["backend"], init_method=spec["init_method"], rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=min(spec["timeout_s"], float(os.environ.get("ICV_WORLD_INIT_TIMEOUT_S", "600")))))
    ctrl = dist.new_group(backend="gloo", timeout=to)
    # the plan's start-up probe, before any weights are loaded (a failed plan is abandoned cheaply): every rank reports
    layouts = {}
    mine = probe_plan(spec["plan"], spec.get("plan_index", 0), rank, world, spec["backend"], layouts, spec.get("probe_timeout_s", 180.0))
    results = [None] * world
    dist.all_gather_object(results, mine, group=ctrl)
    if any(m != "ok" for m in results):
        print(f"[worker {rank}] plan {spec['plan']} failed its probe: {results}", flush=True)
        os._exit(4)               # no clean teardown: the group may be wedged; rank 0 starts the next plan from fresh processes
    # this file runs as __main__ in a worker; the pipeline imports the package module: hand the layouts to THAT module object
    import infinicube_amd.videogen.multigpu as canonical
    canonical._WORKER_LAYOUTS = layouts
    from .inference import WanVideoGenerator
    factory = _resolve(spec["factory"]) if spec.get("factory") else None
    gen = WanVideoGenerator(**spec["ctor"], pipeline_factory=factory)   # device "cuda:0" resolves to cuda:LOCAL_RANK
    gen.pipe.parallelism, gen.pipe.kv_exchange = spec["plan"]           # the plan that passed its probe (rank 0 re-sends both per request)
    dist.barrier(group=ctrl)
    print(f"[worker {rank}] ready", flush=True)
    while True:
        box = [None]
        dist.broadcast_object_list(box, src=0, group=ctrl)
        msg = box[0]
        if msg["cmd"] == "exit":
            break
        bufs = []
        for _ in range(2):
            t = torch.empty(msg["shape"], dtype=torch.uint8)
            dist.broadcast(t, src=0, group=ctrl)
            bufs.append(t.numpy())
        for k, v in msg["settings"].items():
            setattr(gen.pipe, k, v)
        n, h, w, _ = msg["shape"]
        c = msg["call"]
        gen.pipe(prompt=c["prompt"], negative_prompt=c["negative_prompt"], semantic_buffer_video=gen._ndarray_to_pil_list(bufs[0]),
                 coordinate_buffer_video=gen._ndarray_to_pil_list(bufs[1]), height=h, width=w, num_frames=n, seed=c["seed"],
                 tiled=c["tiled"], return_latents=True, join_decode=True)   # rank 0 alone blends the decoded tiles (this rank
        #                                                             computes its share of them), returns frames, writes the mp4
        print(f"[worker {rank}] request done", flush=True)
    from .seqpar import _NativeComm
    _NativeComm.close_all()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(worker_main())
