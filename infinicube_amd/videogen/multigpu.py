"""Multi-GPU generation behind the UNCHANGED single-process caller.

InfiniCube's stage 2 builds ONE ``WanVideoGenerator`` in its own process with the literal ``device="cuda:0"`` and never
touches ``torch.distributed`` [R infinicube/inference/guidance_buffer_generation.py:755-782]; launching that script N
times under ``torch.distributed.run`` would render the voxel world N times and is a change of the caller's contract.
So the N-GPU mode lives behind the constructor: with ``ICV_WORLD=N`` (or ``auto``) in the environment the caller's
process becomes a CLIENT and ``WorkerPool`` starts **N fresh worker processes, rank 0 included** (this module's
``__main__``), each of which builds the same generator on ``cuda:<rank>``, joins one process group (``nccl`` = RCCL over
xGMI; the K/V exchange and the velocity swap of seqpar.py run in it) and then serves ``generate`` commands.

Why rank 0 is not the caller's process (round 6): by the time the reference's caller builds the generator it has long
initialised the GPU [R infinicube/inference/guidance_buffer_generation.py:459-460,626 ``.cuda()`` the voxel world and
render the buffers; the generator is constructed at :755-766].  ``GPU_MAX_HW_QUEUES`` and ``HSA_ENABLE_IPC_MODE_LEGACY``
are read when HIP / HSA initialise, so setting them from the constructor would reach the N-1 other ranks and not rank 0 -
the rank every step's max-reduce waits for - and the copy-engine K|V transport's launch stream could then sit behind its
own spinning pull-waits (profiles/r05/kv_contention.md, "a pending wait blocks its HARDWARE queue").  A rank's
environment is therefore composed for its ``subprocess.Popen`` and nowhere else: the caller's ``os.environ`` is never
written, the caller never joins a process group (a ``torch.distributed`` job of its own stays possible), and rank 0's
streams do not share a process with the caller's fvdb allocations.

Per call the client writes the two uint8 buffers ONCE into a tmpfs file every worker maps (no broadcast), sends the
request (prompt, seed, sampling settings) down one AF_UNIX connection per worker, every rank runs
``WanVideoPipeline.__call__`` on its token shard / CFG branch, rank 0 blends the decoded tiles and writes the uint8
frames into a second tmpfs file, and the client turns them into PIL images and writes the mp4.

Not a scheduler or a serving layer: one generator, one request at a time, exactly the reference's usage.
"""
from __future__ import annotations

import atexit
import contextlib
import datetime
import importlib
import io
import multiprocessing.connection as mpc
import os
import pickle
import shutil
import socket
import subprocess
import sys
import tempfile
import time
import traceback
from typing import List, Optional

import numpy as np
import torch

# pipeline attributes the caller may have changed since construction; sent with every request so the ranks cannot drift
_PIPE_SETTINGS = ("num_inference_steps", "cfg_scale", "sigma_shift", "parallelism", "sp_chunks", "kv_exchange",
                  "reference_rounding", "gemm_dtype", "attn_dtype")

# what a rank reports about the environment its HIP runtime initialised under (WorkerPool.plan_record)
_RANK_ENV_KEYS = ("GPU_MAX_HW_QUEUES", "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_MAX_NCHANNELS")

_ACTIVE_POOL = None   # the one WorkerPool of this process (the reference's caller keeps ONE generator per process)


def pool_for(world: int, ctor_kwargs: dict) -> Optional["WorkerPool"]:
    """The process-wide pool: created on first use, reused by a later generator built with the same arguments; a
    generator with DIFFERENT arguments cannot share the workers (they hold the first one's weights) and is refused.
    None = every multi-GPU plan failed its start-up probe: the caller's generator runs on ONE GPU, in its own process."""
    global _ACTIVE_POOL
    if _ACTIVE_POOL is not None and not _ACTIVE_POOL._closed:
        if _ACTIVE_POOL.ctor_kwargs == ctor_kwargs and _ACTIVE_POOL.world == world:
            return _ACTIVE_POOL
        raise RuntimeError("ICV_WORLD: this process already drives a worker pool built for another WanVideoGenerator "
                           f"({_ACTIVE_POOL.ctor_kwargs}); close it (generator._pool.close()) before building a different one")
    pool = WorkerPool(world, ctor_kwargs)
    if pool.world == 1:
        _ACTIVE_POOL = None
        return None
    _ACTIVE_POOL = pool
    return _ACTIVE_POOL


_DEGRADED = None      # kept for callers of degraded(): the client never holds a process group, so nothing can wedge it


def degraded() -> bool:
    """Rounds 4-5: True after a failed multi-GPU start whose process group could not be torn down in the CALLER's process.
    Since round 6 the caller's process holds no process group (every rank is a worker), so this stays False."""
    return _DEGRADED is not None


def layout_cache():
    """In a worker: the parallel layouts its start-up probe built (process groups are created collectively, once per plan;
    the pipeline reuses them).  None in any other process - the client of a pool never runs the pipeline."""
    return _WORKER_LAYOUTS


def requested_world() -> int:
    """ICV_WORLD = N | auto (= every visible GPU).  1 inside a worker, when unset, or when the process already is a
    rank of somebody else's job (torch.distributed initialised / launched by torch.distributed.run)."""
    v = os.environ.get("ICV_WORLD", "").strip().lower()
    if not v or os.environ.get("ICV_WORKER_RANK") is not None or degraded():
        return 1
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
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


def worker_env(base: dict, rank: int, world: int, port: int, spec_path: str) -> dict:
    """The COMPLETE environment of one rank, composed before its process exists (HIP / HSA / RCCL read theirs at
    initialisation).  ``base`` (the caller's os.environ) is copied, never written."""
    from .seqpar import apply_rccl_channel_cap
    env = dict(base, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), ICV_WORKER_RANK=str(rank),
               ICV_WORKER_SPEC=spec_path, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.pop("ICV_WORLD", None)
    # the host driver supports dmabuf IPC only: RCCL and hipIpc handles fail with the legacy mode
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # one pull stream per peer (copy-engine K|V transport) + the launch stream + torch's / RCCL's own: a pending pull blocks its
    # hardware queue (profiles/r05/kv_contention.md, "pending waits"), so nothing else may share it
    env.setdefault("GPU_MAX_HW_QUEUES", "16")
    # RCCL channel cap of the K|V transports that run RCCL kernels: the ranks' environments only (the caller's own
    # communicators keep RCCL's default)
    apply_rccl_channel_cap(int(base.get("ICV_RCCL_MAX_CHANNELS", "-1")), env=env)
    return env


class _Blob:
    """A tmpfs file mapped by the client and the workers: the uint8 buffers of a request, or the frames of its answer."""

    def __init__(self, path: str):
        self.path = path

    def write(self, arrays: List[np.ndarray]) -> List[tuple]:
        shapes = [tuple(a.shape) for a in arrays]
        total = int(sum(int(np.prod(s)) for s in shapes))
        mm = np.memmap(self.path, dtype=np.uint8, mode="w+", shape=(max(total, 1),))
        off = 0
        for a in arrays:
            n = int(a.size)
            mm[off:off + n].reshape(a.shape)[...] = a          # accepts non-contiguous slices (np.stack(...)[:93])
            off += n
        mm.flush()
        del mm
        return shapes

    def read(self, shapes: List[tuple]) -> List[np.ndarray]:
        total = int(sum(int(np.prod(s)) for s in shapes))
        mm = np.memmap(self.path, dtype=np.uint8, mode="r", shape=(max(total, 1),))
        out, off = [], 0
        for s in shapes:
            n = int(np.prod(s))
            out.append(np.array(mm[off:off + n]).reshape(s))
            off += n
        del mm
        return out


class WorkerPool:
    """The client's handle on the N worker processes (ranks 0 .. N-1) and the process group THEY share."""

    def __init__(self, world: int, ctor_kwargs: dict, backend: Optional[str] = None):
        self.world, self.ctor_kwargs = world, dict(ctor_kwargs)
        self._closed = True
        self.backend = backend or os.environ.get("ICV_DIST_BACKEND", "nccl")
        self.timeout_s = float(os.environ.get("ICV_WORLD_TIMEOUT_S", "3600"))
        self.probe_timeout_s = float(os.environ.get("ICV_WORLD_PROBE_TIMEOUT_S", "180"))
        self.init_timeout_s = min(self.timeout_s, float(os.environ.get("ICV_WORLD_INIT_TIMEOUT_S", "600")))
        self.plan, self.failed_plans, self.procs, self.conns = None, [], [], []
        self.rank_env, self.construction_log, self.rank0_settings, self.last_reply = [], "", {}, None
        self._dir = self._shm = None
        # Staged start (the same ladder as bench.py's launch_guard): the requested layout / K|V transport, then the same layout
        # with RCCL's plain all-gather, then `sp` on the world group (no sub-groups at all), then ONE GPU.  Each plan is PROBED
        # before any weights are loaded: every rank builds the plan's process groups and runs one small collective on each
        # plus one K|V exchange with the plan's transport, under a deadline; a raised error or a hung rank abandons that
        # plan (its processes are killed - the client holds nothing of theirs) and the next plan starts from fresh processes.
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
            self._kill()
        if self.plan is None:
            print(f"[icvideo] no multi-GPU plan could be started ({len(self.failed_plans)} tried); continuing on ONE GPU", file=sys.stderr)
            self.world = 1
            self._cleanup_dirs()
            return
        self._closed = False
        atexit.register(self.close)
        print(f"[icvideo] {world} ranks in {world} worker processes, this process is their client (backend {self.backend}, layout "
              f"{self.plan[0]}, K|V exchange {self.plan[1]}; logs in {self._dir})", file=sys.stderr)
        try:
            self._wait_ready()
        except BaseException:
            self.close()
            raise

    def _resolved(self, plan):
        mode, kv = plan
        if mode == "auto":
            mode = "cfg+sp" if self.world % 2 == 0 else "sp"
        return (mode, kv)

    # -- start-up ------------------------------------------------------------------------------------------------------
    def _start(self, k: int, plan) -> Optional[str]:
        """Spawn the N workers for ``plan``, accept their connections, collect every rank's probe verdict.  None = all ok."""
        world = self.world
        port = _free_port()
        self._cleanup_dirs()
        self._dir = tempfile.mkdtemp(prefix="icv_world_")
        shm_root = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        self._shm = tempfile.mkdtemp(prefix="icv_pool_", dir=shm_root)
        self._req, self._resp = _Blob(os.path.join(self._shm, "request.u8")), _Blob(os.path.join(self._shm, "frames.u8"))
        sock_path = os.path.join(self._dir, "ctl.sock")
        token = os.urandom(16).hex()
        spec = dict(ctor=self.ctor_kwargs, backend=self.backend, init_method=f"tcp://127.0.0.1:{port}", timeout_s=self.timeout_s,
                    init_timeout_s=self.init_timeout_s, factory=os.environ.get("ICV_WORKER_FACTORY"), plan=plan, plan_index=k,
                    probe_timeout_s=self.probe_timeout_s, sock=sock_path, token=token, request=self._req.path, frames=self._resp.path)
        spec_path = os.path.join(self._dir, "spec.pkl")
        with open(spec_path, "wb") as f:
            pickle.dump(spec, f)
        srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        srv.bind(sock_path)
        srv.listen(world)
        srv.settimeout(0.2)
        self.procs, self.conns, self.rank_env = [], [None] * world, [None] * world
        try:
            for r in range(world):
                log = open(os.path.join(self._dir, f"worker{r}.log"), "wb")
                self.procs.append((subprocess.Popen([sys.executable, "-m", "infinicube_amd.videogen.multigpu"],
                                                    env=worker_env(os.environ, r, world, port, spec_path),
                                                    stdout=log, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL), log))
            # every rank connects and says who it is (and under which environment its runtime will initialise)
            t_end = time.time() + self.init_timeout_s
            while any(c is None for c in self.conns):
                self._check_alive()
                if time.time() > t_end:
                    return f"ranks {[r for r, c in enumerate(self.conns) if c is None]} never connected to the client"
                try:
                    s, _ = srv.accept()
                except socket.timeout:
                    continue
                s.settimeout(None)
                conn = mpc.Connection(s.detach())
                kind, hello = conn.recv() if conn.poll(30.0) else ("", {})
                if kind != "hello" or hello.get("token") != token:
                    conn.close()
                    continue
                self.conns[hello["rank"]] = conn
                self.rank_env[hello["rank"]] = {k2: hello[k2] for k2 in ("rank", "pid", "hip_initialised_at_start") + _RANK_ENV_KEYS}
            # the probe: rendezvous of the data-path group + the plan's groups + one exchange; every rank reports the gathered list
            results = self._collect("probe", self.init_timeout_s + self.probe_timeout_s + 60.0)
        except Exception as e:  # noqa: BLE001
            return f"{type(e).__name__}: {e}"
        finally:
            srv.close()
        verdicts = results[0]
        bad = [f"rank {r}: {m}" for r, m in enumerate(verdicts) if m != "ok"]
        return "; ".join(bad) if bad else None

    def _collect(self, kind: str, deadline_s: float) -> list:
        """One message of ``kind`` from every rank (rank order), watching the processes while waiting; an "error" message or a
        dead worker raises with ITS text."""
        got = [None] * self.world
        t_end = time.time() + deadline_s
        while any(g is None for g in got):
            pending = [c for r, c in enumerate(self.conns) if got[r] is None]
            ready = mpc.wait(pending, timeout=0.2)
            for c in ready:
                r = self.conns.index(c)
                try:
                    k, payload = c.recv()
                except (EOFError, OSError):
                    self._blame_dead_worker()
                    raise RuntimeError(f"multi-GPU worker rank {r} closed its connection while the client waited for '{kind}'")
                if k == "error":
                    raise RuntimeError(f"multi-GPU worker rank {r} failed:\n{payload}")
                if k != kind:
                    raise RuntimeError(f"multi-GPU worker rank {r} sent '{k}' while the client waited for '{kind}'")
                got[r] = payload
            if not ready:
                self._check_alive([r for r, g in enumerate(got) if g is None])
                if time.time() > t_end:
                    raise RuntimeError(f"ranks {[r for r, g in enumerate(got) if g is None]} did not report '{kind}' within {deadline_s:.0f} s")
        return got

    def _wait_ready(self):
        """Every rank has built its generator (weights resident).  Rank 0's constructor output is what the caller would have
        seen from a single-process generator: kept, and printed by WanVideoGenerator.__init__."""
        ready = self._collect("ready", self.timeout_s)
        self.construction_log, self.rank0_settings = ready[0]["stdout"], ready[0]["settings"]
        for r, info in enumerate(ready):
            self.rank_env[r]["device"] = info["device"]

    def plan_record(self) -> dict:
        """What runs behind the caller: the plan that passed its probe, what failed before it, and per rank the process and the
        runtime environment it initialised under."""
        return dict(world=self.world, backend=self.backend, plan=list(self.plan) if self.plan else None, failed_plans=list(self.failed_plans),
                    client=dict(pid=os.getpid(), joined_process_group=False, hip_initialised=bool(torch.cuda.is_available() and torch.cuda.is_initialized())),
                    ranks=[dict(e) for e in self.rank_env])

    def client_pipeline(self, device, torch_dtype):
        """The ``.pipe`` of a generator whose ranks live in this pool: the pipeline's SETTINGS (sampling steps, CFG scale,
        layout ...; sent with every request) and no components - nothing is loaded in the caller's process."""
        from .pipeline import WanVideoPipeline
        pipe = WanVideoPipeline(device, torch_dtype)
        for k, v in self.rank0_settings.items():       # what rank 0's pipeline holds after construction (a factory may have set them)
            setattr(pipe, k, v)
        pipe.parallelism, pipe.kv_exchange = self.plan
        pipe.remote = True
        return pipe

    # -- liveness ------------------------------------------------------------------------------------------------------
    def _check_alive(self, ranks=None):
        """Raise with a dead worker's log (``ranks``: only these - a rank that already answered may have left on its own)."""
        for r, (p, _) in enumerate(self.procs):
            if ranks is not None and r not in ranks:
                continue
            rc = p.poll()
            if rc is not None:
                tail = ""
                try:
                    with open(os.path.join(self._dir, f"worker{r}.log"), "rb") as f:
                        tail = f.read()[-2000:].decode(errors="replace")
                except OSError:
                    pass
                raise RuntimeError(f"multi-GPU worker rank {r} exited with code {rc}:\n{tail}")

    def _blame_dead_worker(self, grace_s: float = 10.0):
        """A closed connection usually means a worker is going down: give it a moment to exit, then report ITS log."""
        deadline = time.time() + grace_s
        while time.time() < deadline:
            self._check_alive()
            time.sleep(0.2)

    # -- one request -----------------------------------------------------------------------------------------------------
    def generate(self, semantic: np.ndarray, coordinate: np.ndarray, call_kwargs: dict, pipe) -> np.ndarray:
        """Run one request on the N ranks; returns the frames uint8 [N, H, W, 3] rank 0 decoded."""
        if self._closed:
            raise RuntimeError("WorkerPool.generate: the pool is closed")
        self._check_alive()
        if call_kwargs.get("seed") is None:
            raise ValueError("WorkerPool.generate: the caller resolves seed=None to one drawn integer for all ranks (WanVideoGenerator.generate does)")
        shapes = self._req.write([semantic, coordinate])
        msg = dict(shapes=shapes, call=call_kwargs, settings={k: getattr(pipe, k) for k in _PIPE_SETTINGS if hasattr(pipe, k)})
        try:
            for c in self.conns:
                c.send(("generate", msg))
            done = self._collect("done", self.timeout_s)
        except BaseException:
            # a rank that failed leaves its peers inside collectives nobody will complete: the pool is finished
            self.close(graceful=False)
            raise
        self.last_reply = done[0]
        if done[0].get("kv_autotune") and os.environ.get("ICV_QUIET", "0") != "1":
            a = done[0]["kv_autotune"]
            print(f"[icvideo] K|V exchange autotune on the ranks: {a['chosen'][0]} x {a['chosen'][1]} chunks ({a['seconds']:.1f} s)", file=sys.stderr)
        return self._resp.read([tuple(done[0]["frames_shape"])])[0]

    def close(self, graceful: bool = True):
        if getattr(self, "_closed", True):
            return
        self._closed = True
        try:
            if graceful and all(p.poll() is None for p, _ in self.procs):
                for c in self.conns:
                    with contextlib.suppress(Exception):
                        c.send(("exit", None))
                deadline = time.time() + 30
                for p, _ in self.procs:
                    try:
                        p.wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        pass
        finally:
            self._kill()
            self._cleanup_dirs(keep_logs=not graceful)

    def _kill(self):
        for c in self.conns:
            if c is not None:
                with contextlib.suppress(Exception):
                    c.close()
        self.conns = []
        for p, log in self.procs:
            if p.poll() is None:
                p.kill()          # exactly the PIDs this pool started
                p.wait()
            log.close()
        self.procs = []

    def _cleanup_dirs(self, keep_logs: bool = False):
        if self._shm:
            shutil.rmtree(self._shm, ignore_errors=True)
            self._shm = None
        if self._dir and not keep_logs and os.environ.get("ICV_WORLD_KEEP_LOGS", "0") != "1":
            shutil.rmtree(self._dir, ignore_errors=True)
            self._dir = None


# -- worker side: the start-up probe of one plan ---------------------------------------------------------------------------
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
            # the device the rank's pipeline will run on: its own GPU under RCCL; cuda:0 for gloo ranks that SHARE one GPU (the
            # 1-GPU rehearsals of tests/test_multigpu_rccl.py: ICV_TEST_SHARE_GPU=1); the CPU for the gloo twins
            if backend == "nccl":
                dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
            elif os.environ.get("ICV_TEST_SHARE_GPU") == "1" and torch.cuda.is_available():
                dev = torch.device("cuda", 0)
            else:
                dev = torch.device("cpu")
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
                kg = KVGather(plan_s, lay.sp_group, "allgather" if kv == "auto" else kv.split("+")[0])   # "+arrival" is the consumer's business
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


# -- worker side: main ---------------------------------------------------------------------------------------------------
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
    hip_up = bool(torch.cuda.is_available() and torch.cuda.is_initialized())     # must be False: the environment below is read at initialisation
    _exit_when_parent_dies(os.getppid())
    with open(os.environ["ICV_WORKER_SPEC"], "rb") as f:
        spec = pickle.load(f)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    s.connect(spec["sock"])
    conn = mpc.Connection(s.detach())
    conn.send(("hello", dict(rank=rank, pid=os.getpid(), token=spec["token"], hip_initialised_at_start=hip_up,
                             **{k: os.environ.get(k) for k in _RANK_ENV_KEYS})))
    try:
        import torch.distributed as dist
        to = datetime.timedelta(seconds=spec["timeout_s"])
        # the data-path group's timeout bounds its rendezvous and every RCCL collective (all short); the gloo side group is
        # used by the probe's verdict and the request-scoped host collectives (seed, VAE tiles on CPU twins)
        dist.init_process_group(spec["backend"], init_method=spec["init_method"], rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=spec.get("init_timeout_s", 600.0)))
        ctrl = dist.new_group(backend="gloo", timeout=to)
        # the plan's start-up probe, before any weights are loaded (a failed plan is abandoned cheaply): every rank learns
        # every rank's verdict and tells the client
        layouts = {}
        mine = probe_plan(spec["plan"], spec.get("plan_index", 0), rank, world, spec["backend"], layouts, spec.get("probe_timeout_s", 180.0))
        results = [None] * world
        dist.all_gather_object(results, mine, group=ctrl)
        conn.send(("probe", results))
        if any(m != "ok" for m in results):
            print(f"[worker {rank}] plan {spec['plan']} failed its probe: {results}", flush=True)
            os._exit(4)               # no clean teardown: the group may be wedged; the client starts the next plan from fresh processes
        # this file runs as __main__ in a worker; the pipeline imports the package module: hand the layouts to THAT module object
        import infinicube_amd.videogen.multigpu as canonical
        canonical._WORKER_LAYOUTS = layouts
        from .inference import WanVideoGenerator
        from .pipeline import WanVideoPipeline
        factory = _resolve(spec["factory"]) if spec.get("factory") else None
        out = io.StringIO()
        with contextlib.redirect_stdout(out):    # the constructor's progress lines: rank 0's are the caller's (the client prints them)
            gen = WanVideoGenerator(**spec["ctor"], pipeline_factory=factory)   # device "cuda:0" resolves to cuda:LOCAL_RANK
        gen.pipe.parallelism, gen.pipe.kv_exchange = spec["plan"]           # the plan that passed its probe (the client re-sends both per request)
        dev = getattr(getattr(gen.pipe, "_ops", None), "device", None) or WanVideoPipeline.resolve_device(gen.pipe.device)
        conn.send(("ready", dict(stdout=out.getvalue(), device=str(dev),
                                 settings={k: getattr(gen.pipe, k) for k in _PIPE_SETTINGS if hasattr(gen.pipe, k)})))
        print(f"[worker {rank}] ready", flush=True)
    except BaseException:  # noqa: BLE001
        with contextlib.suppress(Exception):
            conn.send(("error", traceback.format_exc()[-4000:]))
        traceback.print_exc()
        sys.stdout.flush()
        os._exit(5)
    request, frames = _Blob(spec["request"]), _Blob(spec["frames"])
    while True:
        try:
            cmd, msg = conn.recv()
        except (EOFError, OSError):
            print(f"[worker {rank}] the client closed the connection: exiting", flush=True)
            os._exit(3)
        if cmd == "exit":
            break
        try:
            sem, co = request.read(msg["shapes"])
            for k, v in msg["settings"].items():
                setattr(gen.pipe, k, v)
            n, h, w, _ = msg["shapes"][0]
            c = msg["call"]
            tuned_before = getattr(gen.pipe, "kv_autotune", None)
            # rank 0 blends the decoded tiles into frames; the other ranks compute their share of the tiles (join_decode) and
            # return nothing
            got = gen.pipe(prompt=c["prompt"], negative_prompt=c["negative_prompt"], semantic_buffer_video=gen._ndarray_to_pil_list(sem),
                           coordinate_buffer_video=gen._ndarray_to_pil_list(co), height=h, width=w, num_frames=n, seed=c["seed"],
                           tiled=c["tiled"], return_latents=rank != 0, join_decode=True)
            reply = dict(rank=rank)
            if rank == 0:
                arr = np.stack([np.asarray(f) for f in got])
                frames.write([arr])
                reply["frames_shape"] = tuple(arr.shape)
                tuned = getattr(gen.pipe, "kv_autotune", None)
                if tuned is not None and tuned is not tuned_before:
                    reply["kv_autotune"] = tuned
            conn.send(("done", reply))
            print(f"[worker {rank}] request done", flush=True)
        except BaseException:  # noqa: BLE001
            with contextlib.suppress(Exception):
                conn.send(("error", traceback.format_exc()[-4000:]))
            traceback.print_exc()
            sys.stdout.flush()
            os._exit(6)      # the peers sit in collectives this rank will never join: the client tears the pool down
    from .seqpar import _NativeComm
    _NativeComm.close_all()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(worker_main())
