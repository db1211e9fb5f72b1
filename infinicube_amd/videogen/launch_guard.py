"""Fail-soft launches of the N-rank denoising job: staged fallback + per-phase watchdog (bench.py --gpus N).

The reference runs on one GPU [R infinicube/inference/guidance_buffer_generation.py:759-766]; the sequence-parallel path is
this build's own (north_star: "RCCL all-gather of K/V over xGMI"), and its first contact with a real multi-GPU node must
not be able to end without a result: a raised exception in ``new_group``, or one rank wedged inside a blocking RCCL call,
would otherwise cost the whole measurement.  A wedged C call cannot be interrupted from inside its process, so recovery is
done from OUTSIDE: every rank the launcher starts (``torch.distributed.run`` or ``bench.py``'s own launcher) is a small
*supervisor* that never touches the GPU.  The supervisors share a TCP store (the launcher's ``MASTER_ADDR:MASTER_PORT``
rendezvous) and walk the same list of *attempts* (layout / transport plans, most capable first):

    for each attempt k:
        every supervisor starts its worker (the real rank; own MASTER_PORT, plan + phase-file + result-file in its env)
        loop:  worker exited 0          -> publish done;   all ranks done -> success
               worker exited non-zero   -> publish fail(k, rank, phase, reason)
               worker's current phase ran past its budget (it reports phases through a file) -> publish fail(k, ..., "hung")
               somebody else published fail(k) -> stop
        on fail: kill the worker (exactly the process group started here), acknowledge, next attempt

Rank 0's supervisor prints the ONE JSON line: the worker's result with the attempt history attached, or — when every attempt
failed — an error record naming the attempt, rank and phase of each failure.  Workers report phases with ``PhaseReporter``
(also the hook for failure injection in tests: ``ICV_GUARD_INJECT``)."""
from __future__ import annotations

import json
import os
import signal
import socket
import subprocess
import sys
import tempfile
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional

# phase budgets in seconds (a worker may declare its own budget for a phase it can size: PhaseReporter(name, budget_s=))
# Sized so that THREE attempts that each hang in their longest phase still end (with the error record) inside the half hour a
# benchmark driver typically allows: a rank's real phases take 5-60 s each at 14B / N = 8, the budgets are ~4x that.
DEFAULT_PHASE_BUDGET_S = {
    "start": 240.0,      # interpreter + first `import torch` of a fresh box (1-2 min while the image pages in)
    "init": 120.0,       # init_process_group + first all-reduce
    "groups": 120.0,     # sub-groups + a first small collective on each (lazy communicator creation happens HERE)
    "setup": 180.0,      # weights into HBM, workspace, caches
    "autotune": 90.0,
    "warmup": 120.0,
    "timed": 240.0,
    "report": 90.0,
}
FALLBACK_BUDGET_S = 180.0
TOTAL_BUDGET_S = 1500.0      # no new attempt is started after this much supervisor time (ICV_GUARD_TOTAL_BUDGET_S)


@dataclass
class Attempt:
    label: str                                   # goes into the result: which plan ran / failed
    env: Dict[str, str] = field(default_factory=dict)   # plan handed to the worker through its environment


# ---------------------------------------------------------------------------------------------------------------------
# worker side
class PhaseReporter:
    """``phase("groups")`` appends one line to the file the supervisor watches (name, wall-clock, optional budget) and echoes
    it on stderr.  Without a supervisor (plain single-GPU run) it only echoes.  ``ICV_GUARD_INJECT`` =
    ``attempt:rank:phase:raise|hang[,…]`` makes the matching phase entry raise / block forever (tests of the fallback)."""

    def __init__(self, rank: Optional[int] = None):
        self.path = os.environ.get("ICV_GUARD_PHASE_FILE")
        self.attempt = int(os.environ.get("ICV_GUARD_ATTEMPT", "0"))
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.current = "start"
        self._inject = []
        # failure injection: honoured only when the test harness also sets ICV_TEST_HOOKS=1
        inject = os.environ.get("ICV_GUARD_INJECT", "") if os.environ.get("ICV_TEST_HOOKS") == "1" else ""
        for item in filter(None, inject.split(",")):
            a, r, ph, kind = item.split(":")
            self._inject.append((int(a), int(r), ph, kind))

    def __call__(self, name: str, budget_s: Optional[float] = None) -> None:
        self.current = name
        if self.path:
            with open(self.path, "a") as f:
                f.write(f"{name}\t{time.time():.3f}\t{'' if budget_s is None else f'{budget_s:.1f}'}\n")
        print(f"[rank {self.rank}] phase {name}", file=sys.stderr, flush=True)
        for a, r, ph, kind in self._inject:
            if (a, r, ph) == (self.attempt, self.rank, name):
                if kind == "raise":
                    raise RuntimeError(f"injected failure: attempt {a} rank {r} phase {ph}")
                print(f"[rank {self.rank}] injected hang in phase {name}", file=sys.stderr, flush=True)
                while True:
                    time.sleep(3600)


def supervised() -> bool:
    return os.environ.get("ICV_GUARD_ROLE") == "worker"


def write_result(obj: dict) -> bool:
    """Rank 0's worker hands its result to the supervisor (which prints it).  False = no supervisor: caller prints."""
    path = os.environ.get("ICV_GUARD_RESULT_FILE")
    if not path:
        return False
    tmp = path + ".tmp"
    with open(tmp, "w") as f:
        json.dump(obj, f)
    os.replace(tmp, path)
    return True


# ---------------------------------------------------------------------------------------------------------------------
# supervisor side
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _kill_group(proc: subprocess.Popen, grace_s: float = 5.0) -> None:
    """Stop exactly the process group this supervisor started (the worker runs in its own session)."""
    if proc.poll() is not None:
        return
    try:
        os.killpg(proc.pid, signal.SIGTERM)
    except (ProcessLookupError, PermissionError):
        pass
    t0 = time.time()
    while proc.poll() is None and time.time() - t0 < grace_s:
        time.sleep(0.05)
    if proc.poll() is None:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
        proc.wait()


def _last_phase(path: str):
    """(name, started_at, declared budget or None) of the worker's current phase."""
    try:
        with open(path) as f:
            lines = [ln for ln in f.read().splitlines() if ln]
    except OSError:
        lines = []
    if not lines:
        return None
    parts = lines[-1].split("\t")
    return parts[0], float(parts[1]), (float(parts[2]) if len(parts) > 2 and parts[2] else None)


def _tail(path: str, n: int = 1500) -> str:
    try:
        with open(path, "rb") as f:
            return f.read()[-n:].decode(errors="replace")
    except OSError:
        return ""


class Supervisor:
    def __init__(self, rank: int, world: int, attempts: List[Attempt], worker_cmd: List[str], *, store=None,
                 budgets: Optional[Dict[str, float]] = None, budget_scale: float = 1.0, log=None):
        self.rank, self.world, self.attempts, self.cmd = rank, world, attempts, worker_cmd
        self.budgets = dict(DEFAULT_PHASE_BUDGET_S)
        self.budgets.update(budgets or {})
        for item in filter(None, os.environ.get("ICV_GUARD_BUDGETS", "").split(",")):     # "groups=20,timed=90"
            k, v = item.split("=")
            self.budgets[k] = float(v)
        self.scale = budget_scale * float(os.environ.get("ICV_GUARD_BUDGET_SCALE", "1"))
        t0 = time.time()
        self.log = log or (lambda m: print(f"[guard {rank} +{time.time() - t0:6.1f}s] {m}", file=sys.stderr, flush=True))
        self.dir = tempfile.mkdtemp(prefix=f"icv_guard_r{rank}_")
        self.proc: Optional[subprocess.Popen] = None
        self.store = store if store is not None else self._connect_store()

    def _connect_store(self):
        """The launcher's own rendezvous (MASTER_ADDR:MASTER_PORT; torch.distributed.run's agent store when it has one),
        under a private prefix.  The workers rendezvous on a DIFFERENT port per attempt, published here by rank 0."""
        import datetime
        from torch.distributed import PrefixStore, rendezvous
        store, _, _ = next(iter(rendezvous("env://", self.rank, self.world, timeout=datetime.timedelta(seconds=300))))
        store.set_timeout(datetime.timedelta(seconds=300))
        return PrefixStore("icv_guard", store)

    # -- small store helpers (keys are written once, never deleted) ------------------------------------------------
    def _has(self, key: str) -> bool:
        return bool(self.store.check([key]))

    def _wait_all(self, prefix: str, deadline_s: float, abort_key: Optional[str] = None) -> str:
        """'ok' when every rank set ``prefix{rank}``, 'abort' when ``abort_key`` appeared first, 'timeout' otherwise."""
        t_end = time.time() + deadline_s
        pending = set(range(self.world))
        while time.time() < t_end:
            if abort_key and self._has(abort_key):
                return "abort"
            pending = {r for r in pending if not self._has(f"{prefix}{r}")}
            if not pending:
                return "ok"
            time.sleep(0.1)
        return "timeout"

    def _fail(self, k: int, phase: str, reason: str) -> None:
        key = f"fail{k}"
        if not self._has(key):          # first reporter wins (a benign race: both writers describe a real failure)
            self.store.set(key, json.dumps(dict(rank=self.rank, phase=phase, reason=reason[-1200:])))

    # -- one attempt ---------------------------------------------------------------------------------------------------
    def _run_attempt(self, k: int, att: Attempt) -> Optional[dict]:
        """None = every rank finished; otherwise the failure record (shared by all supervisors)."""
        if self.rank == 0:
            # the port is free NOW; the worker binds it a moment later.  A collision shows up as a failed rendezvous of this
            # attempt (phase "init"), which the ladder treats like any other failure and retries on a fresh port
            self.store.set(f"port{k}", str(_free_port()))
        port = self.store.get(f"port{k}").decode()
        phase_file = os.path.join(self.dir, f"phase{k}.txt")
        result_file = os.path.join(self.dir, f"result{k}.json")
        log_file = os.path.join(self.dir, f"worker{k}.log")
        env = dict(os.environ)
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)        # the workers' own rendezvous: rank 0's worker hosts its store
        env.update(att.env)
        # the HIP runtime's own variables are read at the rank's first HIP call: they belong in its environment BEFORE it exists
        # (hardware queues for the copy-engine K|V transport's pull streams; dmabuf IPC for RCCL / hipIpc handles)
        env.setdefault("GPU_MAX_HW_QUEUES", "16")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(ICV_GUARD_ROLE="worker", ICV_GUARD_ATTEMPT=str(k), ICV_GUARD_LABEL=att.label, ICV_GUARD_PHASE_FILE=phase_file,
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(self.rank), WORLD_SIZE=str(self.world))
        if self.rank == 0:
            env["ICV_GUARD_RESULT_FILE"] = result_file
        with open(phase_file, "w") as f:
            f.write(f"start\t{time.time():.3f}\t\n")
        logf = open(log_file, "wb")
        self.log(f"attempt {k} ({att.label}): starting the worker (log {log_file})")
        self.proc = subprocess.Popen(self.cmd, env=env, stdout=logf, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL,
                                     start_new_session=True)
        failed = False
        while True:
            rc = self.proc.poll()
            ph = _last_phase(phase_file) or ("start", time.time(), None)
            if rc is not None:
                if rc == 0:
                    break
                self._fail(k, ph[0], f"worker exited with code {rc}: {_tail(log_file)}")
                failed = True
                break
            budget = (ph[2] if ph[2] is not None else self.budgets.get(ph[0].split(":")[0], FALLBACK_BUDGET_S)) * self.scale
            if time.time() - ph[1] > budget:
                self._fail(k, ph[0], f"phase '{ph[0]}' still running after its {budget:.0f} s budget (hung?): {_tail(log_file, 600)}")
                failed = True
                break
            if self._has(f"fail{k}"):
                failed = True
                break
            time.sleep(0.2)
        if not failed:
            self.store.set(f"done{k}_{self.rank}", "1")
        # ONE verdict per attempt, decided by rank 0's supervisor alone and published through the store: every supervisor follows it,
        # so a rank that finished long before the slowest cannot time the others out on its own clock, and no rank can read "ok"
        # while another has moved on to the next attempt.
        report_budget = self.budgets.get("report", FALLBACK_BUDGET_S) * self.scale + 60.0
        if self.rank == 0:
            if failed:
                verdict = "fail"
            else:
                # every other rank either reports completion or publishes fail{k}; the slowest may still be in its last phase
                state = self._wait_all(f"done{k}_", report_budget, abort_key=f"fail{k}")
                if state == "timeout":
                    self._fail(k, "report", "not every rank reported completion")
                verdict = "ok" if state == "ok" else "fail"
            self.store.set(f"verdict{k}", verdict)
        else:
            # rank 0 decides; it publishes within its own worker's phase budgets + the report budget (a dead rank 0 = the store is
            # gone: the get raises, which ends this supervisor too)
            t_end = time.time() + max(self.budgets.values()) * self.scale + report_budget + 120.0
            while not self._has(f"verdict{k}") and time.time() < t_end:
                if self.proc.poll() is not None and self.proc.returncode != 0 and not self._has(f"fail{k}"):
                    self._fail(k, (_last_phase(phase_file) or ("?",))[0], f"worker exited with code {self.proc.returncode}: {_tail(log_file)}")
                time.sleep(0.1)
            verdict = self.store.get(f"verdict{k}").decode() if self._has(f"verdict{k}") else "fail"
            if verdict != "ok" and not self._has(f"fail{k}"):
                self._fail(k, "report", "rank 0 published no verdict for this attempt")
        if verdict == "ok":
            logf.close()
            sys.stderr.write(_tail(log_file, 4000))
            return None
        self.log(f"attempt {k}: stopping the worker (last phase '{(_last_phase(phase_file) or ('?',))[0]}')")
        _kill_group(self.proc)
        self.log(f"attempt {k}: worker gone")
        logf.close()
        sys.stderr.write(f"---- [guard {self.rank}] attempt {k} worker log tail ----\n{_tail(log_file, 3000)}\n")
        sys.stderr.flush()
        self.store.set(f"ack{k}_{self.rank}", "1")
        self._wait_all(f"ack{k}_", 60.0)                      # every worker of the failed attempt is gone before the next starts
        rec = json.loads(self.store.get(f"fail{k}").decode())
        rec.update(attempt=k, plan=att.label)
        return rec

    def run(self) -> dict:
        """{"ok", "attempt", "plan", "failed": [records], "result": rank 0's worker result or None}"""
        history = []
        out = dict(ok=False, attempt=None, plan=None, failed=history, result=None)
        t_start, total = time.time(), float(os.environ.get("ICV_GUARD_TOTAL_BUDGET_S", TOTAL_BUDGET_S))
        try:
            for k, att in enumerate(self.attempts):
                # every supervisor started within seconds of the others and saw the same failures: the same decision everywhere
                # (rank 0's clock decides for all through the store, so that a slow rank cannot disagree)
                if k > 0:
                    if self.rank == 0:
                        self.store.set(f"go{k}", "1" if time.time() - t_start < total else "0")
                    if self.store.get(f"go{k}").decode() != "1":
                        history.append(dict(attempt=k, plan=att.label, rank=0, phase="supervisor",
                                            reason=f"not started: {total:.0f} s of supervisor time were already spent on the failed plans"))
                        break
                rec = self._run_attempt(k, att)
                if rec is None:
                    out.update(ok=True, attempt=k, plan=att.label)
                    if self.rank == 0:
                        with open(os.path.join(self.dir, f"result{k}.json")) as f:
                            out["result"] = json.load(f)
                    break
                self.log(f"attempt {k} ({att.label}) FAILED on rank {rec['rank']} in phase '{rec['phase']}': {rec['reason'][:300]}")
                history.append(rec)
        finally:
            if self.proc is not None:
                _kill_group(self.proc)
            try:      # the store may live in rank 0's supervisor: nobody leaves before everybody has read what it needs
                self.store.set(f"bye_{self.rank}", "1")
                if self.rank == 0:
                    self._wait_all("bye_", 15.0)
            except Exception:
                pass
        return out


def install_sigterm(handler) -> None:
    """A launcher that loses a rank sends SIGTERM to the others: turn it into an exception the caller's cleanup sees."""
    def _h(signum, frame):
        handler()
        raise SystemExit(128 + signum)
    signal.signal(signal.SIGTERM, _h)
