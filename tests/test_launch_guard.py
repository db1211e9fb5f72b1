"""bench.py's fail-soft N-rank launch (infinicube_amd/videogen/launch_guard.py): gloo twins of the first 8-GPU contact.

Every rank is a supervisor that runs the real rank as a child, plan by plan.  Here the child is tests/guard_worker.py (the
same distributed skeleton as bench.py's rank, no model, gloo) and failures are injected the two ways a first contact with a
new node fails: a RAISED error in group creation, and a rank that HANGS inside a phase.  Reference: none — the reference is
single-GPU [R infinicube/inference/guidance_buffer_generation.py:759-766]; this protects the build's own N-GPU measurement."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import json, os, sys
sys.path.insert(0, {root!r})
from infinicube_amd.videogen import launch_guard as guard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
plans = json.loads(os.environ["TEST_PLANS"])
attempts = [guard.Attempt(p["label"], {{"ICV_BENCH_PLAN": json.dumps(p)}}) for p in plans]
sup = guard.Supervisor(rank, world, attempts, [sys.executable, os.path.join({root!r}, "tests", "guard_worker.py")])
res = sup.run()
if rank == 0:
    print("RESULT " + json.dumps(res), flush=True)
sys.exit(0 if res["ok"] else 1)
'''

PLANS = [dict(label="cfg+sp / auto", parallelism="cfg+sp", kv_exchange="auto"),
         dict(label="cfg+sp / allgather", parallelism="cfg+sp", kv_exchange="allgather"),
         dict(label="sp / allgather", parallelism="sp", kv_exchange="allgather")]


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, extra_env, plans=PLANS, timeout=240):
    extra_env = dict(extra_env, ICV_TEST_HOOKS="1")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TEST_PLANS=json.dumps(plans), OMP_NUM_THREADS="1", PYTHONDONTWRITEBYTECODE="1", **extra_env)
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        procs.append(subprocess.Popen([sys.executable, "-c", DRIVER.format(root=ROOT)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=timeout)
            outs.append((p.returncode, o, e))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    line = [ln for ln in outs[0][1].splitlines() if ln.startswith("RESULT ")]
    assert line, f"rank 0 printed no result:\n{outs[0][2][-3000:]}"
    assert all(not o.strip() for _, o, _ in outs[1:]), "only rank 0 may print"
    return json.loads(line[0][len("RESULT "):]), outs


def test_first_plan_runs_when_nothing_fails():
    res, outs = _launch(2, {})
    assert res["ok"] and res["attempt"] == 0 and res["failed"] == [] and res["plan"] == "cfg+sp / auto"
    assert res["result"]["sum_of_ranks"] == 1.0 and res["result"]["multi_gpu"]["parallelism"] == "cfg+sp"
    assert all(rc == 0 for rc, _, _ in outs)


def test_failing_new_group_falls_back_to_the_world_group_layout():
    """new_group raises on ONE rank in the first two plans (both build sub-groups): every rank moves on together and the run
    finishes on `sp` (no sub-groups); the result names what failed, where, and with which message."""
    res, outs = _launch(4, {"ICV_TEST_FAIL_NEW_GROUP": "0:2"})
    assert res["ok"] and res["attempt"] == 1, res       # attempt 1 no longer matches the injection (attempt index differs)
    res, outs = _launch(4, {"ICV_GUARD_INJECT": "0:2:groups:raise,1:1:groups:raise"})
    assert res["ok"] and res["attempt"] == 2 and res["plan"] == "sp / allgather"
    assert [(f["attempt"], f["rank"], f["phase"]) for f in res["failed"]] == [(0, 2, "groups"), (1, 1, "groups")]
    assert "injected failure" in res["failed"][0]["reason"]
    assert res["result"]["multi_gpu"]["parallelism"] == "sp" and res["result"]["sum_of_ranks"] == 6.0
    assert all(rc == 0 for rc, _, _ in outs)


def test_raising_new_group_message_reaches_the_record():
    res, _ = _launch(2, {"ICV_TEST_FAIL_NEW_GROUP": "0:1"})
    assert res["ok"] and res["attempt"] == 1
    assert res["failed"][0]["phase"] == "groups" and res["failed"][0]["rank"] == 1 and "new_group failed" in res["failed"][0]["reason"]


def test_hung_rank_is_detected_by_the_phase_watchdog():
    """Rank 1 blocks forever inside the timed phase of the first plan (a wedged collective cannot be interrupted from inside
    its process): the supervisor of whichever rank notices first (phase budget, or the peers' collective timeout) fails the
    attempt, all workers are killed, the next plan runs."""
    res, outs = _launch(2, {"ICV_GUARD_INJECT": "0:1:timed:hang", "ICV_GUARD_BUDGETS": "timed=6"})
    assert res["ok"] and res["attempt"] == 1 and len(res["failed"]) == 1
    f = res["failed"][0]
    assert f["phase"] == "timed" and ("budget" in f["reason"] or "exited with code" in f["reason"])
    assert all(rc == 0 for rc, _, _ in outs)


def test_every_plan_failing_is_reported_not_swallowed():
    inj = ",".join(f"{k}:0:init:raise" for k in range(3))
    res, outs = _launch(2, {"ICV_GUARD_INJECT": inj})
    assert not res["ok"] and res["result"] is None and [f["attempt"] for f in res["failed"]] == [0, 1, 2]
    assert all(f["phase"] == "init" for f in res["failed"]) and all(rc == 1 for rc, _, _ in outs)


def test_bench_prints_one_json_line_even_when_it_cannot_measure():
    """No GPU here: `bench.py --gpus 1` and the self-launching `--gpus 2` both end with exactly one JSON line on stdout that
    carries "error" (and a non-zero exit code) - never with silence."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box WITHOUT a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    for argv, n in ((["--gpus", "1", "--steps", "1"], 1), (["--gpus", "2", "--steps", "1"], 2)):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=300, cwd="/tmp", env=env)
        lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
        assert p.returncode != 0 and len(lines) == 1, (p.returncode, p.stdout, p.stderr[-2000:])
        d = json.loads(lines[0])
        assert d["value"] is None and d["n_gpus"] == n and d["error"] and d["metric"].startswith("denoise steps/sec")


def test_bench_supervised_ranks_walk_every_plan_and_report(tmp_path):
    """The real bench.py under its own launcher with two ranks "sharing a GPU" that does not exist: every plan's workers fail in
    `start`, the supervisors walk all three plans together and rank 0 prints ONE line listing them."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box WITHOUT a GPU")
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}, ICV_BENCH_SHARE_GPU="1", ICV_DIST_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--model", "tiny"], capture_output=True,
                       text=True, timeout=300, cwd="/tmp", env=env)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert p.returncode != 0 and len(lines) == 1, (p.stdout, p.stderr[-2000:])
    d = json.loads(lines[0])
    assert [f["plan"] for f in d["failed_attempts"]] == ["cfg+sp / kv-exchange auto / 4 chunks", "cfg+sp / kv-exchange allgather / 4 chunks",
                                                        "sp / kv-exchange allgather / 4 chunks"]
    assert "no GPU visible" in d["error"]


def test_total_budget_stops_the_ladder():
    """A driver gives a bench run a fixed time: once ICV_GUARD_TOTAL_BUDGET_S of supervisor time is spent on failed plans no further
    plan is started (rank 0's clock decides for every rank through the store) and the record says so."""
    res, outs = _launch(2, {"ICV_GUARD_INJECT": "0:1:groups:raise", "ICV_GUARD_TOTAL_BUDGET_S": "0"})
    assert not res["ok"] and [f["phase"] for f in res["failed"]] == ["groups", "supervisor"]
    assert "not started" in res["failed"][1]["reason"] and all(rc == 1 for rc, _, _ in outs)


def test_bench_plan_ladder():
    """bench.py's ladder for a given command line: most capable plan first, duplicates removed, `--no-fallback` = one plan."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    labels = lambda argv, world: [a.label for a in bench.plan_attempts(bench.parse_args(argv), world)]   # noqa: E731
    assert labels([], 8) == ["cfg+sp / kv-exchange auto / 4 chunks", "cfg+sp / kv-exchange allgather / 4 chunks", "sp / kv-exchange allgather / 4 chunks"]
    assert labels([], 3) == ["sp / kv-exchange auto / 4 chunks", "sp / kv-exchange allgather / 4 chunks"]            # odd world: auto = sp
    assert labels(["--parallelism", "sp", "--kv-exchange", "allgather"], 8) == ["sp / kv-exchange allgather / 4 chunks"]
    assert labels(["--kv-exchange", "p2p", "--sp-chunks", "2"], 4) == ["cfg+sp / kv-exchange p2p / 2 chunks", "cfg+sp / kv-exchange allgather / 2 chunks",
                                                                        "sp / kv-exchange allgather / 4 chunks"]
    assert labels(["--no-fallback"], 8) == ["cfg+sp / kv-exchange auto / 4 chunks"]
    plan = json.loads(bench.plan_attempts(bench.parse_args([]), 8)[2].env["ICV_BENCH_PLAN"])
    assert plan == dict(parallelism="sp", kv_exchange="allgather", sp_chunks=4)
    rec = bench.error_record(bench.parse_args(["--gpus", "8"]), 8, "boom", failed=[dict(plan="x")], phase="groups")
    assert rec["value"] is None and rec["error"] == "boom" and rec["failed_phase"] == "groups" and rec["n_gpus"] == 8 and rec["unit"] == "denoise steps/s"
