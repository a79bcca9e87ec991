"""bench.py's own rank launcher (`python bench.py --gpus N` without torch.distributed.run) on CPU: two ranks end to end
in the dry-run test mode (kernels stubbed by tests/fakelib.py, gloo), and a rank that dies at start-up -- the launcher
must stop the survivor and return non-zero within seconds instead of leaving it in a collective until the timeout."""
import importlib.util
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("sm3_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _run(extra_env, timeout=240, gpus=2):
    env = dict(os.environ, SM3_BENCH_DRYRUN="1", **extra_env)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    t0 = time.monotonic()
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "1", "--warmup", "0"],
                        env=env, capture_output=True, text=True, timeout=timeout)
    return pr, time.monotonic() - t0


def test_two_ranks_end_to_end_dry_run():
    pr, _ = _run({})
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, pr.stdout          # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2"
    # the record proves what ran: ranks COUNTED by a collective (not WORLD_SIZE echoed), the backend, how the SyncBN
    # statistics travelled, one ms/step per rank
    w = out["config"]["witness"]
    assert w["rccl_ranks"] == 2 and w["backend"] == "gloo" and w["syncbn_exchange"] == "gloo"
    assert len(w["ms_per_step_per_rank"]) == 2 and w["ms_per_step_min"] <= w["ms_per_step_max"]


def test_eight_ranks_end_to_end_dry_run():
    """BASELINE config 3's rank count through bench.py's own launcher (CPU, gloo, kernels stubbed): eight ranks start, count
    each other with a collective, run the data-parallel step's whole control flow and rank 0 prints ONE line for n_gpus = 8.
    (No 8-GPU node was available to any round: this is the control flow only, config 3 stays unmeasured.)"""
    pr, _ = _run({"OMP_NUM_THREADS": "1"}, timeout=480, gpus=8)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, pr.stdout
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["n_gpus"] == 8 and out["config"]["parallelism"] == "dp8"
    w = out["config"]["witness"]
    assert w["rccl_ranks"] == 8 and len(w["ms_per_step_per_rank"]) == 8


def test_two_ranks_under_a_torchrun_style_environment():
    """How the driver starts N > 1: one process per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment
    (python -m torch.distributed.run ... bench.py --gpus N).  bench.py must then NOT spawn ranks of its own: each process is
    one rank, rank 0 prints the line."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, SM3_BENCH_DRYRUN="1", RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    lines = [[l for l in o[0].splitlines() if l.startswith("{")] for o in outs]
    assert len(lines[0]) == 1 and len(lines[1]) == 0, lines
    out = json.loads(lines[0][0])
    assert out["n_gpus"] == 2 and out["config"]["witness"]["rccl_ranks"] == 2


def test_a_rank_that_dies_stops_the_run_quickly():
    pr, dt = _run({"SM3_BENCH_FAIL_RANK": "1"})
    assert pr.returncode != 0
    assert dt < 90, dt                          # far below the 120 s process-group timeout rank 0 would sit out
    assert "[rank 1]" in pr.stderr and "injected start-up failure" in pr.stderr   # per-rank stderr, tagged
    assert "stopping the other ranks" in pr.stderr
    assert not [l for l in pr.stdout.splitlines() if l.startswith("{")]


def test_gpu_count_comes_from_sysfs_not_from_hip():
    b = _bench()
    n, source = b.visible_gpus()
    assert (n is None or isinstance(n, int)) and isinstance(source, str)
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def spawn_ranks("):src.index("def dry_run_rank(")]
    assert "torch.cuda" not in body             # the parent of the ranks never touches the GPU runtime


def test_visible_device_lists_cap_the_gpu_count(monkeypatch):
    """the pre-check of `--gpus N` honours HIP_ / ROCR_ / CUDA_VISIBLE_DEVICES (a rank beyond the list would die in
    set_device) and says which source the number came from."""
    b = _bench()
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    base, _ = b.visible_gpus()
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    n, source = b.visible_gpus()
    assert n == (1 if base is None else min(base, 1)) or (base == 0 and n == 0)
    if base is None or base >= 1:
        assert "HIP_VISIBLE_DEVICES" in source
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "")
    n, source = b.visible_gpus()
    assert n == 0 and "ROCR_VISIBLE_DEVICES" in source
