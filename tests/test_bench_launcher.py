"""`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment) starts its own N ranks (bench.launch_ranks):
the parent makes no GPU call, gives every child RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as
`torch.distributed.run --nnodes=1 --nproc-per-node N` would, relays rank 0's one JSON line and fails when a child fails --
and it refuses to print a one-GPU line under `--gpus N` on a box without N devices."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra, tmp_path, worker_src=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    if worker_src is not None:
        w = tmp_path / "worker.py"
        w.write_text(textwrap.dedent(worker_src))
        env["SVGR_BENCH_WORKER"] = str(w)
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=300)


STUB = """
    import json, os, sys
    import torch.distributed as dist
    dist.init_process_group("gloo")          # (MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE come from the launcher)
    import torch
    t = torch.tensor([float(os.environ["LOCAL_RANK"])], dtype=torch.float64)
    dist.all_reduce(t)
    if int(os.environ["RANK"]) == int(os.environ.get("STUB_FAIL_RANK", "-1")):
        dist.destroy_process_group()
        sys.exit(3)
    if dist.get_rank() == 0:
        print("not json: a stray line")
        print(json.dumps({"n_gpus": dist.get_world_size(), "sum_of_local_ranks": float(t[0]), "argv": sys.argv[1:],
                          "master": os.environ["MASTER_ADDR"]}))
    dist.barrier()
    dist.destroy_process_group()
"""


def test_launcher_starts_n_ranks_and_relays_rank0s_line(tmp_path):
    r = _run(["--gpus", "3", "--steps", "7"], {"SVGR_BENCH_DEVICE": "0"}, tmp_path, STUB)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["sum_of_local_ranks"] == 3.0 and d["master"] == "127.0.0.1"
    assert d["argv"] == ["--gpus", "3", "--steps", "7"]


def test_launcher_fails_when_a_rank_fails(tmp_path):
    r = _run(["--gpus", "2"], {"SVGR_BENCH_DEVICE": "0", "STUB_FAIL_RANK": "1"}, tmp_path, STUB)
    assert r.returncode != 0
    assert "exit codes" in r.stderr and r.stdout.strip() == ""


HANG_STUB = """
    import os, sys, time
    if int(os.environ["RANK"]) == 1:
        sys.exit(5)            # (dies during start-up: bad device, import error)
    time.sleep(600)            # (rank 0 would sit in init_process_group / a barrier until torch's timeout)
"""


def test_launcher_does_not_wait_for_a_stuck_rank_when_another_died(tmp_path):
    import time

    t0 = time.monotonic()
    r = _run(["--gpus", "3"], {"SVGR_BENCH_DEVICE": "0"}, tmp_path, HANG_STUB)
    assert r.returncode == 5, (r.returncode, r.stderr)
    assert time.monotonic() - t0 < 60, "the launcher waited for the ranks that were stuck"
    assert "exit codes" in r.stderr and r.stdout.strip() == ""


def test_launcher_refuses_more_gpus_than_the_box_has(tmp_path):
    import torch

    if torch.cuda.device_count() >= 8:
        pytest.skip("this box has eight GPUs")
    r = _run(["--gpus", "8"], {}, tmp_path, STUB)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "GPU(s)" in r.stderr


@pytest.mark.gpu
def test_bench_gpus_2_launches_itself_on_one_gpu(tmp_path):
    """The real worker: two gloo ranks on GPU 0 (the rehearsal switches of bench.py), started by `bench.py --gpus 2` alone.
    The line says n_gpus 2, strong scaling of the 8192^2 drawing, and checks rank 0's strips against the oracle."""
    r = _run(["--gpus", "2", "--steps", "6", "--warmup", "2", "--no-companions", "--cpu-paths", "200"],
             {"SVGR_BENCH_BACKEND": "gloo", "SVGR_BENCH_DEVICE": "0"}, tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["parity"]["bad"] == 0 and d["parity"]["values"] > 0, d["parity"]
