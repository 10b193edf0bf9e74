"""`python bench.py --gpus N` starts its N ranks itself (a child `python -m torch.distributed.run`, one rank per GPU): the
launch path on the CPU (gloo, no GPU touched) and, on a GPU box, a whole two-rank run of the small workload with both ranks
on the one device (gloo collectives) -- the path the driver's N > 1 runs take, minus RCCL."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


def test_bench_starts_its_own_ranks():
    p, lines = _run(["--gpus", "2", "--launch-check"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks_counted"] == 2


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    p, lines = _run(["--gpus", "2", "--launch-check"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and not lines


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_without_an_external_launcher():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    p, lines = _run(["--gpus", "2", "--workload", "c1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-single-lane"],
                    {"MUSED_DIST_BACKEND": "gloo", "MUSED_FORCE_DEVICE": "0"}, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["rccl_ranks"] == 2 and res["config"]["collective_backend"] == "gloo"
    assert res["value"] > 0
