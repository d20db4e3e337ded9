"""bench.py's launcher contract, checked without a GPU: `--gpus N` with no WORLD_SIZE self-spawns N ranks (gloo dry run),
a --gpus / WORLD_SIZE mismatch and a missing device are loud errors, and nothing under tests/ is imported by bench.py."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_gpus_2_self_spawns_two_ranks():
    r = _run(["--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["ranks_counted"] == 2 and out["dry_run"] is True


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_no_device_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        return
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


def test_bench_does_not_import_tests():
    src = open(os.path.join(REPO, "bench.py")).read()
    assert "tests" not in [p.strip("\"' ") for p in src.replace("(", " ").replace(")", " ").replace(",", " ").split()
                           if p.strip("\"' ") == "tests"]
    assert "from util import" not in src and "import util" not in src
