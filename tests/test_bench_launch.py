"""bench.py's launch path on CPU: the driver's literal command `python bench.py --gpus N` has to produce N ranks by itself
(VERDICT round 2, item 1).  --rank-check-only stops after the rendezvous (gloo), so no GPU is needed."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra=None, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, env=env, timeout=timeout)


def last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text
    return json.loads(lines[-1])


@pytest.mark.parametrize("n", [1, 2])
def test_gpus_argument_creates_that_many_ranks(n):
    r = run_bench(["--gpus", str(n), "--rank-check-only"])
    assert r.returncode == 0, r.stderr
    line = last_json(r.stdout)
    assert line["n_gpus"] == n and line["ranks_seen"] == n
    assert sum(l.startswith("{") for l in r.stdout.splitlines()) == 1          # ONE JSON line, from rank 0


def test_gpus_argument_must_match_the_launcher():
    r = run_bench(["--gpus", "4", "--rank-check-only"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_more_gpus_than_the_node_has_fails_loudly():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = run_bench(["--gpus", str(have + 1 if have else 2), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "visible GPU" in r.stderr
