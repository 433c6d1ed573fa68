"""bench.py --gpus N without a launcher must start N ranks itself.  CPU rehearsal: the same
launch path with a gloo group; the printed n_gpus must be the number of ranks that really joined."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=280)


def test_gpus_2_spawns_two_ranks():
    r = _run(["--gpus", "2", "--selftest-launcher"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                       # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2


def test_world_size_mismatch_aborts():
    # a launcher that started 1 rank for --gpus 2 must not produce a line that says n_gpus 1
    r = _run(["--gpus", "2", "--selftest-launcher"], {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
                                                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29917"})
    assert r.returncode != 0
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
