"""bench.py --gpus N without a launcher must start N ranks itself.  CPU rehearsal: the same
launch path with a gloo group; the printed n_gpus must be the number of ranks that really joined."""
import json
import os
import subprocess
import sys

import pytest

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
    # the fixed-global-batch record (global 1024, strong scaling) shards 512 rows to each of the two ranks, tiling the batch
    fg = out["fixed_global"]
    assert fg["global_batch"] == 1024 and fg["per_gpu_batch"] == 512 and sorted(map(tuple, fg["rows_by_rank"])) == [(0, 512), (512, 1024)]


def test_world_size_mismatch_aborts():
    # a launcher that started 1 rank for --gpus 2 must not produce a line that says n_gpus 1
    r = _run(["--gpus", "2", "--selftest-launcher"], {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
                                                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29917"})
    assert r.returncode != 0
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_one_rank_rccl_line_carries_the_comm_record():
    """The N > 1 path of bench.py on the one GPU of the test box: a 1-rank RCCL group (S2VT_BENCH_PG=1) runs the overlapped
    all-reduce code, and the JSON line must say how long each gradient group's collective took and how much of it was exposed
    after the backward - the fields the first 8-GPU run will be read by."""
    r = _run(["--gpus", "1", "--steps", "3", "--warmup", "1", "--headline-only"],
             {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29931",
              "S2VT_BENCH_PG": "1", "S2VT_PERSIST": "1", "S2VT_PIPE_BLOCK": "32"})       # (the options the asserted overlap rests on)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert out["rccl_ranks"] == 1 and out["n_gpus"] == 1
    c = out["comm"]
    assert len(c["group_ms"]) == 3 and all(x >= 0 for x in c["group_ms"]) and len(c["bytes"]) == 3
    assert sum(c["bytes"]) == 4 * 48125000                              # the whole flat gradient buffer
    assert c["exposed_after_backward_ms"] >= 0 and len(c["exposed_after_backward_ms_by_rank"]) == 1
    assert c["group_start_after_backward_end_ms"][0] < 0               # out_linear's all-reduce starts under the backward
    assert out["ms_per_step_by_rank"]["min"] <= out["ms_per_step_by_rank"]["max"]
