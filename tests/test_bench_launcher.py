"""CPU tests of `bench.py --gpus N`: started WITHOUT a launcher it must spawn its own N rank processes (fresh children,
the parent never loads the library or touches a device), rendezvous over gloo, aggregate, and print ONE JSON line; under
torch.distributed.run it must use the ranks it is given.  --dry-run keeps the device out of it (value = null)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _json_lines(text):
    return [json.loads(ln) for ln in text.splitlines() if ln.startswith("{")]


@pytest.mark.parametrize("n,workload", [(2, "frames"), (3, "host-stream"), (2, "tiled8k")])
def test_self_spawned_ranks(n, workload):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "2", "--warmup", "1", "--dry-run", "--workload", workload],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == n and line["dry_run"] is True and line["value"] is None
    assert line["ranks_seen"] == list(range(n)) and line["distinct_pids"] == n
    assert line["max_over_ranks_check"] == float(n)            # the max over ranks really saw the last rank
    assert line["config"]["workload"] == workload


def test_single_rank_does_not_spawn():
    r = subprocess.run([sys.executable, BENCH, "--dry-run"], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    line, = _json_lines(r.stdout)
    assert line["n_gpus"] == 1 and line["parent_pid"] == os.getpid()      # the bench process itself is the rank


def test_under_torch_distributed_run():
    """The driver's documented launch line for N > 1."""
    from tests.test_multi_gpu_cpu import _free_port
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line, = _json_lines(r.stdout)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1]
    # ONE line on stdout means one line: what libraries print there (gloo's "[Gloo] Rank 0 is connected to 1 peer ranks")
    # must not sit next to the result
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout


def test_failed_rank_fails_the_launch():
    """A rank that dies must not leave the others (or the parent) hanging: rc != 0, promptly."""
    env = _clean_env()
    env["SRCNN_BENCH_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "rank 1 exited" in r.stderr


def test_launcher_parent_never_loads_the_library():
    """The parent must stay free of HIP: it may not import libsrcnn_amd (whose first call initialises the runtime)."""
    code = ("import sys, runpy; sys.argv=['bench.py','--gpus','2','--dry-run'];\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    assert e.code == 0, e.code\n"
            "assert 'libsrcnn_amd' not in sys.modules and 'torch' not in sys.modules, sorted(m for m in sys.modules if 'torch' in m or 'srcnn' in m)\n" % BENCH)
    r = subprocess.run([sys.executable, "-c", code], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
