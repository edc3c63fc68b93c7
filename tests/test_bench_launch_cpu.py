"""bench.py launches its own ranks: `python bench.py --gpus N` from a plain environment (no RANK / WORLD_SIZE) must start
N processes through torch.distributed.run BEFORE any GPU call, and rank 0 must print ONE JSON line with n_gpus == N.
Runs on CPU: --launch-check does the rendezvous (gloo) and nothing else."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plain_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_bench_gpus2_self_launches_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                       env=_plain_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_met"] == 2 and sorted(out["devices"]) == [0, 1]


def test_bench_gpus_mismatch_is_refused():
    """a rank environment that disagrees with --gpus must fail loudly instead of measuring fewer GPUs than asked for"""
    env = _plain_env()
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_bench_single_rank_launch_check_unchanged_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launch-check"], env=_plain_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1
