"""bench.py's launcher contract, on CPU: `--gpus N` without a launcher environment starts N rank processes (distinct RANK, one shared
rendezvous) before any GPU call and relays rank 0's line; a WORLD_SIZE that disagrees with --gpus is an error; a failing rank fails the run.
`--dry-launch` makes the ranks report their environment instead of touching a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_gpus_flag_spawns_that_many_ranks():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    out = [json.loads(line) for line in p.stdout.splitlines() if line.strip()]
    assert len(out) == 1 and out[0]["rank"] == 0 and out[0]["world_size"] == 2 and out[0]["n_gpus"] == 2   # ONE line on stdout: rank 0's
    others = [json.loads(line.split("] ", 1)[1]) for line in p.stderr.splitlines() if line.startswith("[rank ")]
    ranks = sorted([out[0]["rank"]] + [o["rank"] for o in others])
    assert ranks == [0, 1]
    assert {o["master"] for o in others} == {out[0]["master"]} and out[0]["master"].startswith("127.0.0.1:")
    assert len({out[0]["pid"]} | {o["pid"] for o in others}) == 2      # fresh processes, not the parent


def test_four_ranks_and_local_rank():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--dry-launch", "--workload", "assoc_sharded"], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    others = [json.loads(line.split("] ", 1)[1]) for line in p.stderr.splitlines() if line.startswith("[rank ")]
    assert sorted(o["local_rank"] for o in others) == [1, 2, 3] and all(o["world_size"] == 4 for o in others)


def test_world_size_mismatch_is_an_error():
    env = _clean_env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-launch"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr and not p.stdout.strip()


def test_under_a_launcher_the_environment_decides():
    env = _clean_env()
    env.update(RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and not p.stdout.strip()            # only rank 0 prints to stdout
    assert json.loads(p.stderr.strip())["rank"] == 1


def test_failing_rank_fails_the_run():
    # without a GPU every rank fails at its first GPU call: the parent must report failure, not hang and not print a number
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1", "--cpu-frames", "0"], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and not p.stdout.strip()
