"""bench.py --gpus N cannot print a line for fewer GPUs than it names (round-4 verdict, item 2): started without a launcher it
becomes the launcher; with fewer visible devices than N it refuses with exit code 2 and an empty stdout."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_more_gpus_than_are_visible():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode == 2 and r.stdout.strip() == "" and "refusing to run" in r.stderr


def test_bench_refuses_a_launcher_with_another_rank_count():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode != 0 and r.stdout.strip() == "" and "must equal WORLD_SIZE" in r.stderr
