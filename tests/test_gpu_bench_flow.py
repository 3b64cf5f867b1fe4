"""bench.py's launch contract, exercised on the GPU box: the N=1 line, and the N=2 flow exactly as the driver
starts it (python -m torch.distributed.run ... bench.py --gpus 2) — with both ranks on the one GPU a gpurun box
has (gloo control plane, peer-to-peer windows between the two processes).  stdout must be ONE JSON line."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _one_json_line(out):
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, lines[:5]
    return json.loads(lines[0])


def test_bench_line_n1():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--workload", "c2",
                        "--cpu-sample", "20000"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["unit"] == "decisions/s" and d["value"] > 1e9
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert d["stats_last_step"]["claimed"] + d["stats_last_step"]["spilled"] + d["stats_last_step"]["unplaced"] == 1000000


@pytest.mark.parametrize("exchange", ["p2p", "torch"])
def test_bench_two_ranks_as_the_driver_launches_it(exchange):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2",
           "--workload", "c2", "--objects", "500000", "--no-cpu-baseline", "--backend", "gloo", "--same-device", "--exchange", exchange]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["objects_per_gpu"] == 500000
    assert d["stats_last_step"]["n_objects"] == 1000000      # global counters: both shards
    assert d["config"]["slow_path_steps"] == 0
    # gloo cannot all-gather device tensors, so the torch path is expected to drop... to nothing: it IS the last rung
    assert d["config"]["exchange"] in ("p2p", "torch")


def test_bench_eight_ranks_on_one_gpu():
    """The driver's N=8 command line, all eight ranks on the one GPU of the box: eight processes exchanging through each
    other's IPC-mapped windows, global capacities, one JSON line.  Workload c2 (256 nodes): ranks that SHARE a GPU must be
    co-resident, and eight spinning exchange kernels at 1 024 nodes would fill the chip."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "3",
           "--workload", "c2", "--objects", "1000000", "--no-cpu-baseline", "--backend", "gloo", "--same-device"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 8 and d["config"]["exchange"] == "p2p", d["config"]
    st = d["stats_last_step"]
    assert st["n_objects"] == 8_000_000 and st["claimed"] == 8_000_000 and d["config"]["slow_path_steps"] == 0


def test_bench_config5_churn_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c5", "--steps", "12", "--warmup", "3",
                        "--objects", "1000000"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 1e8 and d["objects_moved_per_s"] > 0
    st = d["stats_last_step"]
    assert st["kept"] + st["claimed"] + st["spilled"] + st["unplaced"] == 1000000 and st["evicted"] > 0

