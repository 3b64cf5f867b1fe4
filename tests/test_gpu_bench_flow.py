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
    # the headline is a stream of committed ticks: by the last one every row is kept where the first tick put it ...
    last = d["stats_last_step"]
    assert last["kept"] + last["claimed"] + last["spilled"] + last["unplaced"] == 1000000 and last["kept"] > 0
    assert d["committed_tick_frac"] > 0 and d["roofline"]["frac_committed_tick"] == d["committed_tick_frac"]
    # ... and the cold table, re-solved without committing, is all pending rows
    cold = d["stats_cold_step"]
    assert cold["claimed"] + cold["spilled"] + cold["unplaced"] == 1000000 and d["cold_resolve_uncommitted"]["value"] > 1e9
    # the run checks itself against the oracle (and would have exited with rc 3 on a mismatch)
    assert d["parity"]["equal"] is True and d["parity"]["checked_rows"] == 1000000
    assert d["dependent_tick_ms"] > 0 and "traffic_source" in d["roofline"]
    # ... and against the string-level restatement of the reference: the map the cpu_baseline leg built, read back
    port = d["parity"]["against_reference_port"]
    assert port["equal"] is True and port["rows"] == 20000
    assert d["roofline"]["gpu_ms_per_step_events"] > 0


def test_bench_headline_line_has_parity_traffic_and_config4():
    """The default command line of the driver at N=1, shortened: config 3 at full size with the in-run parity check, the
    in-run PMC passes and the config-4-on-one-GPU data point (100 M x 4 096, parity at size)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5", "--cpu-sample", "200000"],
                       capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_json_line(r.stdout)
    assert d["config"]["objects_per_gpu"] == 10_000_000 and d["config"]["nodes"] == 1024
    assert d["parity"]["equal"] is True and d["parity"]["checked_rows"] == 10_000_000
    # the reference restatement's own map (200 000 get_or_create_placement calls here, 10 M in the driver's run), read back
    assert d["parity"]["against_reference_port"]["equal"] is True and d["parity"]["against_reference_port"]["rows"] == 200_000
    # one comparable quantity for every N: the points of the strong (config 4) and weak (config 3) curves under the same keys
    sp = d["scaling_points"]
    assert sp["strong_config4_committed_tick"]["n_gpus"] == 1 and sp["strong_config4_committed_tick"]["rows_total"] == 100_000_000
    assert sp["strong_config4_committed_tick"]["value"] == d["config4_single_gpu"]["committed_tick"]["value"]
    assert sp["weak_config3_committed_tick"]["value"] == d["value"] and "committed ticks" in sp["weak_config3_committed_tick"]["definition"]
    # the kernel the roofline is quoted on, and the honest denominators next to it: the DRAM-bound form of the same step,
    # the committed tick (what `value` is), the dependent tick
    rf = d["roofline"]
    # (a chained committed tick costs less than k_scan's own launch: consecutive scans overlap their ramp-up and tail — so the
    #  tick's fraction may lie above the kernel's; both stay below the chip)
    assert rf["frac"] > 0.5 and rf["frac_dram_bound"] > 0.4 and 0.2 < rf["frac_committed_tick"] < 0.95
    assert abs(d["value"] - 10_000_000 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]       # value IS the committed tick
    assert rf["frac_dependent_tick"] > 0.1 and d["cold_resolve_uncommitted"]["slow_path_steps"] == 0
    c5 = d["config5_churn"]
    assert c5["parity"]["equal"] is True and c5["pipelined"]["equal_to_synchronous_stream"] is True and c5["slow_path_ticks"] > 0
    c2 = d["config2"]
    assert c2["parity"]["equal"] is True and c2["committed_tick"]["value"] > 1e9
    if d["roofline"]["traffic"] is not None:   # rocprofv3 present: measured in this run, not replayed
        assert d["roofline"]["traffic_source"].startswith("measured in this run")
        assert 0.9 < d["roofline"]["traffic"] / d["roofline"]["algorithmic_bytes_per_launch"] < 1.3
    c4 = d["config4_single_gpu"]
    assert c4["parity"]["equal"] is True and c4["parity"]["checked_rows"] == 100_000_000
    assert c4["cold_resolve_uncommitted"]["slow_path_steps"] == 0 and c4["committed_tick"]["frac"] > 0.3
    ch4 = c4["churn_tick_pipelined"]    # config 5's churn on config 4's table, replayed by the oracle at 100 M rows
    assert ch4["parity"]["equal"] is True and ch4["parity"]["ticks_replayed"] == 13 and ch4["stats_last_tick"]["slow_path"] == 1
    assert 0.2 < ch4["ms_per_tick"] < 3.0
    cold = d["roofline"]["beyond_infinity_cache"]
    assert cold["rows"] == 40_000_000 and cold["whole_step_frac"] > 0.4 and cold["committed_tick_frac"] > 0.3


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


def test_bench_gpus_2_without_a_launcher_spawns_its_own_ranks():
    """`python3 bench.py --gpus 2 ...` started the way the driver starts the N = 1 line (no torchrun, no WORLD_SIZE) must not
    print a one-GPU line under another name: it starts the two ranks itself and prints ONE line with n_gpus = 2 — and without
    --same-device on this one-GPU box it refuses loudly instead."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--workload", "c2",
           "--objects", "300000", "--no-cpu-baseline", "--backend", "gloo", "--same-device"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["objects_per_gpu"] == 300000 and d["stats_last_step"]["n_objects"] == 600000
    assert [x["rank"] for x in d["config"]["ranks"]] == [0, 1]
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([c for c in cmd if c != "--same-device"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
        assert r.returncode == 2 and r.stdout.strip() == "" and "refusing to run" in r.stderr


def _check_scaling_fields(d, world, exchange):
    """What every N > 1 line says about itself, in the same words at every N: `value` = committed ticks of the ONE config-4
    table (the definition string is the N = 1 line's), who ran where, and what RCCL was asked for."""
    import bench
    assert d["config"]["value_is"] == bench.DEF_STRONG and d["weak_config3"]["value_is"] == bench.DEF_WEAK
    sp = d["scaling_points"]
    assert sp["strong_config4_committed_tick"]["value"] == d["value"] and sp["strong_config4_committed_tick"]["n_gpus"] == world
    assert sp["strong_config4_committed_tick"]["definition"] == bench.DEF_STRONG
    assert sp["weak_config3_committed_tick"]["value"] == d["weak_config3"]["value"]
    assert sp["weak_config3_committed_tick"]["definition"] == bench.DEF_WEAK
    last = d["stats_last_step"]      # a committed stream: by the timed ticks every row is kept where tick 1 put it
    assert last["kept"] == last["n_objects"] == d["config"]["objects_total"] and last["slow_path"] == 0
    cold = d["cold_resolve_uncommitted"]   # ... and rounds 2-3's quantity next to it, under its own name
    cs = cold["stats_last_step"]           # (the cold table: every row pending)
    assert cs["kept"] == 0 and cs["claimed"] + cs["spilled"] + cs["unplaced"] == d["config"]["objects_total"] and cold["value"] > 0
    ranks = d["config"]["ranks"]
    assert [r["rank"] for r in ranks] == list(range(world)) and all(r["exchange"] == exchange for r in ranks)
    assert all(r["device"] == 0 for r in ranks)          # --same-device
    rc = d["config"]["rccl"]
    assert rc["control_plane_backend"] == "gloo" and rc["control_plane_communicators"] == 0
    assert rc["data_path_communicators"] == (1 if exchange == "native" else 0)
    assert rc["data_path_comm_ranks"] == [world if exchange == "native" else 0] * world
    assert "peer_access" in d["config"] and ("tick_async" in d["config"]["tick"]) == (exchange == "p2p")


def test_bench_ladder_one_rank_fails_p2p_every_rank_lands_on_the_same_rung():
    """The exchange ladder p2p -> native -> torch is agreed by ALL ranks: here the p2p rung is made to fail on rank 1 only,
    after the windows are mapped (its peers are already polling for its words: they time out inside their kernels), and every
    rank must drop it together and land on the same next rung that works — on this box, where both processes share the one
    GPU, RCCL refuses the native rung on every rank too, so that is `torch` (over gloo); with a GPU per rank it is `native`."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--total-objects", "600000", "--no-weak", "--no-sharded-churn", "--no-cpu-baseline", "--backend", "gloo", "--same-device"]
    env = dict(os.environ, RIO_GP_BENCH_FAIL_RUNG="p2p", RIO_GP_BENCH_FAIL_RANK="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_json_line(r.stdout)
    ladder = d["config"]["exchange_ladder"]
    assert ladder[0]["path"] == "p2p" and ladder[0]["ok"] is False
    assert d["config"]["exchange"] in ("native", "torch") and ladder[-1]["ok"] is True and ladder[-1]["path"] == d["config"]["exchange"]
    assert all(x["exchange"] == d["config"]["exchange"] for x in d["config"]["ranks"])
    assert d["parity"]["equal"] is True and d["parity"]["checked_rows"] == 600_000


@pytest.mark.parametrize("exchange", ["p2p", "torch"])
def test_bench_config4_strong_scaling_two_ranks(exchange):
    """north_star's config 4 is what `bench.py --gpus N` measures for N > 1: ONE table split over the ranks (strong
    scaling), here 2 M x 4 096 over two ranks that share the one GPU — through the peer-to-peer windows (the default
    first rung: ranks that find each other on one device at the handshake bound their exchange kernels' grids, so 512
    node groups a rank no longer have to be co-resident workgroup by workgroup) and through the collective path (torch /
    gloo).  The weak-scaled config 3 is the second measurement of the same run, and both check themselves against the
    whole-table oracle."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2",
           "--total-objects", "2000000", "--objects", "300000", "--backend", "gloo", "--same-device", "--exchange", exchange]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["objects_total"] == 2_000_000 and d["config"]["objects_per_gpu"] == 1_000_000 and d["config"]["nodes"] == 4096
    assert d["config"]["workload"].startswith("config 4:")
    assert d["config"]["exchange"] == exchange and d["config"]["exchange_ladder"][0]["ok"] is True
    assert d["parity"]["equal"] is True and d["parity"]["checked_rows"] == 2_000_000
    _check_scaling_fields(d, 2, exchange)
    w = d["weak_config3"]
    assert w["scaling"] == "weak" and w["objects_per_gpu"] == 300000 and w["nodes"] == 1024 and w["parity"]["equal"] is True
    # the sharded table under committed ticks: churn-free (every row kept, fast path) and config 5's churn (10 % of the
    # nodes flip per tick: every tick takes the fix-up exchanges), the final table against the chained oracle
    for rec in (d["committed_ticks"], w["committed_ticks"]):
        assert "error" not in rec, rec
        assert rec["committed_tick_no_churn"]["slow_path_ticks"] == 0 and rec["committed_tick_no_churn"]["value"] > 0
        ch = rec["churn"]
        assert ch["slow_path_ticks"] == rec["ticks"] and ch["objects_moved_per_s"] > 0 and ch["stats_last_tick"]["evicted"] > 0
        if exchange == "p2p":   # the asynchronous forms over the windows: nothing waits on the host between ticks
            assert rec["committed_tick_no_churn_async"]["slow_path_ticks"] == 0
            cha = rec["churn_async"]
            assert cha["slow_path_ticks"] == rec["ticks"] and cha["stats_last_tick"]["evicted"] > 0
            assert cha["ms_per_tick"] > 0   # (no comparison of the two: both ranks time-slice ONE GPU here, the order flips by run)
        assert rec["parity"]["equal"] is True, rec["parity"]


def test_bench_config4_eight_ranks_p2p_on_one_gpu():
    """Config 4's node count (4 096) over EIGHT ranks on the one GPU, the driver's command line with its default exchange:
    eight exchange kernels of 512 node groups each, kept to 64 workgroups a rank by the co-residency rule."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "6", "--warmup", "2",
           "--total-objects", "4000000", "--objects", "200000", "--no-cpu-baseline", "--backend", "gloo", "--same-device"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["nodes"] == 4096
    assert d["config"]["exchange"] == "p2p" and d["config"]["exchange_ladder"][0]["ok"] is True, d["config"]
    assert d["parity"]["equal"] is True and d["parity"]["checked_rows"] == 4_000_000
    assert d["weak_config3"]["parity"]["equal"] is True
    _check_scaling_fields(d, 8, "p2p")
    # the committed / churn tick streams, synchronous and asynchronous, between eight processes: every tick of a quiet
    # asynchronous stream lands in the next window slot (a rank that is through a tick must not overwrite what a slower one
    # has not read — found exactly here, with a tick that took four sequence numbers and four slots)
    for rec in (d["committed_ticks"], d["weak_config3"]["committed_ticks"]):
        assert "error" not in rec, rec
        assert rec["parity"]["equal"] is True and rec["churn_async"]["slow_path_ticks"] == rec["ticks"]


def test_bench_eight_ranks_on_one_gpu():
    """The driver's N=8 command line, all eight ranks on the one GPU of the box: eight processes exchanging through each
    other's IPC-mapped windows, global capacities, one JSON line (workload c2, 256 nodes)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "3",
           "--workload", "c2", "--objects", "1000000", "--no-cpu-baseline", "--backend", "gloo", "--same-device"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 8 and d["config"]["exchange"] == "p2p", d["config"]
    st = d["stats_last_step"]
    assert st["n_objects"] == 8_000_000 and st["kept"] == 8_000_000 and d["config"]["slow_path_steps"] == 0   # committed ticks
    assert d["cold_resolve_uncommitted"]["stats_last_step"]["claimed"] == 8_000_000


def test_bench_config5_churn_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c5", "--steps", "12", "--warmup", "3",
                        "--objects", "1000000"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_json_line(r.stdout)
    rec = d["config5_churn"]
    assert d["n_gpus"] == 1 and d["value"] > 1e8 and rec["objects_moved_per_s"] > 0 and d["value"] == rec["pipelined"]["value"]
    st = rec["stats_last_tick"]
    assert st["kept"] + st["claimed"] + st["spilled"] + st["unplaced"] == 1000000 and st["evicted"] > 0
    assert d["parity"]["equal"] is True and rec["pipelined"]["equal_to_synchronous_stream"] is True

