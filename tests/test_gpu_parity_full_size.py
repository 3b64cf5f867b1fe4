"""Oracle parity at the sizes BASELINE.json names ("what's missing" #1): the HIP path through the C ABI
against the CPU oracle, bit for bit, on configs 2-5 at their STATED sizes — not just properties:

  config 2   1 000 000 objects x 256 nodes, load 1, uniform capacities               (every fix-up variant)
  config 3   10 000 000 x 1 024, Zipf(1.1), cap 1.25x: cold (all pending) and warm (all placed)
  config 5   the config-3 table, warm, 6 committed churn ticks (10 % of the nodes down per tick): assignment column,
             `used` and every counter after EVERY tick
  config 4   100 000 000 x 4 096 on one GPU, and as 8 row shards (8 handles on the one device, the protocol the
             multi-GPU bench runs): assignment column, `used`, counters
The oracle (oracle/placement_oracle.c, orc_tick) does 10 M rows in ~25 ms and 100 M in ~0.3 s; what costs time here is
generating the synthetic tables on the host (config 4: ~30 s), which is why they are cached per module."""
import functools

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
NONE = 0xFFFFFFFF


@pytest.fixture(scope="module")
def gp():
    import rio_gp
    rio_gp.build()
    return rio_gp


@functools.lru_cache(maxsize=4)
def cfg_of(name):
    return synth.config(name)


def _mk(gp, cfg, cur=None, rounds=2, lab=False):
    g = gp.GpuPlacement(cfg["n"], cfg["m"], spill_rounds=rounds, lab=lab)
    g.set_nodes(cfg["cap"], cfg["alive"])
    g.set_objects(cfg["n"], cfg["load"], cfg["aff"])
    if cur is not None:
        g.set_assign(cur)
    return g


def _same(g, oracle, cur, cfg, alive=None, rounds=2, commit=True):
    alive = cfg["alive"] if alive is None else alive
    want, used, ost = oracle.tick(cur, cfg["load"], cfg["aff"], cfg["cap"], alive, rounds)
    st = g.solve()
    got = g.get_solved()
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
    assert st == ost
    if commit:
        g.commit()
        assert np.array_equal(g.get_nodes()[2], used)
    return want, used, ost


def test_config2_exact_every_fixup_variant(gp, oracle):
    """1 M x 256, load 1: cap = ceil(1.25 * 1e6 / 256) = 4 883; cold — every claimant fits (fast path) — and with the
    capacity squeezed to 3 000 per node (every node cut, 232 000 rows water-filled or left unplaced), in every
    implementation of the fix-up."""
    cfg = cfg_of("c2")
    assert cfg["n"] == 1_000_000 and cfg["m"] == 256 and int(cfg["cap"][0]) == 4883 and int(cfg["load"].max()) == 1
    g = _mk(gp, cfg)
    _, _, st = _same(g, oracle, cfg["cur"], cfg)
    assert st["slow_path"] == 0 and st["claimed"] == cfg["n"]
    g.close()
    tight = dict(cfg, cap=np.full(256, 3000, np.uint64))
    from test_gpu_parity import FIXUP_VARIANTS, apply_variant
    for compact, spec in ((None, None),) + FIXUP_VARIANTS:
        g = _mk(gp, tight, lab=compact is not None)   # (None: the product library left to itself)
        if compact is not None:
            apply_variant(g, (compact, spec))
        _, _, st = _same(g, oracle, tight["cur"], tight)
        assert st["cut_nodes"] == 256 and st["unplaced"] > 0, (compact, spec)
        g.close()


def test_config3_cold_and_warm_at_10m(gp, oracle):
    cfg = cfg_of("c3")
    assert cfg["n"] == 10_000_000 and cfg["m"] == 1024
    g = _mk(gp, cfg)
    want, used, st = _same(g, oracle, cfg["cur"], cfg)            # cold: all pending, every claim fits
    assert st["slow_path"] == 0 and st["claimed"] == cfg["n"]
    _, _, st2 = _same(g, oracle, want, cfg)                         # second tick over the committed table: all kept
    assert st2["kept"] == cfg["n"] and st2["slow_path"] == 0
    g.close()
    warm = synth.warm_assign(cfg["n"], cfg["m"])                    # warm start somewhere else than the affinity
    g = _mk(gp, cfg, cur=warm)
    _, _, st3 = _same(g, oracle, warm, cfg)
    assert st3["kept"] == cfg["n"]
    g.close()


def test_config3_contended_at_10m(gp, oracle):
    """The headline table with 0.9x of the load as capacity: ~1 020 cut nodes, ~1 M rows water-filled or unplaced."""
    cfg = cfg_of("c3")
    tight = dict(cfg, cap=(cfg["cap"].astype(np.float64) * 0.72).astype(np.uint64))
    g = _mk(gp, tight, lab=True)
    _, _, st = _same(g, oracle, tight["cur"], tight, commit=False)
    assert st["cut_nodes"] > 900 and st["unplaced"] > 0
    # the same table again: the first solve sent ~10 % of the rows to the water-fill, so this one packs them at the cut pass
    _, _, st2 = _same(g, oracle, tight["cur"], tight, commit=False)
    assert st2 == st
    g.set_compact("auto", cut_pack="never")
    _, _, st3 = _same(g, oracle, tight["cur"], tight, commit=False)
    assert st3 == st
    # the two-pass form of the whole-table fix-up (k_cut_find, then the re-marking inside round 0), packing and not
    for cp in ("always", "never"):
        g.set_compact("auto", cut_pack=cp, cut_apply="never")
        _, _, st4 = _same(g, oracle, tight["cur"], tight, commit=False)
        assert st4 == st, cp
    g.set_compact("auto")
    _, _, st5 = _same(g, oracle, tight["cur"], tight)
    assert st5 == st
    g.close()


def test_config3_skew_at_10m(gp, oracle):
    """The headline table with Lomax(1.1) affinities: two dozen servers asked for by most objects are cut within their first
    claimants, ~94 % of the rows are water-filled elsewhere (bench.py's config3_skew record)."""
    cfg = cfg_of("c3")
    skew = dict(cfg, aff=synth.skew_affinity(cfg["n"], cfg["m"]))
    g = _mk(gp, skew, lab=True)
    _, _, st = _same(g, oracle, skew["cur"], skew, commit=False)
    assert 10 < st["cut_nodes"] < 100 and st["spilled"] > 9_000_000
    g.set_compact("auto", cut_apply="never")
    _, _, st2 = _same(g, oracle, skew["cur"], skew, commit=False)
    assert st2 == st
    g.set_compact("auto", cut_apply="always")    # (the one-pass form over a table where every wave range has work)
    _, _, st2b = _same(g, oracle, skew["cur"], skew, commit=False)
    assert st2b == st
    g.set_compact("auto", cut_pack="always")     # (packs 94 % of the table: what the adaptive rule never picks here)
    _, _, st3 = _same(g, oracle, skew["cur"], skew)
    assert st3 == st
    g.close()


@pytest.mark.parametrize("compact", [None, "never", "inc-never"])
def test_config5_churn_ticks_at_10m(gp, oracle, compact):
    """Six committed ticks of the config-5 stream at full size, every tick compared: ~1 M rows evicted and re-placed per
    tick, hundreds of cut nodes, the packed fix-up from the second tick on (product library) or the whole-table fix-up
    (lab build, packed fix-up switched off).  The product library scans committed ticks in place (k_inc_scan + k_rebal); the
    lab build also runs the stream with the in-place scan switched off."""
    cfg = cfg_of("c3")
    n, m = cfg["n"], cfg["m"]
    ref = synth.warm_assign(n, m)
    g = _mk(gp, cfg, cur=ref, lab=compact is not None)   # None: the product library, adaptive
    if compact == "inc-never":   # the packed fix-up behind k_scan<COMPACT>: round 3's tick
        g.set_compact("auto", inc="never")
    elif compact is not None:
        g.set_compact(compact)
    for tick in range(6):
        alive = synth.churn_mask(m, 2 + tick)
        g.set_alive_all(alive)
        st = g.tick()
        ref, used, ost = oracle.tick(ref, cfg["load"], cfg["aff"], cfg["cap"], alive, 2)
        assert st == ost, (tick, st, ost)
        assert np.array_equal(g.get_assign(), ref), tick
        assert np.array_equal(g.get_nodes()[2], used), tick
        assert tick == 0 or (ost["slow_path"] == 1 and ost["evicted"] > 500_000)
    g.close()


def test_config4_one_gpu_at_100m(gp, oracle):
    cfg = cfg_of("c4")
    assert cfg["n"] == 100_000_000 and cfg["m"] == 4096
    g = _mk(gp, cfg)
    _, _, st = _same(g, oracle, cfg["cur"], cfg)
    assert st["claimed"] + st["spilled"] + st["unplaced"] == cfg["n"]
    # and one churn tick at that size (10 % of 4 096 nodes down): the fix-up path over 100 M rows
    alive = synth.churn_mask(cfg["m"], 3)
    g.set_alive_all(alive)
    cur = g.get_assign()
    want, used, ost = oracle.tick(cur, cfg["load"], cfg["aff"], cfg["cap"], alive, 2)
    st = g.tick()
    assert st == ost and ost["evicted"] > 5_000_000
    assert np.array_equal(g.get_assign(), want)
    assert np.array_equal(g.get_nodes()[2], used)
    g.close()
    # the same size through the in-place scan + k_rebal (the product keeps k_scan<COMPACT> on tables whose blocks are too big
    # for the in-resolve cut search; the lab build's "always" runs it there too: the cuts of the balanced rows are then found
    # by k_cut_find over the packed table)
    gl = _mk(gp, cfg, cur=want, lab=True)
    gl.set_compact("always", inc="always")
    alive2 = synth.churn_mask(cfg["m"], 4)
    gl.set_alive_all(alive2)
    want2, used2, ost2 = oracle.tick(want, cfg["load"], cfg["aff"], cfg["cap"], alive2, 2)
    assert gl.tick() == ost2
    assert np.array_equal(gl.get_assign(), want2) and np.array_equal(gl.get_nodes()[2], used2)
    gl.close()


def test_config4_as_8_row_shards(gp, oracle):
    """North-star config 4 in its sharded form: 8 shards of 12.5 M rows (8 handles on the one device, LocalExchange —
    the same ShardedSolver and the same shard kernels bench.py --gpus 8 runs), equal to the whole-table oracle."""
    import sharded
    from test_gpu_sharded import make_engines
    cfg = cfg_of("c4")
    case = (cfg["cur"], cfg["load"], cfg["aff"], cfg["cap"], cfg["alive"])
    bounds = sharded.shard_bounds(cfg["n"], 8)
    assert bounds[1] - bounds[0] == 12_500_000
    engines = make_engines(gp, case, bounds)
    sol = sharded.ShardedSolver(engines, sharded.LocalExchange(8))
    want, used, ost = oracle.tick(*case, 2)
    st = sol.solve()
    for e, lo, hi in zip(engines, bounds[:-1], bounds[1:]):
        assert np.array_equal(e.g.get_solved(), want[lo:hi]), lo
    assert st == ost
    sol.commit()
    for e in engines:
        assert np.array_equal(e.g.get_nodes()[2], used)
    # squeeze the capacities: cuts on every rank, forced nodes on the upper ranks, water-fill across shards
    cap = (cfg["cap"] * np.uint64(72)) // np.uint64(100)   # 0.9 x the load
    for e in engines:
        e.g.set_nodes(cap, cfg["alive"])
        e.g.set_assign(np.full(e.g.num_objects, NONE, np.uint32))
    want2, used2, ost2 = oracle.tick(cfg["cur"], cfg["load"], cfg["aff"], cap, cfg["alive"], 2)
    st2 = sol.solve()
    assert st2 == ost2 and ost2["cut_nodes"] > 1000
    for e, lo, hi in zip(engines, bounds[:-1], bounds[1:]):
        assert np.array_equal(e.g.get_solved(), want2[lo:hi]), lo
    for e in engines:
        e.g.close()
