"""Deterministic synthetic workloads of BASELINE.json.

Counter-based PRNG, identical for the CPU oracle and the GPU run, no file I/O:
    r(i, k) = splitmix64(seed ^ (k << 56) ^ i),   seed = 0x52494F5F52530001
Object i *is* its dense row; its reference-side key is ObjectId("Obj", str(i)) -> "Obj.<i>"
(object_placement/local.rs:26-29).  Node j has the well-formed address
"10.<j>>16>.<(j>>8)&255>.<j&255>:5000" so service.rs:204-213 never trips.
"""
import numpy as np

SEED = 0x52494F5F52530001
NONE = 0xFFFFFFFF
CAP_INF = 0xFFFFFFFFFFFFFFFF
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def r(i, k, seed=SEED):
    """r(i,k) for an array (or scalar) of counters i and a stream id k."""
    i = np.asarray(i, dtype=np.uint64)
    return splitmix64(np.uint64(seed) ^ (np.uint64(k) << np.uint64(56)) ^ i)


def node_address(j):
    return "10.%d.%d.%d:5000" % (j >> 16, (j >> 8) & 255, j & 255)


def object_id(i, struct_name="Obj"):
    return (struct_name, str(int(i)))


DEFAULT_ROWS = {"c1": 1000, "c2": 1_000_000, "c3": 10_000_000, "c3w": 10_000_000, "c4": 100_000_000, "c4shard": 12_500_000}
_zipf_cdf_cache = {}


def zipf_loads(n, s=1.1, kmax=65536, stream=3, seed=SEED, start=0):
    """load[i] = 1 + floor(Zipf(s) truncated to [0, kmax-1]) by inverse-CDF table lookup from r(i,3)."""
    key = (s, kmax)
    if key not in _zipf_cdf_cache:
        w = np.arange(1, kmax + 1, dtype=np.float64) ** (-s)
        cdf = np.cumsum(w)
        cdf /= cdf[-1]
        _zipf_cdf_cache[key] = cdf
    cdf = _zipf_cdf_cache[key]
    u = (r(np.arange(start, start + n, dtype=np.uint64), stream, seed) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    k = np.searchsorted(cdf, u, side="right")
    return (np.minimum(k, kmax - 1) + 1).astype(np.uint32)


def affinity(n, m, stream=1, seed=SEED, start=0):
    return (r(np.arange(start, start + n, dtype=np.uint64), stream, seed) % np.uint64(m)).astype(np.uint32)


def warm_assign(n, m, stream=2, seed=SEED, start=0):
    return (r(np.arange(start, start + n, dtype=np.uint64), stream, seed) % np.uint64(m)).astype(np.uint32)


def uniform_cap(load, m, headroom=1.25):
    """cap[j] = ceil(headroom * sum(load) / m), the same for every node."""
    total = int(np.asarray(load, dtype=np.uint64).sum())
    num = int(round(headroom * 1000))  # exact integer arithmetic: ceil(total*num / (1000*m))
    c = -((-total * num) // (1000 * m))
    return np.full(m, c, dtype=np.uint64)


def config(name, scale=1.0, seed=SEED, start=0, n_override=None):
    """Return dict(n, m, load, aff, cur, cap, alive) for a BASELINE.json config.

    name: "c1" 1 000 x 4 (plumbing) | "c2" 1M x 256 uniform | "c3" 10M x 1 024 Zipf (cold) |
          "c3w" warm variant | "c4" 100M x 4 096 Zipf (cold; a row shard of it = n_override + start) |
          "c4shard" one 12.5M x 4 096 row shard of config 4.
    `scale` shrinks n (tests); `start` offsets the object counter (row shards).
    """
    if name == "c1":
        n, m = 1000, 4
    elif name == "c2":
        n, m = 1_000_000, 256
    elif name in ("c3", "c3w"):
        n, m = 10_000_000, 1024
    elif name == "c4":
        n, m = 100_000_000, 4096
    elif name == "c4shard":
        n, m = 12_500_000, 4096
    else:
        raise ValueError(name)
    n = int(n_override) if n_override is not None else max(1, int(n * scale))
    if name in ("c1", "c2"):
        load = np.ones(n, np.uint32)
    else:
        load = zipf_loads(n, seed=seed, start=start)
    aff = affinity(n, m, seed=seed, start=start)
    if name == "c1":
        cap = np.full(m, CAP_INF, np.uint64)
    else:
        cap = uniform_cap(load, m)
    cur = warm_assign(n, m, seed=seed, start=start) if name == "c3w" else np.full(n, NONE, np.uint32)
    alive = np.ones(m, np.uint8)
    return dict(name=name, n=n, m=m, load=load, aff=aff, cur=cur, cap=cap, alive=alive)


def churn_mask(m, tick, frac=0.10, seed=SEED):
    """alive[] for churn tick `tick` (config 5): frac of the nodes, chosen by r(tick*m + j, 4), are down."""
    k = max(1, int(m * frac))
    score = r(np.uint64(tick) * np.uint64(m) + np.arange(m, dtype=np.uint64), 4, seed)
    dead = np.argsort(score, kind="stable")[:k]
    alive = np.ones(m, np.uint8)
    alive[dead] = 0
    return alive


# ---- the two tables on which CAPACITY BINDS (the solver's bin-packing part runs: cuts, water-fill, unplaced rows) ----

def contended_cap(cfg, factor=0.72):
    """Config 3 with 0.72 x its capacities: 0.9 x the total load fits, every node is cut, ~10 % of the rows go on to the
    water-fill and most of those stay unplaced (service.rs:244-252 has no such limit: this is the solver's own rule)."""
    return (cfg["cap"].astype(np.float64) * factor).astype(np.uint64)


def skew_affinity(n, m, shape=1.1, stream=5, seed=SEED, start=0):
    """Affinities from a Lomax(shape) tail by inverse CDF over r(i, 5), clipped to the last node: most objects ask for
    the first few servers, which are cut within their first claimants; ~94 % of the rows are water-filled elsewhere."""
    u = (r(np.arange(start, start + n, dtype=np.uint64), stream, seed) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    x = np.power(1.0 - u, -1.0 / shape) - 1.0
    return np.minimum(x, float(m - 1)).astype(np.int64).astype(np.uint32)
