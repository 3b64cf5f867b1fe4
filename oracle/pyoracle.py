"""ctypes front-end of the CPU oracle (oracle/liboracle_rio.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg — never by the product package.  See placement_oracle.h for what each
function restates (reference file:line) and for the parity status (map semantics + policy
pinned; capacity/spill "parity unpinned").
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_DIR, "liboracle_rio.so")
NONE = 0xFFFFFFFF
CAP_INF = 0xFFFFFFFFFFFFFFFF

u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


class Stats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in (
        "n_objects", "kept", "evicted", "claimed", "spilled", "unplaced",
        "load_kept", "load_claimed", "load_spilled", "load_unplaced")] + [
        (k, C.c_uint32) for k in ("cut_nodes", "slow_path", "rounds_run", "reserved")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "reserved"}


def build(force=False):
    """Compile the oracle with gcc/g++ (Makefile in this directory)."""
    srcs = [os.path.join(_DIR, f) for f in ("placement_oracle.c", "placement_oracle.h", "local_placement_oracle.cpp")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.run(["make", "-C", _DIR, "-s"], check=True)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_lookup_batch.argtypes = [u32p, C.c_uint64, u32p, C.c_uint64, u32p]
        L.orc_update_batch.argtypes = [u32p, C.c_uint64, C.c_uint32, u32p, u32p, C.c_uint64]
        L.orc_remove_batch.argtypes = [u32p, C.c_uint64, u32p, C.c_uint64]
        L.orc_clean_servers.argtypes = [u32p, C.c_uint64, u64p, C.c_uint32]
        L.orc_clean_servers.restype = C.c_uint64
        L.orc_recompute_used.argtypes = [u32p, u32p, C.c_uint64, C.c_uint32, u64p]
        L.orc_recompute_used.restype = None
        L.orc_tick.argtypes = [u32p, u32p, u32p, C.c_uint64, u64p, u8p, C.c_uint32, C.c_uint32, u32p, u64p,
                               C.POINTER(Stats)]
        L.orc_place_pending.argtypes = [u32p, u32p, C.c_uint64, u64p, u8p, u64p, C.c_uint32, C.c_uint32,
                                        u32p, u32p, C.c_uint64, u32p, u32p]
        L.orc_tick_ex.argtypes = [u32p, u32p, u32p, C.c_uint64, u64p, u8p, C.c_uint32, C.c_uint32, C.c_uint32, u32p, u64p,
                                  C.POINTER(Stats)]
        L.orc_place_pending_ex.argtypes = [u32p, u32p, C.c_uint64, u64p, u8p, u64p, C.c_uint32, C.c_uint32, C.c_uint32,
                                           u32p, u32p, C.c_uint64, u32p, u32p]
        L.orc_splitmix64.argtypes = [C.c_uint64]
        L.orc_splitmix64.restype = C.c_uint64
        # string layer
        for name in ("lpo_new", "lpo_clone", "mem_new"):
            getattr(L, name).restype = C.c_void_p
        L.lpo_clone.argtypes = [C.c_void_p]
        L.lpo_free.argtypes = [C.c_void_p]
        L.mem_free.argtypes = [C.c_void_p]
        L.lpo_update.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]
        L.lpo_lookup.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
        L.lpo_clean_server.argtypes = [C.c_void_p, C.c_char_p]
        L.lpo_remove.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.lpo_len.argtypes = [C.c_void_p]
        L.lpo_len.restype = C.c_uint64
        L.mem_push.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
        L.mem_set_is_active.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
        L.mem_is_active.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.lpo_get_or_create_placement.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                                  C.c_char_p, C.c_size_t]
        L.lpo_check_address_mismatch.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p]
        L.lpo_bench_policy.argtypes = [C.c_uint64, C.c_uint32, u32p, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        L.lpo_bench_policy.restype = C.c_double
        L.lpo_policy_readback.argtypes = [C.c_uint64, C.c_uint32, u32p, C.c_void_p, C.c_void_p, u32p]
        L.lpo_policy_readback.restype = C.c_double
        L.lpo_bench_clean_server.argtypes = [C.c_uint64, C.c_uint32, u32p, C.c_uint32]
        L.lpo_bench_clean_server.restype = C.c_double
        L.lpo_hardware_concurrency.restype = C.c_uint
        _lib = L
    return _lib


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


# ---- dense-index oracle -------------------------------------------------------------------

def lookup_batch(assign, idx):
    idx = _u32(idx)
    out = np.empty(len(idx), np.uint32)
    rc = lib().orc_lookup_batch(_u32(assign), len(assign), idx, len(idx), out)
    if rc:
        raise ValueError("orc_lookup_batch rc=%d" % rc)
    return out


def update_batch(assign, m, idx, node):
    """In place on `assign` (must be a C-contiguous uint32 array)."""
    idx, node = _u32(idx), _u32(node)
    return lib().orc_update_batch(assign, len(assign), m, idx, node, len(idx))


def remove_batch(assign, idx):
    idx = _u32(idx)
    return lib().orc_remove_batch(assign, len(assign), idx, len(idx))


def dead_bitmap(m, dead_nodes):
    bm = np.zeros((m + 63) // 64, np.uint64)
    for j in dead_nodes:
        bm[j >> 6] |= np.uint64(1) << np.uint64(j & 63)
    return bm


def clean_servers(assign, m, dead_nodes):
    return int(lib().orc_clean_servers(assign, len(assign), dead_bitmap(m, dead_nodes), m))


def recompute_used(assign, load, m):
    used = np.zeros(m, np.uint64)
    lib().orc_recompute_used(_u32(assign), _u32(load), len(assign), m, used)
    return used


REF_SELF_ASSIGN = 2   # ORC_REF_SELF_ASSIGN = RIO_GP_CFG_REF_SELF_ASSIGN


def tick(cur, load, aff, cap, alive, rounds=2, flags=0):
    """flags=REF_SELF_ASSIGN: a claim does not need an active node (service.rs:244-252 self-assigns unconditionally)."""
    cur, load, aff, cap, alive = _u32(cur), _u32(load), _u32(aff), _u64(cap), _u8(alive)
    n, m = len(cur), len(cap)
    nxt = np.empty(n, np.uint32)
    used = np.zeros(max(m, 1), np.uint64)
    st = Stats()
    rc = lib().orc_tick_ex(cur, load, aff, n, cap, alive, m, rounds, flags, nxt, used, C.byref(st))
    if rc:
        raise RuntimeError("orc_tick rc=%d" % rc)
    return nxt, used[:m], st.as_dict()


def place_pending(assign, load, cap, alive, used, idx, requester, rounds=2, flags=0):
    """In place on `assign` and `used`; returns (out_node, out_flag)."""
    idx, requester = _u32(idx), _u32(requester)
    out_node = np.empty(len(idx), np.uint32)
    out_flag = np.empty(len(idx), np.uint32)
    rc = lib().orc_place_pending_ex(assign, _u32(load), len(assign), _u64(cap), _u8(alive), used, len(cap), rounds, flags,
                                    idx, requester, len(idx), out_node, out_flag)
    if rc:
        raise ValueError("orc_place_pending rc=%d" % rc)
    return out_node, out_flag


# ---- string-level oracle (LocalObjectPlacement + policy) ------------------------------------

class LocalObjectPlacement:
    """local.rs:12-68 restated; `clone()` shares the map like the Arc does."""

    def __init__(self, _h=None):
        self._h = _h if _h is not None else lib().lpo_new()

    def clone(self):
        return LocalObjectPlacement(lib().lpo_clone(self._h))

    def prepare(self):  # mod.rs:42-44 default no-op
        return None

    def update(self, struct_name, object_id, server_address):
        a = None if server_address is None else server_address.encode()
        lib().lpo_update(self._h, struct_name.encode(), object_id.encode(), a)

    def lookup(self, struct_name, object_id):
        buf = C.create_string_buffer(512)
        if lib().lpo_lookup(self._h, struct_name.encode(), object_id.encode(), buf, 512):
            return buf.value.decode()
        return None

    def clean_server(self, address):
        lib().lpo_clean_server(self._h, address.encode())

    def remove(self, struct_name, object_id):
        lib().lpo_remove(self._h, struct_name.encode(), object_id.encode())

    def __len__(self):
        return int(lib().lpo_len(self._h))

    def __del__(self):
        try:
            lib().lpo_free(self._h)
        except Exception:
            pass


class LocalStorage:
    """cluster/storage/local.rs + the is_active default of cluster/storage/mod.rs:95-110."""

    def __init__(self):
        self._h = lib().mem_new()

    def push(self, ip, port, active=True):
        lib().mem_push(self._h, ip.encode(), str(port).encode(), int(active))

    def set_is_active(self, ip, port, active):
        lib().mem_set_is_active(self._h, ip.encode(), str(port).encode(), int(active))

    def is_active(self, ip, port):
        return bool(lib().mem_is_active(self._h, ip.encode(), str(port).encode()))

    def __del__(self):
        try:
            lib().mem_free(self._h)
        except Exception:
            pass


def get_or_create_placement(provider, storage, self_address, handler_type, handler_id):
    """service.rs:193-254"""
    buf = C.create_string_buffer(512)
    lib().lpo_get_or_create_placement(provider._h, storage._h, self_address.encode(), handler_type.encode(),
                                      handler_id.encode(), buf, 512)
    return buf.value.decode()


def check_address_mismatch(provider, storage, self_address, server_address):
    """service.rs:261-298 -> 'ok' | 'redirect' | 'deallocate' | 'malformed'"""
    rc = lib().lpo_check_address_mismatch(provider._h, storage._h, self_address.encode(), server_address.encode())
    return {0: "ok", 1: "redirect", 2: "deallocate", -1: "malformed"}[rc]


def bench_policy(n_objects, m_nodes, aff, threads=1, warm=False):
    """Seconds for n_objects get_or_create_placement calls on the string path."""
    dec = C.c_uint64(0)
    s = lib().lpo_bench_policy(n_objects, m_nodes, _u32(aff), threads, int(warm), C.byref(dec))
    return float(s), int(dec.value)


def policy_readback(n_objects, m_nodes, aff, alive=None, cur=None):
    """One get_or_create_placement per object on the string restatement of LocalObjectPlacement + Service (optionally over
    a warm map and with inactive members), then every object looked up again: (seconds of the calls, node index per object
    — 0xFFFFFFFF for a miss)."""
    aff = _u32(aff)
    out = np.empty(n_objects, np.uint32)
    al = None if alive is None else _u8(alive)
    cu = None if cur is None else _u32(cur)
    s = lib().lpo_policy_readback(n_objects, m_nodes, aff, None if al is None else al.ctypes.data_as(C.c_void_p),
                                  None if cu is None else cu.ctypes.data_as(C.c_void_p), out)
    return float(s), out


def hardware_concurrency():
    return int(lib().lpo_hardware_concurrency())
