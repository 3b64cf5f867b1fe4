//! `GpuObjectPlacement` — drop-in `ObjectPlacement` provider backed by the MI355X solver.
//!
//! Goes into the reference tree as `rio-rs/src/object_placement/gpu.rs` behind a cargo feature
//! `gpu` (next to `local`, `sqlite`, `postgres`, `redis`, rio-rs/src/object_placement/mod.rs:10-17).
//! It is a thin FFI wrapper over `librio_gp.so` (include/rio_gpu_object_placement.h); all string
//! interning, batching and the HBM tables live behind the C ABI.
//!
//! NOT COMPILED IN THIS REPOSITORY: the build image has no cargo/rustc.  The C ABI it binds is
//! exercised by tests/test_gpu_object_placement.py through the ctypes twin of this file
//! (rio-rs_amd/rio_gp.py::GpuObjectPlacement), which re-runs the reference's own tests, and
//! tests/test_abi_symbols.py checks every `extern "C"` declaration below against the headers,
//! parameter type by parameter type.  The rest of the crate fragment is next to this file:
//! `../Cargo.toml.patch` (feature + dependency), `../build.rs` (link search path) and
//! `../tests/object_placement_backend_gpu.rs` (the `gpu` twin of the reference's conformance block).
//!
//! Blocking: a native call that needs the device is a round trip (8-25 us when the caller is alone;
//! callers that arrive meanwhile are combined into the same round trip and spin, yield, then sleep).
//! `LocalObjectPlacement` returns in nanoseconds and runs inline on a tokio worker (local.rs:42-49
//! never yields).  This provider does the same for what its host shadow can answer: `lookup` — and
//! the sticky branch of `get_or_create_placement` — first call `rio_op_try_*` INLINE on the async
//! worker (a hash-map read under a reader lock, ~0.25 us, never the device, never a lock a device
//! call holds) and only on `RIO_GP_EAGAIN` hand the blocking call to `tokio::task::spawn_blocking`
//! (a thread hand-off of several microseconds; the blocking pool is where the reference's own SQL
//! providers effectively wait, too).  Every other trait method goes to the blocking pool.

use std::ffi::{c_char, c_int, c_void, CStr, CString};
use std::fmt;
use std::sync::Arc;

use async_trait::async_trait;

use crate::errors::ObjectPlacementError;
use crate::object_placement::{ObjectPlacement, ObjectPlacementItem};
use crate::ObjectId;

#[repr(C)]
struct RioOpCfg {
    struct_size: u32,
    device: i32,
    max_objects: u64,
    max_nodes: u32,
    spill_rounds: u32,
    flags: u32,
    collect_ns: u32,
}

const RIO_GP_OK: c_int = 0;
const RIO_OP_CFG_LIVE_FIRST_TOUCH: u32 = 4; // include/rio_gpu_object_placement.h
const RIO_GP_EINVAL: c_int = 1;
/// the output buffer was too small: nothing is truncated, `rio_op_last_address_len` says what is needed
const RIO_GP_ERANGE: c_int = 5;
/// `rio_op_try_*`: the host shadow cannot answer; nothing was done — make the blocking call
const RIO_GP_EAGAIN: c_int = 6;
/// include/rio_gpu_placement.h RIO_GP_ABI_VERSION: the defaults of the flags changed between 1 and 2 (first touch follows the
/// reference unless RIO_OP_CFG_LIVE_FIRST_TOUCH is set) — a library of another version is refused, not guessed at
const RIO_GP_ABI_VERSION: u32 = 2;
/// `rio_gp_stats` (include/rio_gpu_placement.h): counters of one whole-table solve.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RioGpStats {
    pub n_objects: u64, pub kept: u64, pub evicted: u64, pub claimed: u64, pub spilled: u64, pub unplaced: u64,
    pub load_kept: u64, pub load_claimed: u64, pub load_spilled: u64, pub load_unplaced: u64,
    pub cut_nodes: u32, pub slow_path: u32, pub rounds_run: u32, pub reserved: u32,
}

pub const FLAG_LOCAL: u32 = 0;
pub const FLAG_REDIRECT: u32 = 1;
pub const FLAG_PLACED: u32 = 2;
pub const FLAG_SPILLED: u32 = 3;
pub const FLAG_UNPLACED: u32 = 4;
/// OR-ed onto PLACED / SPILLED / UNPLACED: the object was found on a server that is not alive; that server was cleaned
/// and the object re-placed by this call (service.rs:268-285).
pub const FLAG_REPLACED: u32 = 0x10;
pub const FLAG_MASK: u32 = 0x0F;

#[link(name = "rio_gp")]
extern "C" {
    fn rio_gp_abi_version() -> u32;
    fn rio_op_create(cfg: *const RioOpCfg, out: *mut *mut c_void) -> c_int;
    fn rio_op_clone(p: *mut c_void) -> *mut c_void;
    fn rio_op_release(p: *mut c_void);
    fn rio_op_prepare(p: *mut c_void) -> c_int;
    fn rio_op_last_error(p: *mut c_void) -> *const c_char;
    // keys travel with their lengths (`_n`): a Rust String may hold a NUL byte (service_object.rs:19-26)
    fn rio_op_update_n(p: *mut c_void, ty: *const c_char, ty_len: usize, id: *const c_char, id_len: usize,
                       addr: *const c_char) -> c_int;
    fn rio_op_lookup_n(p: *mut c_void, ty: *const c_char, ty_len: usize, id: *const c_char, id_len: usize,
                       out: *mut c_char, cap: usize, found: *mut c_int) -> c_int;
    // ... answered from the host shadow or RIO_GP_EAGAIN: never the device, never a wait (called inline on the async worker)
    fn rio_op_try_lookup_n(p: *mut c_void, ty: *const c_char, ty_len: usize, id: *const c_char, id_len: usize,
                           out: *mut c_char, cap: usize, found: *mut c_int) -> c_int;
    fn rio_op_try_get_or_create_placement_n(p: *mut c_void, ty: *const c_char, ty_len: usize, id: *const c_char, id_len: usize,
                                            self_addr: *const c_char, out: *mut c_char, cap: usize,
                                            flag: *mut u32) -> c_int;
    fn rio_op_last_address_len(p: *mut c_void) -> usize;
    fn rio_op_clean_server(p: *mut c_void, addr: *const c_char) -> c_int;
    fn rio_op_remove_n(p: *mut c_void, ty: *const c_char, ty_len: usize, id: *const c_char, id_len: usize) -> c_int;
    fn rio_op_set_member(p: *mut c_void, addr: *const c_char, active: c_int, capacity: u64) -> c_int;
    fn rio_op_tick(p: *mut c_void, stats: *mut RioGpStats) -> c_int;
    fn rio_op_snapshot(p: *mut c_void, n_out: *mut u64, tys: *mut *const *const c_char, ids: *mut *const *const c_char,
                       addrs: *mut *const *const c_char) -> c_int;
    fn rio_op_snapshot_key_lengths(p: *mut c_void, ty_lens: *mut *const usize, id_lens: *mut *const usize) -> c_int;
    fn rio_op_get_or_create_placement_n(p: *mut c_void, ty: *const c_char, ty_len: usize, id: *const c_char, id_len: usize,
                                        self_addr: *const c_char, out: *mut c_char, cap: usize,
                                        flag: *mut u32) -> c_int;
    // the batched calls, keys with their lengths as well
    fn rio_op_update_batch_n(p: *mut c_void, n: u64, tys: *const *const c_char, ty_lens: *const usize, ids: *const *const c_char,
                             id_lens: *const usize, addrs: *const *const c_char) -> c_int;
    fn rio_op_lookup_batch_n(p: *mut c_void, n: u64, tys: *const *const c_char, ty_lens: *const usize, ids: *const *const c_char,
                             id_lens: *const usize, out_node_ids: *mut u32) -> c_int;
    fn rio_op_get_or_create_placement_batch_n(p: *mut c_void, n: u64, tys: *const *const c_char, ty_lens: *const usize,
                                              ids: *const *const c_char, id_lens: *const usize, self_addrs: *const *const c_char,
                                              out_node_ids: *mut u32, out_flags: *mut u32) -> c_int;
    fn rio_op_node_address(p: *mut c_void, node_id: u32) -> *const c_char;
}

/// One reference on the shared native state; `Drop` releases it (the HBM tables go with the last).
struct Handle(*mut c_void);
// The native handle is internally synchronized (one mutex + one HIP stream per state).
unsafe impl Send for Handle {}
unsafe impl Sync for Handle {}
impl Drop for Handle {
    fn drop(&mut self) {
        unsafe { rio_op_release(self.0) }
    }
}

/// `Clone` shares the placement table, exactly like `LocalObjectPlacement`'s inner `Arc`
/// (rio-rs/src/object_placement/local.rs:12-18); servers in one process can share one clone
/// (rio-rs/tests/server_utils.rs:62-73).
#[derive(Clone)]
pub struct GpuObjectPlacement {
    inner: Arc<Handle>,
}

impl fmt::Debug for GpuObjectPlacement {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        f.debug_struct("GpuObjectPlacement").finish()
    }
}

#[bon::bon]
impl GpuObjectPlacement {
    #[builder]
    pub fn new(
        #[builder(default = 0)] device: i32,
        #[builder(default = 1 << 24)] max_objects: u64,
        #[builder(default = 4096)] max_nodes: u32,
        #[builder(default = 2)] spill_rounds: u32,
        /// `true` (default): first touch goes to `self.address` whether or not membership marks it active, exactly like
        /// `Service::get_or_create_placement` (service.rs:244-252); `false` opts out — an inactive requester is not a
        /// placement target and its first touches go to the water-fill (RIO_OP_CFG_LIVE_FIRST_TOUCH, the capacity-aware
        /// extension)
        #[builder(default = true)] reference_self_assign: bool,
    ) -> Result<Self, ObjectPlacementError> {
        let cfg = RioOpCfg {
            struct_size: std::mem::size_of::<RioOpCfg>() as u32,
            device, max_objects, max_nodes, spill_rounds,
            flags: if reference_self_assign { 0 } else { RIO_OP_CFG_LIVE_FIRST_TOUCH }, collect_ns: 0,
        };
        let abi = unsafe { rio_gp_abi_version() };
        if abi != RIO_GP_ABI_VERSION {
            return Err(ObjectPlacementError::Unknown(format!(
                "librio_gp speaks ABI version {abi}, this binding was written against {RIO_GP_ABI_VERSION}")));
        }
        let mut h: *mut c_void = std::ptr::null_mut();
        let rc = unsafe { rio_op_create(&cfg, &mut h) };
        if rc != RIO_GP_OK {
            return Err(to_err(rc, std::ptr::null_mut()));
        }
        Ok(Self { inner: Arc::new(Handle(h)) })
    }

    /// Liveness/capacity feed: call from wherever `MembershipStorage::set_is_active` is called
    /// (rio-rs/src/cluster/storage/mod.rs:80; the gossip loop, peer_to_peer.rs:170-191).
    pub fn set_member(&self, address: &str, active: bool, capacity: Option<u64>) -> Result<(), ObjectPlacementError> {
        let a = cstr(address)?;
        check(unsafe { rio_op_set_member(self.inner.0, a.as_ptr(), active as c_int, capacity.unwrap_or(u64::MAX)) }, self)
    }

    /// `get_or_create_placement` for an async caller: the sticky branch (service.rs:199-242 — the object sits on a server that
    /// is an active member) is answered inline from the host shadow; a first touch, an object on a dead server, an unknown
    /// requester go to the blocking pool.  What an unchanged `Service` would call per request.
    pub async fn get_or_create_placement_async(&self, object_id: &ObjectId, self_address: &str)
        -> Result<(Option<String>, u32), ObjectPlacementError> {
        let me_addr = cstr(self_address)?;
        let (ty, id) = (&object_id.0, &object_id.1);
        let mut buf = [0 as c_char; 256];
        let mut flag = 0u32;
        let rc = unsafe { rio_op_try_get_or_create_placement_n(self.inner.0, ty.as_ptr() as *const c_char, ty.len(),
                                                               id.as_ptr() as *const c_char, id.len(), me_addr.as_ptr(),
                                                               buf.as_mut_ptr(), buf.len(), &mut flag) };
        if rc == RIO_GP_OK {
            return Ok((Some(unsafe { CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned()), flag));
        }
        if rc != RIO_GP_EAGAIN && rc != RIO_GP_ERANGE { return Err(to_err(rc, self.inner.0)); }
        let (me, oid, addr) = (self.clone(), object_id.clone(), self_address.to_string());
        blocking(move || me.get_or_create_placement(&oid, &addr)).await
    }

    /// Batched replacement of `Service::get_or_create_placement` + `check_address_mismatch`
    /// (rio-rs/src/service.rs:193-298) for callers that want one call instead of
    /// lookup + is_active + clean_server + update.  Blocking (a device round trip unless the shadow answers).
    pub fn get_or_create_placement(&self, object_id: &ObjectId, self_address: &str)
        -> Result<(Option<String>, u32), ObjectPlacementError> {
        let (ty, id, me) = (&object_id.0, &object_id.1, cstr(self_address)?);
        let mut buf = vec![0 as c_char; 512];
        let mut flag = 0u32;
        let rc = unsafe { rio_op_get_or_create_placement_n(self.inner.0, ty.as_ptr() as *const c_char, ty.len(),
                                                           id.as_ptr() as *const c_char, id.len(), me.as_ptr(),
                                                           buf.as_mut_ptr(), buf.len(), &mut flag) };
        if rc == RIO_GP_ERANGE {
            // the decision is made and `flag` is set; an address longer than the buffer is one lookup (a pure read) away
            return Ok((lookup_owned(self, ty, id)?, flag));
        }
        check(rc, self)?;
        let s = unsafe { CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned();
        Ok((if flag & FLAG_MASK == FLAG_UNPLACED { None } else { Some(s) }, flag))
    }
}

/// (pointer, length) arrays over borrowed key parts: what the `_batch_n` entry points take
struct KeyArrays { ty: Vec<*const c_char>, tl: Vec<usize>, id: Vec<*const c_char>, il: Vec<usize> }
fn key_arrays(keys: &[ObjectId]) -> KeyArrays {
    KeyArrays {
        ty: keys.iter().map(|k| k.0.as_ptr() as *const c_char).collect(), tl: keys.iter().map(|k| k.0.len()).collect(),
        id: keys.iter().map(|k| k.1.as_ptr() as *const c_char).collect(), il: keys.iter().map(|k| k.1.len()).collect(),
    }
}

impl GpuObjectPlacement {
    fn node_address(&self, node: u32) -> Option<String> {
        if node == u32::MAX { return None; }
        let p = unsafe { rio_op_node_address(self.inner.0, node) };  // storage that never moves (a deque of immutable strings)
        if p.is_null() { None } else { Some(unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()) }
    }

    /// `update` for many items in ONE device call (warm start from a placement database: INTEGRATION.md).  Call from a
    /// blocking context (`spawn_blocking`).
    pub fn update_batch(&self, items: &[ObjectPlacementItem]) -> Result<(), ObjectPlacementError> {
        let keys: Vec<ObjectId> = items.iter().map(|i| i.object_id.clone()).collect();
        let ka = key_arrays(&keys);
        let addrs: Vec<Option<CString>> = items.iter().map(|i| i.server_address.as_deref().map(cstr).transpose()).collect::<Result<_, _>>()?;
        let ap: Vec<*const c_char> = addrs.iter().map(|a| a.as_ref().map_or(std::ptr::null(), |c| c.as_ptr())).collect();
        check(unsafe { rio_op_update_batch_n(self.inner.0, keys.len() as u64, ka.ty.as_ptr(), ka.tl.as_ptr(), ka.id.as_ptr(),
                                             ka.il.as_ptr(), ap.as_ptr()) }, self)
    }

    /// `lookup` for many keys in ONE device call.
    pub fn lookup_batch(&self, keys: &[ObjectId]) -> Result<Vec<Option<String>>, ObjectPlacementError> {
        let ka = key_arrays(keys);
        let mut out = vec![u32::MAX; keys.len()];
        check(unsafe { rio_op_lookup_batch_n(self.inner.0, keys.len() as u64, ka.ty.as_ptr(), ka.tl.as_ptr(), ka.id.as_ptr(),
                                             ka.il.as_ptr(), out.as_mut_ptr()) }, self)?;
        Ok(out.into_iter().map(|n| self.node_address(n)).collect())
    }

    /// `Service::get_or_create_placement` for a batch of requests `(object, server the request arrived at)`, processed as if
    /// sequentially in slice order (service.rs:193-298): one device call for a front-end that pre-routes request batches.
    pub fn get_or_create_placement_batch(&self, requests: &[(ObjectId, String)])
        -> Result<Vec<(Option<String>, u32)>, ObjectPlacementError> {
        let keys: Vec<ObjectId> = requests.iter().map(|r| r.0.clone()).collect();
        let ka = key_arrays(&keys);
        let me: Vec<CString> = requests.iter().map(|r| cstr(&r.1)).collect::<Result<_, _>>()?;
        let mp: Vec<*const c_char> = me.iter().map(|c| c.as_ptr()).collect();
        let (mut node, mut flag) = (vec![u32::MAX; keys.len()], vec![0u32; keys.len()]);
        check(unsafe { rio_op_get_or_create_placement_batch_n(self.inner.0, keys.len() as u64, ka.ty.as_ptr(), ka.tl.as_ptr(),
                                                              ka.id.as_ptr(), ka.il.as_ptr(), mp.as_ptr(), node.as_mut_ptr(),
                                                              flag.as_mut_ptr()) }, self)?;
        Ok(node.into_iter().zip(flag).map(|(n, f)| (self.node_address(n), f)).collect())
    }

    /// Eager rebalance (the batched form of the lazy `clean_server` + first-touch path):
    /// every object gets a decision in one call; evicted objects are re-placed at once.
    pub fn tick(&self) -> Result<RioGpStats, ObjectPlacementError> {
        let mut st = RioGpStats::default();
        check(unsafe { rio_op_tick(self.inner.0, &mut st) }, self)?;
        Ok(st)
    }

    /// Every placed entry as `(struct_name, object_id, server_address)` — the columns of the reference's
    /// `object_placement` table, e.g. to write back through `SqliteObjectPlacement::update`.
    /// The arrays `rio_op_snapshot` hands out are copies owned by the calling thread until its next snapshot,
    /// so they are read here, on the same thread, before anything else can run.
    pub fn snapshot(&self) -> Result<Vec<(String, String, String)>, ObjectPlacementError> {
        let (mut n, mut ty, mut id, mut ad) = (0u64, std::ptr::null(), std::ptr::null(), std::ptr::null());
        check(unsafe { rio_op_snapshot(self.inner.0, &mut n, &mut ty, &mut id, &mut ad) }, self)?;
        let (mut tl, mut il) = (std::ptr::null(), std::ptr::null());
        check(unsafe { rio_op_snapshot_key_lengths(self.inner.0, &mut tl, &mut il) }, self)?;
        let s = |p: *const *const c_char, k: usize| unsafe { CStr::from_ptr(*p.add(k)).to_string_lossy().into_owned() };
        // key parts by their true lengths (a NUL byte inside a key is part of the key)
        let key = |p: *const *const c_char, l: *const usize, k: usize| unsafe {
            String::from_utf8_lossy(std::slice::from_raw_parts(*p.add(k) as *const u8, *l.add(k))).into_owned()
        };
        Ok((0..n as usize).map(|k| (key(ty, tl, k), key(id, il, k), s(ad, k))).collect())
    }
}

/// `lookup` into an owned `String` of ANY length (local.rs:42-49 returns `Option<String>`): the native call never
/// truncates — a buffer that is too small comes back as RIO_GP_ERANGE together with the length to allocate.
/// Runs on the calling thread (`rio_op_last_address_len` is that thread's).
fn lookup_owned(me: &GpuObjectPlacement, ty: &str, id: &str) -> Result<Option<String>, ObjectPlacementError> {
    let mut buf = vec![0 as c_char; 512];
    loop {
        let mut found: c_int = 0;
        let rc = unsafe { rio_op_lookup_n(me.inner.0, ty.as_ptr() as *const c_char, ty.len(), id.as_ptr() as *const c_char,
                                          id.len(), buf.as_mut_ptr(), buf.len(), &mut found) };
        if rc == RIO_GP_ERANGE {
            // (another writer may have moved the object to an even longer address meanwhile: loop)
            buf = vec![0 as c_char; unsafe { rio_op_last_address_len(me.inner.0) } + 1];
            continue;
        }
        check(rc, me)?;
        return Ok(if found != 0 { Some(unsafe { CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned()) } else { None });
    }
}

/// server addresses only ("{ip}:{port}" of a Member: no NUL byte can be part of one); object keys travel with their lengths
fn cstr(s: &str) -> Result<CString, ObjectPlacementError> {
    CString::new(s).map_err(|e| ObjectPlacementError::Unknown(e.to_string()))
}

/// errors.rs:135-142: bad argument -> Unknown, anything from HIP -> Upstream.  `rio_op_last_error` returns the text of
/// the CALLING THREAD's last failed call, so this must run on the thread that made the call (it does: inside `blocking`).
fn to_err(rc: c_int, h: *mut c_void) -> ObjectPlacementError {
    let text = unsafe { CStr::from_ptr(rio_op_last_error(h)) }.to_string_lossy().into_owned();
    if rc == RIO_GP_EINVAL { ObjectPlacementError::Unknown(text) } else { ObjectPlacementError::Upstream(text) }
}
fn check(rc: c_int, p: &GpuObjectPlacement) -> Result<(), ObjectPlacementError> {
    if rc == RIO_GP_OK { Ok(()) } else { Err(to_err(rc, p.inner.0)) }
}

/// Run one FFI call on tokio's blocking pool: a device round trip must not park an async worker.
async fn blocking<T, F>(f: F) -> Result<T, ObjectPlacementError>
where
    T: Send + 'static,
    F: FnOnce() -> Result<T, ObjectPlacementError> + Send + 'static,
{
    tokio::task::spawn_blocking(f)
        .await
        .map_err(|e| ObjectPlacementError::Unknown(format!("blocking task failed: {e}")))?
}

#[async_trait]
impl ObjectPlacement for GpuObjectPlacement {
    // mod.rs:42-44: the tables were allocated by `builder()`; forwarded so that a failed device shows up at start-up
    // (Server::prepare, server.rs:122-123) and not on the first request
    async fn prepare(&self) -> Result<(), ObjectPlacementError> {
        let me = self.clone();
        blocking(move || check(unsafe { rio_op_prepare(me.inner.0) }, &me)).await
    }

    // mod.rs:46-49 / local.rs:22-40
    async fn update(&self, object_placement: ObjectPlacementItem) -> Result<(), ObjectPlacementError> {
        let ObjectId(ty, id) = object_placement.object_id.clone();
        let addr = match &object_placement.server_address {
            Some(a) => Some(cstr(a)?),
            None => None, // None deletes (local.rs:36-37)
        };
        let me = self.clone();
        blocking(move || {
            let ap = addr.as_ref().map_or(std::ptr::null(), |a| a.as_ptr());
            check(unsafe { rio_op_update_n(me.inner.0, ty.as_ptr() as *const c_char, ty.len(), id.as_ptr() as *const c_char,
                                           id.len(), ap) }, &me)
        })
        .await
    }

    // mod.rs:50 / local.rs:42-49: a miss is Ok(None).  The call an unchanged Server makes for every request
    // (server.rs:292-304 -> service.rs:199-200): answered inline when the host shadow can (no thread hand-off, like
    // LocalObjectPlacement's hash-map read), on the blocking pool only when the device has to be asked.
    async fn lookup(&self, object_id: &ObjectId) -> Result<Option<String>, ObjectPlacementError> {
        let (ty, id) = (&object_id.0, &object_id.1);
        let mut buf = [0 as c_char; 256];
        let mut found: c_int = 0;
        let rc = unsafe { rio_op_try_lookup_n(self.inner.0, ty.as_ptr() as *const c_char, ty.len(), id.as_ptr() as *const c_char,
                                              id.len(), buf.as_mut_ptr(), buf.len(), &mut found) };
        if rc == RIO_GP_OK {
            return Ok(if found != 0 { Some(unsafe { CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned()) } else { None });
        }
        // RIO_GP_ERANGE (an address longer than the stack buffer) takes the blocking path as well: it sizes its buffer
        if rc != RIO_GP_EAGAIN && rc != RIO_GP_ERANGE { return Err(to_err(rc, self.inner.0)); }
        let (ty, id) = (object_id.0.clone(), object_id.1.clone());
        let me = self.clone();
        blocking(move || lookup_owned(&me, &ty, &id)).await
    }

    // mod.rs:52 / local.rs:51-58
    async fn clean_server(&self, address: String) -> Result<(), ObjectPlacementError> {
        let a = cstr(&address)?;
        let me = self.clone();
        blocking(move || check(unsafe { rio_op_clean_server(me.inner.0, a.as_ptr()) }, &me)).await
    }

    // mod.rs:55 / local.rs:60-68
    async fn remove(&self, object_id: &ObjectId) -> Result<(), ObjectPlacementError> {
        let (ty, id) = (object_id.0.clone(), object_id.1.clone());
        let me = self.clone();
        blocking(move || {
            check(unsafe { rio_op_remove_n(me.inner.0, ty.as_ptr() as *const c_char, ty.len(), id.as_ptr() as *const c_char, id.len()) }, &me)
        })
        .await
    }
}

#[cfg(test)]
mod test {
    use super::*;

    // ObjectId holds any Rust string (service_object.rs:19-26): a NUL byte is part of the key, not its end
    #[tokio::test]
    async fn keys_with_an_interior_nul_are_distinct_keys() {
        let p = GpuObjectPlacement::builder().build().unwrap();
        let a = ObjectId("t".to_string(), "a\0b".to_string());
        let b = ObjectId("t".to_string(), "a".to_string());
        p.update(ObjectPlacementItem::new(a.clone(), Some("0.0.0.0:81".to_string()))).await.unwrap();
        assert_eq!(p.lookup(&a).await.unwrap().as_deref(), Some("0.0.0.0:81"));
        assert!(p.lookup(&b).await.unwrap().is_none());
        p.remove(&a).await.unwrap();
        assert!(p.lookup(&a).await.unwrap().is_none());
    }

    // ... through the batched calls too (`_batch_n`: pointer + length arrays)
    #[tokio::test]
    async fn batched_calls_keep_interior_nuls() {
        let p = GpuObjectPlacement::builder().build().unwrap();
        let a = ObjectId("t".to_string(), "a\0b".to_string());
        let b = ObjectId("t".to_string(), "a".to_string());
        p.update_batch(&[ObjectPlacementItem::new(a.clone(), Some("0.0.0.0:81".to_string())),
                         ObjectPlacementItem::new(b.clone(), Some("0.0.0.0:82".to_string()))]).unwrap();
        let got = p.lookup_batch(&[a.clone(), b.clone()]).unwrap();
        assert_eq!(got, vec![Some("0.0.0.0:81".to_string()), Some("0.0.0.0:82".to_string())]);
        let r = p.get_or_create_placement_batch(&[(a, "0.0.0.0:83".to_string())]).unwrap();
        assert_eq!(r[0].0.as_deref(), Some("0.0.0.0:81"));
    }

    // the same assertions as local.rs:71-123, against the GPU provider
    #[tokio::test]
    async fn gpu_object_placement_provider_is_clonable() {
        let provider = GpuObjectPlacement::builder().build().unwrap();
        let cloned_provider = provider.clone();
        provider
            .update(ObjectPlacementItem::new(ObjectId("test".to_string(), "1".to_string()), Some("0.0.0.0:80".to_string())))
            .await
            .unwrap();
        assert!(provider.lookup(&ObjectId("test".to_string(), "1".to_string())).await.unwrap().is_some());
        assert!(cloned_provider.lookup(&ObjectId("test".to_string(), "1".to_string())).await.unwrap().is_some());
        cloned_provider.clean_server("0.0.0.0:80".to_string()).await.unwrap();
        assert!(provider.lookup(&ObjectId("test".to_string(), "1".to_string())).await.unwrap().is_none());
        assert!(cloned_provider.lookup(&ObjectId("test".to_string(), "1".to_string())).await.unwrap().is_none());
    }
}
