// rio_gp_capi.hip — host side of the C ABI (include/rio_gpu_placement.h): handle, HBM tables,
// stream, and the kernel sequences behind every entry point.  No torch, no CPU fallback.
//
// Reference interfaces replaced (relative to /root/reference):
//   trait ObjectPlacement            rio-rs/src/object_placement/mod.rs:38-56
//   LocalObjectPlacement             rio-rs/src/object_placement/local.rs:22-68
//   Service::get_or_create_placement rio-rs/src/service.rs:193-254 (+ check_address_mismatch :261-298)
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rio_gpu_placement.h"
#ifdef RIO_GP_LAB
#include "../../include/rio_gpu_placement_debug.h"
#endif
#include "placement_kernels.h"

using namespace riogp;

namespace {

thread_local std::string g_create_error;
struct RioGpNcclId { char internal[128]; };  // ncclUniqueId, passed BY VALUE to ncclCommInitRank
constexpr int kRing = 64;  // in-flight async solves whose verdicts we keep

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace

// RCCL, resolved at run time (dlopen of the copy already in the process, else librccl.so.1): the library has
// no link-time dependency on it, and a host that never shards never loads it.  Only the four entry points
// the data path needs; types restated from rccl.h (ncclUniqueId = 128 opaque bytes, ncclUint64 = 5).
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, RioGpNcclId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
constexpr int kNcclUint64 = 5;
constexpr int kShardRing = 4;

struct ShardComm {
    RcclApi api;
    void* comm = nullptr;
    u32 rank = 0, R = 1;
    hipStream_t side = nullptr;  // exchange + global resolve of solve k run here, overlapping solve k+1's scan
    hipEvent_t ready[kShardRing] = {}, done[kShardRing] = {};
    bool done_valid[kShardRing] = {};
    u64* X[kShardRing] = {};
    u64* XG[kShardRing] = {};
    u32 k = 0;
};

// Peer-to-peer exchange windows (rio_gp_shard_p2p_*): one uncached window per rank, IPC-mapped by every peer.
//   xdata [kP2PSlots][R][Wx]  k_resolve_xchg's data-tagged words of rank r (a region of its own: a raw record word whose
//                             upper half happened to equal a step's tag would be taken for that step's data)
//   data  [kP2PSlots][R][W]   raw record of rank r for the step using that slot (fix-up exchanges)
//   flags [kP2PSlots][R][8]   sequence number of the step whose raw record is complete (one 64 B line each)
//   hello [R][8]              set-up handshake
constexpr int kP2PSlots = 4;
struct P2P {
    u32 rank = 0, R = 1;
    size_t W = 0;                 // u64 words per raw record row
    size_t Wx = 0;                // u64 words per tagged row (shard_xchg_words)
    u64* win = nullptr;           // this rank's window
    std::vector<void*> opened;    // peers' windows as mapped here (nullptr for our own)
    u64** d_peers = nullptr;      // device array [R] of window bases (ours included)
    u64* d_err = nullptr;         // set by a waiting kernel that timed out
    u64* scratch = nullptr;       // [W] staging of the local record
    u64 seq = 0;
    // Window slots rotate per KIND of exchange (tagged one-launch exchange | raw fix-up records), not with the sequence number:
    // a tick takes 1 + (1 + rounds) sequence numbers, and when that is a multiple of kP2PSlots every tick's tagged exchange
    // would land in the same slot — a rank that is through a churn-free tick (no further wait on its peers) would overwrite
    // the words a slower rank has not read yet, which then waits for a tag that is gone.  With slots of their own two
    // consecutive exchanges of a kind never share one, and no rank is ever more than one exchange ahead of another.
    u64 xslot_n = 0, yslot_n = 0;
    // A call that has taken sequence numbers / window slots and then fails (a launch error) leaves this rank's counters ahead
    // of what its peers will ever see: every later exchange would wait for, or overwrite, the wrong record.  The session is
    // marked and every later call on it fails with RIO_GP_EUPSTREAM until the windows are connected afresh.
    bool out_of_step = false;
    u32 co_resident = 1;          // ranks whose kernels run on THIS device, ours included (learnt at the handshake)
    size_t xdata_off(u32 slot, u32 r) const { return ((size_t)slot * R + r) * Wx; }
    size_t xwords() const { return (size_t)kP2PSlots * R * Wx; }
    size_t data_off(u32 slot, u32 r) const { return xwords() + ((size_t)slot * R + r) * W; }
    size_t flag_off(u32 slot, u32 r) const { return xwords() + (size_t)kP2PSlots * R * W + ((size_t)slot * R + r) * 8; }
    size_t hello_off(u32 r) const { return xwords() + (size_t)kP2PSlots * R * W + (size_t)kP2PSlots * R * 8 + (size_t)r * 8; }
    size_t total_words() const { return xwords() + (size_t)kP2PSlots * R * W + (size_t)kP2PSlots * R * 8 + (size_t)R * 8; }
};

struct StepGuard {  // see P2P::out_of_step
    P2P* q;
    bool done = false;
    ~StepGuard() { if (q && !done) q->out_of_step = true; }
};
static const char* const kOutOfStep = "the peer-to-peer session lost step with its peers in an earlier failed call: close it on every rank "
                                      "(rio_gp_shard_p2p_close), then export and connect fresh windows";

struct rio_gp {
    ShardComm* sc = nullptr;
    P2P* p2p = nullptr;
    std::mutex mu;
    std::string err;
    int device = 0;
    bool lifecycle = false;            // RIO_GP_CFG_ROW_LIFECYCLE: the affinity column also says which rows are objects
    u32 sa = 0;                        // RIO_GP_CFG_REF_SELF_ASSIGN: claims / first touches do not need a live node (Plan::sa)
    hipStream_t stream = nullptr;      // the stream every call of this handle is enqueued on
    hipStream_t own_stream = nullptr;  // created by rio_gp_create (rio_gp_set_stream may point `stream` elsewhere)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    u64 cap_obj = 0, cap_rows = 0;
    u32 cap_nodes = 0, rounds = 2;
    u64 n = 0;
    u32 m = 0;
    // object table (HBM): two assignment columns (ping-pong), load, affinity, position scratch
    u32* assign[2] = {nullptr, nullptr};
    int cur = 0;
    u32 *load = nullptr, *aff = nullptr, *pos = nullptr;
    // node table
    u64 *cap = nullptr, *used = nullptr;
    u32 *alive_bits = nullptr, *dead_bits = nullptr;
    // A liveness push is a bitmap written into a ring slot of mapped pinned memory (no launch).  It reaches alive_bits with
    // the next whole-table scan, which reads it from the slot (scan_nodes), or through flush_alive when something else
    // needs the device array first.
    u32 *h_alive_ring = nullptr, *d_alive_ring = nullptr;
    u32 alive_slot_words = 0, alive_slot = 0;
    bool alive_dirty = false;
    uint8_t* alive_bytes = nullptr;
    std::vector<uint8_t> h_alive;
    bool used_valid = true;
    // water-fill rounds keep what they admit apart from the solve's `used` vector (SolveBufs::D): the committed vector is
    // h->used + the first parts_rounds rows of D until somebody folds them in (the next solve's k_resolve, or fold_used)
    u64* D = nullptr;
    bool used_parts = false;
    u32 parts_rounds = 0;
    bool solve_used_D = false;  // the solve waiting for its commit ran with sb.D set
    // solve scratch
    SolveBufs sb{};
    DevStats* dstats = nullptr;
    DevStats* h_stats = nullptr;  // pinned scratch for D2H copies of the device accumulators ([0]) + verdicts
    u64* h_slots = nullptr;       // pinned+mapped, kRing slots x slot_rows x 8: k_resolve partial counters
    u64* d_slots = nullptr;       // the same memory as the device sees it
    size_t slot_rows = 0;
    bool all_alive = true;
    Plan plan{};
    bool have_solved = false;
    u32 ring_n = 0;
    u32 ring_slow = 0;     // fix-up verdicts among the rio_gp_solve_async solves whose ring slots were recycled
    bool ring_any = false; // a rio_gp_solve_async solve has been enqueued since the last rio_gp_solve_wait
    // asynchronous committed ticks (rio_gp_tick_async): verdict slots [kRing, 2 kRing) and their own ring of device-stats
    // copies, so that synchronous calls made while ticks are in flight do not touch what has not been harvested yet
    u32 tick_n = 0;
    u32 tick_G[kRing] = {};  // workgroups of the streaming grid of asynchronous tick k (how many counter rows to fold)
    std::vector<rio_gp_stats> tick_done;
    // fix-up counters as per-workgroup rows (FxRows, placement_kernels.h): device rows + pinned slots [1 + kRing][kMaxBlocks][8]
    // (slot 0: synchronous solves, slots 1..kRing: asynchronous ticks)
    u64* fx_dev = nullptr;
    u64* h_fx = nullptr;
    u64* d_fx = nullptr;
    // row-sharded solve (rio_gp_shard_*): global `used` snapshots, forced-node bitmap, spill base, verdict words
    u64 *sh_gprev = nullptr, *sh_gfinal = nullptr, *sh_rank_base = nullptr, *sh_verdict = nullptr;
    u32* sh_forced = nullptr;
    // outputs of the LOCAL column sums (k_resolve in shard mode): kept apart from the solver's arrays, which the
    // global resolve of the PREVIOUS solve may still be writing on the exchange stream
    u64 *sh_lkept = nullptr, *sh_lclaim = nullptr, *sh_lcur = nullptr;
    u32 *sh_lcutblk = nullptr, *sh_lcutidx = nullptr;
    u32 sh_rows = 1;        // verdict rows of the last shard resolve in its pinned slot (1 | resolve_blocks(m))
    u32 sh_rank = 0, sh_R = 1;
    int sh_state = 0;       // 0 idle | 1 scanned | 2 resolved | 3 cut exported | 4 merged | 5 spill exported
    bool sh_slow = false;   // the solve in flight took the fix-up path
    u32 sh_slot = 0;        // verdict slot of the last rio_gp_shard_resolve
    hipStream_t sh_side = nullptr;  // stream the last rio_gp_shard_resolve ran on, when not the handle's
    // packed fix-up (PackOut, placement_kernels.h): scratch columns + per-wave counts; chosen adaptively per tick
    PackOut pk{};
    bool last_pending_valid = false;
    bool searched = false;             // the last enqueue_scan_resolve had k_resolve search the cuts itself
    // A tick that took the fast path leaves every object placed; until the next call that changes an input of the solve
    // (mut_epoch counts those) every further tick keeps every row where it is, and rio_gp_tick_async enqueues no speculative
    // fix-up behind it: two launches a tick instead of five.  quiet_epoch = the mut_epoch such a tick was enqueued under.
    u64 mut_epoch = 0, quiet_epoch = ~0ull;
    u64 tick_mark[kRing] = {}, tick_epoch[kRing] = {};
    bool tick_quiet[kRing] = {};       // the tick was enqueued without its fix-up (checked against its verdict when harvested)
    u32 tick_peeked = 0;               // ticks [0, tick_peeked) of the ring have had their verdicts looked at
    u64 last_pending = 0;
    int compact_mode = 0;  // 0 auto | 1 always | 2 never (rio_gp_debug_set_compact)
    // A committed tick over a mostly-placed table updates the assignment column in place and builds no kept histogram
    // (k_inc_scan), then k_rebal deals the pending rows out evenly to the fix-up's workgroups: 0 auto | 2 never (bits 7-8
    // of rio_gp_debug_set_compact; A/B runs, parity tests)
    int inc_mode = 0;
    int inc_now = 0;            // how the solve waiting for its commit scanned: 0 k_scan | 2 k_inc_scan + k_rebal
    bool solve_inplace = false; // ... and wrote its decisions into the committed column itself: the commit swaps no columns
    PackOut pk2{};              // the balanced pack columns (k_rebal); the undecided rows' lists of k_cut_apply
    u64* Tg = nullptr;          // [max_nodes][16] k_cut_apply's wave sums of the undecided rows (placement_kernels.h, SolveBufs::Tg)
    Plan vplan{};               // the plan of the packed table the fix-up of the solve in flight runs over
    int cutpack_mode = 0;  // the same for packing at the cut pass of whole-table solves (bits 5-6 of rio_gp_debug_set_compact)
    bool ca_now = false;   // the whole-table fix-up of the solve being enqueued is k_cut_apply (set by the caller of enqueue_scan_resolve)
    // Quiet ticks overlap (round 6): a tick that cannot need the fix-up is k_scan + k_resolve, and the NEXT tick's scan reads
    // nothing this tick's k_resolve writes — so k_resolve runs on a stream of its own, behind an event of its scan, while the
    // next scan already streams (H / blkstat alternate between two buffers).  Every other entry point first makes the main
    // stream wait for the last such k_resolve (side_join, in the Locked guard every entry takes).
    // The main stream carries nothing but the scans: each scan's completion IS its event (hipExtLaunchKernel's stop event: no
    // marker packet behind it), and the histograms rotate through a ring of buffers as long as the ticks' ring, so a scan never
    // has to wait for the k_resolve that read its buffer last (checked on the host; a wait is enqueued only if it is not done).
    hipStream_t side = nullptr;
    hipEvent_t ev_scan[kRing] = {}, ev_res[kRing] = {};
    bool ev_res_valid[kRing] = {};
    bool side_pending = false;
    u32 side_last = 0, ov_count = 0, ov_bufs = 0;
    u64* H_ring[kRing] = {}; u64* blk_ring[kRing] = {};
    int overlap_mode = 0;  // 0 on | 2 never (lab builds: bit 11 of rio_gp_debug_set_compact)
    // ... and CHAIN (ScanChain, placement_kernels.h): the scans of a run of overlapped quiet ticks alternate between the main
    // stream and `scan2` and hand their rows over wave range by wave range (a flag per wave; per workgroup in the other form), so the ramp-down of one scan and
    // the ramp-up of the next overlap.  A run starts on the main stream (which orders it behind everything else) and ends with
    // the first side_join: the last k_resolve waits for the last scan, and that scan's workgroups have waited for every earlier one.
    hipStream_t scan2 = nullptr;
    u32* chain_flags = nullptr;                       // [kMaxBlocks] per workgroup + [kMaxBlocks * kWaves] per wave range
    u32 *h_chain_err = nullptr, *d_chain_err = nullptr;  // mapped host word: a chained wait gave up
    u32 chain_seq = 0, chain_prev = 0, chain_pos = 0;  // last sequence number handed out | the run's last scan (0: no run) | its length
    hipEvent_t ev_run = nullptr;  // recorded on the main stream in front of a run's first scan: the run's first scan on `scan2` waits for
                                  // it, so that nothing but the two scans of the chain competes for the chip while one of them waits
    u64 overlap_event_min_rows = (u64)1 << 22;  // ... and with its k_resolve behind events on the side stream from here on
    u64 overlap_min_rows = (u64)1 << 18;  // (2^22 until the in-line form of the chain: below it the events cost more than they hid; lab builds,
                                          //  RIO_GP_OVERLAP_MIN_ROWS: the parity tests run the overlapped / chained ticks on small tables)
    // Small tables (below `inline_below` rows): the k_resolve of a chained tick goes onto ITS SCAN'S stream, right behind the scan —
    // no event ties the two together and none ties the histogram ring to the resolves (a buffer comes round again on the same
    // stream, 64 ticks later): a tick is two plain launches, and the host's enqueue (15.7 us per tick with the events, 7.1
    // without) stops being what bounds a 6 us scan.  On big tables it is the wrong trade: the next scan on that stream then sits
    // behind the resolve while its waves are needed resident (10 M rows: 28.6 against 25.5 us per tick).
    u64 inline_below = (u64)5 << 20;  // (same-run A/Bs at 0.26 / 0.5 / 1 / 2 / 4 / 10 M rows: the two forms cross at ~5 M; lab builds: RIO_GP_CHAIN_INLINE_BELOW)
    bool side_inline = false;     // what side_join has to join is the second scan stream (a chained run), not a k_resolve's event
    hipEvent_t ev_join = nullptr;
    u32 chain_per_wave = 1;       // ScanChain::per_wave: the hand-over per wave range (same-run A/B: 25.6-25.9 against 26.2-26.5 us per tick per
                                  // workgroup; lab builds: RIO_GP_CHAIN_PER_WAVE=0 for the other form)
    int chain_diag = 0;           // lab builds, RIO_GP_CHAIN_DIAG: 1 = the chained kernel on the main stream, no waits | 2 = alternating streams, no waits | 3 = every link waits
                                  // for a sequence number nobody will ever store (the bounded spin and the error path under test)
    bool chain_ok = false;        // two workgroups of the chained scan fit a CU (scan_chain_fits at the table's node count)
    u64 chain_total = 0;   // chained scans enqueued so far (lab builds: rio_gp_debug_chained_scans)
    int chain_mode = 0;    // 0 on | 2 never (lab builds: bit 12 of rio_gp_debug_set_compact)
    int cutapply_mode = 0; // whole-table fix-up by k_cut_apply (cuts + re-marking in one pass): 0 when the solve packs at the cut pass
                           // | 1 always | 2 never (k_cut_find + k_fill<APPLY>) (bits 9-10 of rio_gp_debug_set_compact)
    u64 last_fix_rows = 0;  // rows the previous solve sent to the water-fill (spill candidates + rejected claimants)
    bool last_fix_valid = false;
    int spec_mode = 0;     // speculative fix-up enqueue: 0 auto (after a solve that needed it) | 1 always | 2 never
    int part_mode = 0;     // partitioned CRUD batches: 0 when the batch qualifies | 2 never (rio_gp_debug_set_compact bit 4)
    bool last_slow = false;
    // clean_server(s): dead bitmap + evicted count in mapped pinned memory, self-resetting device counter + ticket
    u32* h_cs = nullptr;
    u32* d_cs = nullptr;
    size_t cs_words = 0;
    u64* cs_cnt = nullptr;
    unsigned int* cs_ticket = nullptr;
    // micro-batch staging: pinned host memory mapped into the device, [6][kSmallBatch] u32 = idx | req | node | flag | status | completion word
    u32* h_small = nullptr;
    u32* d_small = nullptr;
    u64 wait_seq = 1;       // sequence numbers of the synchronous solves (spin_rows): never 0 or 1
    u32* h_mid = nullptr;   // medium batches (<= kMidBatch), mapped pinned memory, [4][kMidBatch] u32: lookup idx | out; place_pending idx | req | out | flag
    u32* d_mid = nullptr;
    unsigned int* mid_ticket = nullptr;  // device word of the several-workgroup completion protocol
    u32* h_req = nullptr;   // request batches of up to kReqBatch entries from host buffers, mapped pinned memory, [4][kReqBatch] u32: idx | req | out | flag
    u32* d_req = nullptr;
    u32* pp_bad = nullptr;  // device word of the general request path: != 0 while a batch with an invalid entry is in flight
                            // ([1]: the window-sorted path's verdict word)
    u64* pp_claim = nullptr;  // [max_nodes + 1] window-sorted request path: claim load per requester + its "needs the solve" counter
    bool pp_last_slow = false;  // the last general-path request batch needed the cut / water-fill: the next one enqueues it speculatively
    DevBuf rq[4];           // staging of bigger host-buffer request batches (the general path's own scratch is vt / stage)
    void* pp_stage = nullptr;            // staging table of the three-launch request path (k_pp_stage / _decide / _apply)
    u32 small_seq = 0;      // sequence number of the last micro-batch call; its completion word is row 5, word 0
    // virtual table (place_pending) and staging for host-pointer calls
    DevBuf vt[4], stage[4];
    DevBuf vrec;  // big place_pending batches: virtual-table records {cur | load}, 8 bytes per request
    DevBuf part;  // scratch of the partitioned update / remove batches (records + fragment tables)
    bool timer_stopped = false;  // rio_gp_timer_stop has recorded the closing event of the measurement in progress
    u32 sh_tick_n = 0;    // asynchronous row-sharded ticks in flight (their records: verdict slots of the tick ring, h_fx slots)
    u64 sh_tick_mark[kRing] = {};
    std::vector<void*> allocs;
};

namespace {

#define HIPCHK(h, call)                                                                       \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e__);                     \
            return RIO_GP_EUPSTREAM;                                                          \
        }                                                                                     \
    } while (0)

void p2p_free(rio_gp* h) {
    P2P* q = h->p2p;
    if (!q) return;
    for (void* o : q->opened) if (o) (void)hipIpcCloseMemHandle(o);
    if (q->d_peers) (void)hipFree(q->d_peers);
    if (q->d_err) (void)hipFree(q->d_err);
    if (q->scratch) (void)hipFree(q->scratch);
    if (q->win) (void)hipFree(q->win);
    delete q;
    h->p2p = nullptr;
}

void shard_comm_free(rio_gp* h) {
    p2p_free(h);
    ShardComm* sc = h->sc;
    if (!sc) return;
    if (sc->side) (void)hipStreamSynchronize(sc->side);
    if (sc->comm && sc->api.CommDestroy) (void)sc->api.CommDestroy(sc->comm);
    for (int q = 0; q < kShardRing; ++q) {
        if (sc->ready[q]) (void)hipEventDestroy(sc->ready[q]);
        if (sc->done[q]) (void)hipEventDestroy(sc->done[q]);
        if (sc->X[q]) (void)hipFree(sc->X[q]);
        if (sc->XG[q]) (void)hipFree(sc->XG[q]);
    }
    if (sc->side) (void)hipStreamDestroy(sc->side);
    delete sc;
    h->sc = nullptr;
}

int fail(rio_gp* h, int rc, const std::string& msg) {
    h->err = msg;
    return rc;
}

template <typename T>
int dalloc(rio_gp* h, T** out, size_t count) {
    void* p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(&p, count * sizeof(T));
    if (e != hipSuccess) {
        h->err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? RIO_GP_ENOMEM : RIO_GP_EUPSTREAM;
    }
    h->allocs.push_back(p);
    *out = static_cast<T*>(p);
    return RIO_GP_OK;
}

int ensure(rio_gp* h, DevBuf& b, size_t bytes) {
    bytes = (bytes + 4095) & ~(size_t)4095;
    bytes += 8 * kTile * sizeof(u32);  // k_scan prefetches up to 4 tiles past n
    if (b.bytes >= bytes) return RIO_GP_OK;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) return fail(h, e == hipErrorOutOfMemory ? RIO_GP_ENOMEM : RIO_GP_EUPSTREAM,
                                      std::string("hipMalloc(staging): ") + hipGetErrorString(e));
    b.bytes = bytes;
    return RIO_GP_OK;
}

void fill_stats(const DevStats& d, u64 n, rio_gp_stats* s) {
    if (!s) return;
    memset(s, 0, sizeof *s);
    (void)n;
    s->n_objects = d.kept + d.claimants + d.spillcand;  // = rows, minus the rows that are not objects (RIO_GP_AFF_INACTIVE)
    s->kept = d.kept;
    s->evicted = d.evicted;
    s->claimed = d.claimants - d.rejected;
    s->spilled = d.spilled;
    s->unplaced = d.unplaced;
    s->load_kept = d.load_kept;
    s->load_claimed = d.load_claim_tot - d.load_rejected;
    s->load_spilled = d.load_spilled;
    s->load_unplaced = d.load_unplaced;
    s->cut_nodes = (uint32_t)d.n_cut;
    s->slow_path = (d.n_cut > 0 || d.spillcand > 0) ? 1u : 0u;
    s->rounds_run = (uint32_t)d.rounds_run;
}

u32* aff_life(rio_gp* h) { return h->lifecycle ? h->aff : nullptr; }
Plan hplan(rio_gp* h, u64 n) {  // the decomposition of a table of n rows under this handle's policy flags
    Plan p = make_plan(n, h->m, 0);
    p.sa = h->sa;
    return p;
}
Table real_table(rio_gp* h) { return Table{h->assign[h->cur], h->load, h->aff, h->assign[h->cur ^ 1]}; }
constexpr u64 kSearchMaxBlockRows = 1u << 17;  // rows per block up to which k_resolve searches the cuts itself
constexpr u32 kAliveSlots = 2 * kRing + 4;  // every slot handed to a scan belongs to a solve or tick of a ring of kRing
void launch_alive_words(rio_gp* h) {
    WordPack pk;
    const u32 words = (h->m + 31) / 32;
    memcpy(pk.w, h->h_alive_ring + (size_t)h->alive_slot * h->alive_slot_words, sizeof(u32) * (words ? words : 1));
    launch_store_words(pk, words, h->alive_bits, h->stream);
}
// the device's liveness bitmap is up to date after this (one tiny kernel if a push is pending)
void flush_alive(rio_gp* h) {
    if (!h->alive_dirty) return;
    h->alive_dirty = false;
    launch_alive_words(h);
}
NodeTab real_nodes(rio_gp* h) { flush_alive(h); return NodeTab{h->cap, h->alive_bits, nullptr}; }
// the node tables for a whole-table solve that starts with k_scan: a pending liveness push rides in that launch
NodeTab scan_nodes(rio_gp* h) {
    NodeTab nt{h->cap, h->alive_bits, nullptr};
    if (h->alive_dirty) {
        h->alive_dirty = false;
        nt.alive_src = h->d_alive_ring + (size_t)h->alive_slot * h->alive_slot_words;
    }
    return nt;
}

// One run of chained scans per process and DEVICE at a time: the chain's progress argument counts the workgroup slots of ONE pair
// of launches (two per CU); a second handle's pair on the same device could hold the slots the first one's earlier launch needs.
// (A host that drives one handle per GPU from one process chains on every one of them.)
constexpr int kChainDevices = 64;
std::atomic<rio_gp*> g_chain_owner[kChainDevices];
std::atomic<rio_gp*>& chain_owner(rio_gp* h) { return g_chain_owner[(unsigned)h->device % kChainDevices]; }
bool chain_begin(rio_gp* h) {
    if (h->chain_prev) return true;  // (a run in progress is this handle's)
    rio_gp* none = nullptr;
    return chain_owner(h).compare_exchange_strong(none, h, std::memory_order_acq_rel) || none == h;
}
void chain_end(rio_gp* h) {
    h->chain_prev = 0;
    h->chain_pos = 0;
    rio_gp* me = h;
    (void)chain_owner(h).compare_exchange_strong(me, nullptr, std::memory_order_acq_rel);
}
// the main stream waits for the k_resolve of the last overlapped quiet tick (no-op when there is none in flight)
void side_join(rio_gp* h) {
    if (!h->side_pending) return;
    (void)hipSetDevice(h->device);
    if (h->side_inline) {  // a chained run with its resolves in line: what is not on the main stream is on `scan2`
        if (h->chain_pos >= 2 &&
            (hipEventRecord(h->ev_join, h->scan2) != hipSuccess || hipStreamWaitEvent(h->stream, h->ev_join, 0) != hipSuccess)) {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(h->scan2);
        }
    } else if (hipStreamWaitEvent(h->stream, h->ev_res[h->side_last], 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(h->side);
    }
    h->side_pending = false;
    h->side_inline = false;
    chain_end(h);  // (the run of chained scans ends here: the next one starts on the main stream, behind this wait)
}
// what every entry point holds: the handle's mutex, with the side stream joined (rio_gp_tick_async joins only when it must)
struct Locked {
    std::unique_lock<std::mutex> l;
    explicit Locked(rio_gp* h, bool join = true) : l(h->mu) { if (join) side_join(h); }
};
bool use_cut_apply(rio_gp* h, u32 m) { return h->cutapply_mode != 2 && !h->sb.forced_bits && cut_apply_fits(m); }
// ... which it is when the solve packs at the cut pass (few rows go on to the water-fill: the ranges with work are a fraction of
// the table and k_cut_apply deals them out over the chip); a solve that re-marks most of the table keeps the two-pass form
bool cut_apply_for(rio_gp* h, bool cutpack) {
    return use_cut_apply(h, h->m) && (h->cutapply_mode == 1 || (cutpack && h->rounds >= 1 && fill_can_pack(h->m)));
}

// The fix-up of a solve whose fast path said it needs one (or may need one: every kernel here guards itself on the
// device, so the sequence can be enqueued before the host has read the verdict):
//   the exact cut search — k_cut_find, unless launch_resolve already searched (packed pending rows: `searched`);
//   round 0 = k_fill<APPLY, FILL> (re-mark + water-fill; cutpack: it also packs the rows that go on to the water-fill, and
//   the later rounds run over those rows only); rounds 1.. = k_fill<FILL>.
void enqueue_slow(rio_gp* h, const Plan& p, const Table& t, const NodeTab& nt, bool virt, bool searched, bool cutpack = false) {
    cutpack = cutpack && !virt && !p.wcnt && h->rounds >= 1 && fill_can_pack(p.m);
    // Whole-table solve of the real table: ONE pass finds the exact cuts, re-marks and (cutpack) packs — k_cut_apply — and
    // every round, the first included, is a plain water-fill round (over the packed rows / over the table).
    if (!virt && !searched && !p.wcnt && h->ca_now) {
        launch_cut_apply(p, t, nt, h->sb, h->pk, h->pk2, h->Tg, cutpack, h->all_alive, h->stream);
        if (cutpack) {
            Plan pp = p;
            pp.wcnt = h->pk.wcnt;
            Table vt{h->pos /* all-NONE column: every packed row is pending */, h->pk.load, h->pk.aff, h->pk.next};
            vt.pk_idx = h->pk.idx;
            vt.real_next = t.next;
            vt.none_prewritten = true;
            for (u32 r = 0; r < h->rounds; ++r) launch_fill(pp, vt, nt, h->sb, true, false, true, (int)r, r + 1 == h->rounds, h->stream);
        } else {
            for (u32 r = 0; r < h->rounds; ++r) launch_fill(p, t, nt, h->sb, false, false, true, (int)r, r + 1 == h->rounds, h->stream);
        }
        return;
    }
    if (!searched) launch_cut_find(p, t, nt, h->sb, virt, h->stream, true);
    launch_fill(p, t, nt, h->sb, virt, true, true, 0, h->rounds == 1, h->stream, cutpack ? &h->pk : nullptr);
    if (cutpack) {
        Plan pp = p;
        pp.wcnt = h->pk.wcnt;
        Table vt{h->pos /* all-NONE column: every packed row is pending */, h->pk.load, h->pk.aff, h->pk.next};
        vt.pk_idx = h->pk.idx;
        vt.real_next = t.next;
        vt.none_prewritten = true;
        for (u32 r = 1; r < h->rounds; ++r) launch_fill(pp, vt, nt, h->sb, true, false, true, (int)r, r + 1 == h->rounds, h->stream);
        return;
    }
    for (u32 r = 1; r < h->rounds; ++r) launch_fill(p, t, nt, h->sb, virt, false, true, (int)r, r + 1 == h->rounds, h->stream);
}

// committed `used` = h->used + the D rows of the last committed solve, until they are folded in: by the next solve's
// k_resolve (for free), or here when somebody needs the vector first
void fold_used(rio_gp* h) {
    if (!h->used_parts) return;
    launch_used_fold(h->used, h->D, h->m, h->parts_rounds, h->stream);
    h->used_parts = false;
}
// scan + resolve of one solve over the REAL table: the packed pending rows' cuts are searched inside k_resolve, the previous
// committed solve's D rows are folded into the committed vector before k_resolve zeroes them
void enqueue_scan_resolve(rio_gp* h, const Table& t, const NodeTab& nt, bool compact, u64* host_rows, int inc = 0, bool overlap = false,
                          bool chained = false) {
    h->sb.D = h->D;
    h->solve_used_D = h->sb.D != nullptr;
    h->inc_now = inc;
    h->solve_inplace = inc != 0;
    const PackOut& pkx = inc == 2 ? h->pk2 : h->pk;  // where the fix-up finds the packed rows
    h->vplan = h->plan;
    // a whole-table fix-up by k_cut_apply takes its ordered spill totals from its own pass: k_scan / k_resolve need not
    // maintain the rejected-load tables R / RP (k_resolve: one prefix over the blocks per node group that owns a cut)
    SolveBufs rb = h->sb;
    h->ca_now = h->ca_now && !compact && !inc;
    if (h->ca_now) { rb.R = nullptr; rb.RP = nullptr; rb.Tg = h->Tg; }
    hipStream_t rs = h->stream;  // where k_resolve goes
    hipStream_t ss = h->stream;  // where the scan goes
    ScanChain ch{h->chain_flags, h->d_chain_err, 0, 0, h->chain_per_wave};
    if (chained) {
        if (!h->chain_prev) {
            (void)hipEventRecord(h->ev_run, h->stream);
        } else if ((h->chain_pos & 1u) && h->chain_diag != 1) {
            ss = h->scan2;
            if (h->chain_pos == 1) (void)hipStreamWaitEvent(ss, h->ev_run, 0);
        }
        ch.wait = (h->chain_diag == 3 && h->chain_prev) ? h->chain_prev + 1000u : h->chain_diag ? 0 : h->chain_prev;  // (3: a predecessor that never comes)
        ch.set = ++h->chain_seq;
        ++h->chain_total;
        h->chain_prev = ch.set;
        ++h->chain_pos;
    }
    u32 par = 0;
    const bool inl = chained && h->n < h->inline_below && h->ev_join && h->chain_diag == 0;
    if (overlap) {  // (a quiet tick: plain k_scan, no fix-up behind it — the cuts' tables are not maintained)
        par = h->ov_count++ % h->ov_bufs;
        rb.R = nullptr; rb.RP = nullptr; rb.Tg = nullptr;
        rb.H = h->H_ring[par]; rb.blkstat = h->blk_ring[par];
        // this scan rewrites the histograms the k_resolve of ov_bufs ticks ago read: done long ago (the ring of ticks is harvested
        // at least as often) — if the runtime does not say so, the main stream waits for it
        if (h->ev_res_valid[par] && hipEventQuery(h->ev_res[par]) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipStreamWaitEvent(ss, h->ev_res[par], 0);
        }
        rs = inl ? ss : h->side;
        if (inl) h->ev_res_valid[par] = false;  // (nothing of this tick is behind an event: stream order and the run's join order it)
    }
    if (inc) {
        // (t.cur is read AND written: the tick is committed, nobody is promised the table as it was)
        launch_inc_scan(h->plan, h->assign[h->cur], h->load, h->aff, nt, h->sb, h->pk, h->stream);
        h->vplan = rebal_plan(h->plan);
        launch_rebal(h->plan, h->vplan, h->pk, nt, h->pk2, h->sb, h->stream);
    } else {
        launch_scan(h->plan, t, nt, rb, false, h->all_alive, ss, nullptr, (overlap && !inl) ? h->ev_scan[par] : nullptr,
                    compact ? &h->pk : nullptr, chained ? &ch : nullptr);
    }
    h->vplan.wcnt = compact ? pkx.wcnt : nullptr;
    // The exact cut search rides in k_resolve when a block's packed rows are few enough for a wave pair per node to stream
    // (config 3: 39 K rows a block, 18 us against 6 + 14 for a resolve and a search launch of their own); on big blocks
    // (config 4 on one GPU: 390 K rows, ~39 K packed) a wave pair per node takes 91 us where k_cut_find's (block, node slice)
    // work items spread over the chip take 36: there the search stays a launch of its own.
    h->searched = compact && h->plan.G && h->n / h->plan.G <= kSearchMaxBlockRows;
    Plan rp = h->vplan;
    if (!h->searched) rp.wcnt = nullptr;
    if (overlap && !inl) (void)hipStreamWaitEvent(rs, h->ev_scan[par], 0);  // (the scan's own stop event)
    launch_resolve(rp, nt, rb, host_rows, rs, nullptr, nullptr, h->searched ? &pkx : nullptr,
                   h->used_parts ? h->used : nullptr, h->parts_rounds, inc ? h->used : nullptr);
    if (overlap && inl) {
        h->side_pending = true;
        h->side_inline = true;
    } else if (overlap) {
        (void)hipEventRecord(h->ev_res[par], rs);
        h->ev_res_valid[par] = true;
        h->side_pending = true;
        h->side_last = par;
    }
    h->used_parts = false;
}
// the fix-up over the rows the scan packed (compact): the water-fill writes every decision through the packed rows' indices
// into the real column itself — the other assignment column, or (k_inc_scan) the committed one
void enqueue_slow_packed(rio_gp* h, const NodeTab& nt) {
    const PackOut& pkx = h->inc_now == 2 ? h->pk2 : h->pk;
    Table vt{h->pos /* all-NONE column: every packed row is pending */, pkx.load, pkx.aff, pkx.next};
    vt.pk_idx = pkx.idx;
    vt.real_next = h->solve_inplace ? h->assign[h->cur] : h->assign[h->cur ^ 1];
    enqueue_slow(h, h->vplan, vt, nt, true, h->searched, false);
}

u64* slot_dev(rio_gp* h, u32 k) { return h->d_slots + (size_t)(k % kRing) * h->slot_rows * 8; }
constexpr u32 kTickSlot0 = (u32)kRing;  // slot index k >= kRing: the asynchronous ticks' half of the slot table

// host-side fold of the per-workgroup partial rows k_resolve stored into a pinned slot
DevStats reduce_rows(rio_gp* h, size_t slot, u32 m);
DevStats reduce_slot(rio_gp* h, u32 k, u32 m) { return reduce_rows(h, k % kRing, m); }                    // solve ring
DevStats reduce_tick_slot(rio_gp* h, u32 k, u32 m) { return reduce_rows(h, kTickSlot0 + k % kRing, m); }  // tick ring
DevStats reduce_rows(rio_gp* h, size_t slot, u32 m) {
    DevStats d;
    memset(&d, 0, sizeof d);
    const u64* rows = h->h_slots + slot * h->slot_rows * 8;
    const unsigned nb = resolve_blocks(m);
    for (unsigned r = 0; r < nb; ++r) {
        const u64* x = rows + (size_t)r * 8;
        d.load_kept += x[0]; d.load_claim_tot += x[1]; d.n_cut += x[2];
        d.kept += x[3]; d.evicted += x[4]; d.claimants += x[5]; d.spillcand += x[6];
    }
    return d;
}

// the fix-up kernels of the next solve write their per-workgroup counter rows into pinned slot `slot` (0: synchronous
// solves, 1 + k: asynchronous tick k)
void use_fx_slot(rio_gp* h, u32 slot) {
    h->sb.fx.dev = h->fx_dev;
    h->sb.fx.host = h->d_fx + (size_t)slot * kMaxBlocks * 8;
    h->sb.fx.seq = 0;
}
// Waiting without the runtime: the kernels of a synchronous solve store the solve's sequence number into word 7 of every
// row they write into mapped pinned memory (k_resolve's partial rows, the last water-fill round's counter rows); the host
// spins until every row carries it.  launch + hipStreamSynchronize costs 12.6 us, launch + spin 7.3 us
// (tools/sync_probe.py).  false: not there after 50 ms — the caller asks the stream (a dead kernel shows up there).
bool spin_rows(const u64* rows, u32 nrows, u64 seq) {
    const volatile u64* r = rows;
    const auto t0 = std::chrono::steady_clock::now();
    u32 spins = 0;
    for (u32 i = 0; i < nrows;) {
        if (r[(size_t)i * 8 + 7] == seq) { ++i; continue; }
        __builtin_ia32_pause();
        if ((++spins & 0xFFFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) return false;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return true;
}
// fold those rows into a fast-path verdict (after the stream has been waited for)
void fold_fx(rio_gp* h, u32 slot, u32 G, DevStats* v) {
    const u64* rows = h->h_fx + (size_t)slot * kMaxBlocks * 8;
    u64 x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (u32 b = 0; b < G; ++b)
        for (int c = 0; c < 8; ++c) x[c] += rows[(size_t)b * 8 + c];
    v->rejected = x[0]; v->load_rejected = x[1];
    v->spilled = x[2]; v->load_spilled = x[3];
    v->unplaced = x[4]; v->load_unplaced = x[5];
    v->rounds_run = x[6];
}
int merge_slow(rio_gp* h, DevStats* v) {
    if (!(h->sb.fx.seq && spin_rows(h->h_fx, h->plan.G, h->sb.fx.seq))) HIPCHK(h, hipStreamSynchronize(h->stream));
    fold_fx(h, 0, h->plan.G, v);
    return RIO_GP_OK;
}

// A solve that works in place (k_inc_scan) has rewritten part of the committed column by the time anything behind it can
// fail.  If the call does not reach its commit, the flags must not outlive it (a later solve that does not pass through
// enqueue_scan_resolve — the row-sharded calls, rio_gp_solve_wait + rio_gp_commit — would skip its column swap and publish a
// stale column), and the table is no longer what the `used` vector and the pending-row statistics describe: the next tick
// re-solves it from scratch (plain k_scan: the kept histogram is rebuilt from the rows).
struct InplaceGuard {
    rio_gp* h;
    bool ok = false;
    ~InplaceGuard() {
        if (ok) return;
        if (h->solve_inplace) { h->used_valid = false; h->used_parts = false; h->last_pending_valid = false; h->last_fix_valid = false; }
        h->solve_inplace = false;
        h->inc_now = 0;
        h->have_solved = false;
    }
};
// every solve entry point that does not go through enqueue_scan_resolve starts from "not in place"
void reset_inplace(rio_gp* h) { h->solve_inplace = false; h->inc_now = 0; }

int commit_enqueue(rio_gp* h) {
    if (!h->have_solved) return fail(h, RIO_GP_EINVAL, "rio_gp_commit: no solve to commit");
    if (!h->solve_inplace) h->cur ^= 1;  // (k_inc_scan and its fix-up wrote the committed column itself)
    h->solve_inplace = false;
    std::swap(h->used, h->sb.used_cur);  // publication = two pointer swaps: the solve's `used` vector becomes the committed one
    h->used_valid = true;
    h->used_parts = h->solve_used_D;     // ... plus what its water-fill rounds admitted (D rows), folded in later
    h->parts_rounds = h->rounds;
    h->have_solved = false;  // (not an input change: mut_epoch stays)
    return RIO_GP_OK;
}

// 0 k_scan | 2 k_inc_scan + k_rebal.  Only a COMMITTED tick may work in place, only a valid `used` vector can stand in for
// the kept histogram, and the rings of pending rows must fit the LDS next to the liveness bitmap.
int inc_choice(rio_gp* h, bool compact, bool commit) {
    if (!compact || !commit || !h->used_valid || h->inc_mode == 2 || !inc_scan_fits(h->m) || h->m == 0) return 0;
    // (Tables whose blocks are beyond the in-resolve cut search — config 4 on one GPU, 390 K rows a block — take this path too:
    // 100 M x 4 096, three boxes, same-run A/B: 700-720 us pipelined against 734-797 with k_scan<COMPACT>; round 0 of the
    // water-fill alone is 60-90 us shorter over the balanced rows.  The cut search stays k_cut_find's launch there.)
    return 2;
}

// One whole-table solve; with `commit` the publication (two pointer swaps) happens before the last wait.
// Host waits: verdict + completion when the fix-up is needed, verdict only on the fast path — and ONE wait when the
// fix-up was enqueued speculatively (below).
int solve_locked(rio_gp* h, rio_gp_stats* stats, bool commit = false) {
    InplaceGuard ipg{h};
    h->plan = hplan(h, h->n);
    h->ring_n = 0; h->ring_slow = 0; h->ring_any = false;
    use_fx_slot(h, 0);
    const u64 seq = ++h->wait_seq;
    h->plan.mark = seq;  // k_resolve's partial rows carry it ...
    // ... and so do the counter rows of the last water-fill round, when that round is the solve's last kernel
    const bool fx_last = h->rounds >= 1;
    if (fx_last) h->sb.fx.seq = seq;
    const Table t = real_table(h);
    const NodeTab nt = scan_nodes(h);
    // Adaptive packed fix-up: when the previous solve left few rows pending — but some: a stream without churn keeps the
    // plain scan, two tiles in flight — (a churn stream: most rows are kept),
    // k_scan also packs this solve's pending rows per wave, and — if the verdict then asks for the fix-up — the cut
    // and water-fill kernels run over the packed rows only (O(pending) passes instead of O(rows)); results identical.
    const bool compact = h->compact_mode == 1 ||
                         (h->compact_mode == 0 && h->last_pending_valid && h->last_pending > 0 && h->last_pending * 4 <= h->n && h->n >= 65536);
    // Speculative fix-up: when the previous solve needed the fix-up (a churn stream needs it every tick), its kernels
    // are enqueued right behind k_resolve instead of after a host round trip for the verdict.  Every fix-up kernel
    // guards itself on device (the cut search: stats->n_cut; the water-fill rounds: pending-row count), so a solve that
    // turns out not to need them pays a few no-op launches and gets the same result.
    const bool spec = h->spec_mode != 2 && (h->spec_mode == 1 || h->last_slow);
    // Packing at the cut pass: a whole-table solve (nothing known to be kept) whose previous solve sent few rows to the
    // water-fill — a contended table re-solved: ~10 % of the rows — lets round 0 of k_fill pack those rows on its way, and
    // the later rounds run over them instead of streaming the table again.  Results identical.
    const bool cutpack = !compact && (h->cutpack_mode == 1 ||
                                      (h->cutpack_mode == 0 && h->last_fix_valid && h->last_fix_rows * 4 <= h->n && h->n >= 65536));
    // ... and when the tick is committed and the library's `used` vector is valid, the scan streams the assignment column
    // alone and works in place (k_inc_scan)
    const int inc = inc_choice(h, compact, commit);
    h->ca_now = cut_apply_for(h, cutpack);
    enqueue_scan_resolve(h, t, nt, compact, slot_dev(h, 0), inc);
    DevStats v;
    bool slow = false;
    const u64* vrows = h->h_slots;  // slot 0 of the solve ring
    if (!spec) {
        if (!spin_rows(vrows, resolve_blocks(h->m), seq)) HIPCHK(h, hipStreamSynchronize(h->stream));
        v = reduce_slot(h, 0, h->m);
        slow = v.n_cut > 0 || v.spillcand > 0;
    }
    if (spec || slow) {
        if (compact) enqueue_slow_packed(h, nt);
        else enqueue_slow(h, h->plan, t, nt, false, false, cutpack);
    }
    h->have_solved = true;
    h->ring_n = 0; h->ring_slow = 0; h->ring_any = false;
    if (commit) {
        int rc = commit_enqueue(h);
        if (rc) return rc;
    }
    if (spec) {
        if (!(fx_last && spin_rows(h->h_fx, h->plan.G, seq))) HIPCHK(h, hipStreamSynchronize(h->stream));
        v = reduce_slot(h, 0, h->m);
        slow = v.n_cut > 0 || v.spillcand > 0;
        if (slow) fold_fx(h, 0, h->plan.G, &v);  // the water-fill rounds stored every workgroup's row into the pinned slot
    } else if (slow) {
        int rc = merge_slow(h, &v);  // waits for the last round's rows (or the stream)
        if (rc) return rc;
    }  // (fast path: k_resolve was the last kernel and its rows are here; the publication is two host-side swaps)
    HIPCHK(h, hipGetLastError());
    h->last_pending = v.claimants + v.spillcand;
    h->last_pending_valid = true;
    h->last_fix_rows = v.spillcand + (slow ? v.rejected : 0);
    h->last_fix_valid = true;
    h->last_slow = slow;
    fill_stats(v, h->n, stats);
    ipg.ok = true;
    return RIO_GP_OK;
}

int commit_locked(rio_gp* h) { return commit_enqueue(h); }

// wait for the asynchronous ticks in flight and turn their verdict slots + device-stats copies into rio_gp_stats
// verdicts of enqueued ticks that have already landed in their pinned slots (every row carries the tick's mark): no wait
void peek_ticks(rio_gp* h) {
    const unsigned nb = resolve_blocks(h->m);
    for (; h->tick_peeked < h->tick_n; ++h->tick_peeked) {
        const u32 k = h->tick_peeked;
        const volatile u64* rows = h->h_slots + (size_t)(kTickSlot0 + k) * h->slot_rows * 8;
        for (unsigned r = 0; r < nb; ++r)
            if (rows[(size_t)r * 8 + 7] != h->tick_mark[k]) return;
        std::atomic_thread_fence(std::memory_order_acquire);
        const DevStats v = reduce_tick_slot(h, k, h->m);
        if (!(v.n_cut > 0 || v.spillcand > 0) && h->tick_epoch[k] == h->mut_epoch) h->quiet_epoch = h->mut_epoch;
    }
}

int harvest_ticks(rio_gp* h) {
    if (!h->tick_n) return RIO_GP_OK;
    side_join(h);
    // When the last tick in flight is a quiet one, its k_resolve is the last kernel of everything in flight (it waited for its
    // scan, the scan's waves for every earlier link of the chain, and the run's first link sat behind everything older on the
    // main stream) and its verdict rows carry the tick's mark: spin on them in mapped memory instead of asking the runtime
    // (launch + hipStreamSynchronize 12.6 us, launch + spin 7.3: tools/sync_probe.py).  Not there after 50 ms: the stream is asked.
    const u32 last = h->tick_n - 1;
    if (!(h->tick_quiet[last] &&
          spin_rows(h->h_slots + (size_t)(kTickSlot0 + last) * h->slot_rows * 8, resolve_blocks(h->m), h->tick_mark[last])))
        HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    if (h->h_chain_err && *reinterpret_cast<volatile u32*>(h->h_chain_err)) {  // never seen; must not pass silently if it happens
        *h->h_chain_err = 0;
        h->tick_n = 0;
        return fail(h, RIO_GP_EUPSTREAM, "rio_gp_tick_wait: a chained scan gave up waiting for the previous tick's rows (tables are stale: reload them)");
    }
    h->tick_peeked = 0;
    for (u32 k = 0; k < h->tick_n; ++k) {
        DevStats v = reduce_tick_slot(h, k, h->m);
        const bool slow = v.n_cut > 0 || v.spillcand > 0;
        if (slow && h->tick_quiet[k]) {  // cannot happen (see mut_epoch); if it ever does it must not pass silently
            h->tick_n = 0;
            return fail(h, RIO_GP_EUPSTREAM, "rio_gp_tick_wait: a tick that was enqueued without its fix-up needed one (tables are stale: reload them)");
        }
        if (!slow && h->tick_epoch[k] == h->mut_epoch) h->quiet_epoch = h->mut_epoch;
        if (slow) fold_fx(h, 1 + k, h->tick_G[k], &v);
        rio_gp_stats st;
        fill_stats(v, h->n, &st);
        h->tick_done.push_back(st);
        h->last_pending = v.claimants + v.spillcand;
        h->last_pending_valid = true;
        h->last_slow = slow;
    }
    h->tick_n = 0;
    return RIO_GP_OK;
}

// One committed tick, nothing waits on the host: k_scan (packing when the last known solve left few rows pending),
// k_resolve into this tick's verdict slot, the whole fix-up behind it (every fix-up kernel guards itself on the device),
// the publication (two pointer swaps, host side) and an asynchronous copy of the device accumulators into this tick's
// pinned record.  The result is the one rio_gp_tick computes; only the counters arrive later (rio_gp_tick_wait).
int tick_async_locked(rio_gp* h) {
    if (h->ring_n) return fail(h, RIO_GP_EINVAL, "rio_gp_tick_async: rio_gp_solve_async solves are in flight (call rio_gp_solve_wait)");
    // (the row-sharded ticks keep their records in the same verdict slots and counter rows)
    if (h->sh_tick_n) return fail(h, RIO_GP_EINVAL, "rio_gp_tick_async: row-sharded ticks are in flight (call rio_gp_shard_tick_wait)");
    if (h->tick_n == (u32)kRing) { int rc = harvest_ticks(h); if (rc) return rc; }
    peek_ticks(h);
    // nothing has changed since a tick that left every object placed: this one keeps every row, no fix-up can be needed
    // (lab builds: rio_gp_debug_set_speculate(always) keeps the launches)
    const bool quiet = h->quiet_epoch == h->mut_epoch && h->spec_mode != 1;
    // ... where it pays: the scan long enough to hide the event's cost behind it (config 2's 6 us scan lost 3-8 us per tick to it)
    // and a histogram ring as long as the ticks' ring (64 buffers within 256 MiB: up to 1 024 nodes), so that no scan ever waits
    // for a resolve.  Config 4 on one GPU (4 096 nodes: 16 MB of histograms per tick) was measured both ways — 16 buffers: the
    // host runs 64 ticks ahead and every scan waits for a resolve that is itself starved beside a scan; 64 buffers (1 GB): no
    // waits, and still 330 against 313 us per tick: a k_resolve of 512 workgroups over 16 MB beside a DRAM-bound scan costs the
    // scan more than it saves.
    // Tables below 2^22 rows overlap only as links of a chain with the resolves in line (two plain launches a tick: 9.4-11.5 us
    // against 11.2-14.5 at 0.26-2 M rows); with the events of the other form they lose (15.8-17.7).
    const bool ov_can = quiet && h->overlap_mode != 2 && h->side && h->stream == h->own_stream && h->n >= h->overlap_min_rows &&
                        h->ov_bufs >= (u32)kRing;
    if (ov_can && h->chain_seq >= 0x70000000u) {  // (sequence numbers compare by signed difference: start over long before they wrap)
        side_join(h);
        (void)hipMemsetAsync(h->chain_flags, 0, (size_t)kMaxBlocks * (1 + kWaves) * sizeof(u32), h->stream);
        h->chain_seq = 0;
    }
    // (chained: a pushed liveness bitmap rides in a scan that nothing behind it may overtake — a quiet tick has none; the chained
    //  scan addresses its columns by 32-bit byte offsets: tables below 2^30 rows, 4 GiB a column)
    const bool chained = ov_can && h->chain_mode != 2 && h->scan2 && h->chain_ok && !h->alive_dirty && h->cap_rows < ((size_t)1 << 30) &&
                         chain_begin(h);
    const bool overlap = ov_can && (chained || h->n >= h->overlap_event_min_rows);
    if (!overlap || (!chained && h->chain_prev)) side_join(h);
    InplaceGuard ipg{h};
    h->plan = hplan(h, h->n);
    const Table t = real_table(h);
    const NodeTab nt = scan_nodes(h);
    const bool compact = !quiet && (h->compact_mode == 1 ||
                         (h->compact_mode == 0 && h->last_pending_valid && h->last_pending > 0 && h->last_pending * 4 <= h->n && h->n >= 65536));
    const u32 k = h->tick_n;
    use_fx_slot(h, 1 + k);
    h->tick_G[k] = h->plan.G;
    h->tick_epoch[k] = h->mut_epoch;
    h->tick_quiet[k] = quiet;
    h->tick_mark[k] = h->plan.mark = (1ull << 40) | ++h->wait_seq;  // column 7 of the verdict rows: peek_ticks knows them by it
    h->ca_now = cut_apply_for(h, false);
    enqueue_scan_resolve(h, t, nt, compact, h->d_slots + (size_t)(kTickSlot0 + k) * h->slot_rows * 8, inc_choice(h, compact, true),
                         overlap, chained);
    if (quiet) {
        // (k_scan + k_resolve only)
    } else if (compact) {
        enqueue_slow_packed(h, nt);
    } else {
        enqueue_slow(h, h->plan, t, nt, false, false);
    }
    h->have_solved = true;
    int rc = commit_enqueue(h);
    if (rc) return rc;
    HIPCHK(h, hipGetLastError());
    h->tick_n = k + 1;
    ipg.ok = true;
    return RIO_GP_OK;
}

// Micro-batch calls (one workgroup) end by storing their sequence number into a word of mapped pinned memory
// (signal_done, placement_kernels.hip); the host spins on it — launch + hipStreamSynchronize measured 12.6 us, launch +
// spin 7.3 us (tools/sync_probe.py).  A kernel that dies never writes the word: after ~50 ms the stream is asked.
u32 small_begin(rio_gp* h) {
    if (++h->small_seq == 0) h->small_seq = 1;
    return h->small_seq;
}
u32* small_done_dev(rio_gp* h) { return h->d_small + 5 * kSmallBatch; }
bool small_inline(SmallInline* inl, uint64_t n, const uint32_t* a, const uint32_t* b) {
    if (n > 4) return false;
    for (uint64_t k = 0; k < 4; ++k) {
        inl->a[k] = k < n ? a[k] : 0u;
        inl->b[k] = (k < n && b) ? b[k] : 0u;
    }
    return true;
}
int small_wait(rio_gp* h, u32 seq, volatile u32* w = nullptr) {
    if (!w) w = h->h_small + 5 * kSmallBatch;
    const auto t0 = std::chrono::steady_clock::now();
    for (u32 spins = 1; *w != seq; ++spins) {
        __builtin_ia32_pause();
        if ((spins & 0xFFFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
    }
    if (*w == seq) {
        std::atomic_thread_fence(std::memory_order_acquire);
        return RIO_GP_OK;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    if (*w != seq) return fail(h, RIO_GP_EUPSTREAM, "micro-batch kernel left no completion word");
    return RIO_GP_OK;
}

int ensure_used(rio_gp* h) {
    if (h->used_valid) { fold_used(h); return RIO_GP_OK; }
    h->used_parts = false;  // rebuilt from the assignment column: nothing to fold
    launch_recompute_used(h->assign[h->cur], h->load, h->n, h->m, h->used, h->stream);
    h->used_valid = true;
    return RIO_GP_OK;
}

int zero_stats(rio_gp* h) {
    HIPCHK(h, hipMemsetAsync(h->dstats, 0, sizeof(DevStats), h->stream));
    return RIO_GP_OK;
}

int read_stats(rio_gp* h) {
    HIPCHK(h, hipMemcpyAsync(h->h_stats, h->dstats, sizeof(DevStats), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return RIO_GP_OK;
}

}  // namespace

extern "C" {

uint32_t rio_gp_abi_version(void) { return RIO_GP_ABI_VERSION; }

const char* rio_gp_last_error(rio_gp_t* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

const char* rio_gp_backend(rio_gp_t*) { return "hip:gfx950"; }

int rio_gp_create(const rio_gp_cfg* cfg, rio_gp_t** out) {
    if (out) *out = nullptr;
    if (!cfg || !out || cfg->struct_size != sizeof(rio_gp_cfg)) {
        g_create_error = "rio_gp_create: bad cfg (struct_size mismatch)";
        return RIO_GP_EINVAL;
    }
    if (cfg->max_objects > RIO_GP_MAX_OBJECTS || cfg->max_nodes > RIO_GP_MAX_NODES) {
        g_create_error = "rio_gp_create: max_objects/max_nodes above the solver limits";
        return RIO_GP_EINVAL;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
        g_create_error = std::string("rio_gp_create: no usable HIP device (") +
                         (e != hipSuccess ? hipGetErrorString(e) : "device ordinal out of range") +
                         "); there is no CPU fallback";
        return RIO_GP_ENODEV;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("rio_gp_create: device is not gfx950 (MI355X): ") + prop.gcnArchName;
        return RIO_GP_ENODEV;
    }
    rio_gp* h = new rio_gp();
    h->device = cfg->device;
    h->lifecycle = (cfg->flags & RIO_GP_CFG_ROW_LIFECYCLE) != 0;
    h->sa = (cfg->flags & RIO_GP_CFG_REF_SELF_ASSIGN) ? 1u : 0u;
    h->cap_obj = cfg->max_objects;
    h->cap_rows = ((cfg->max_objects + kTile - 1) / kTile) * kTile + 8 * kTile;  // k_scan prefetches past the end
    h->cap_nodes = cfg->max_nodes ? cfg->max_nodes : 1;
    h->rounds = cfg->spill_rounds ? cfg->spill_rounds : 2;
    if (h->rounds > kFillRounds) {
        g_create_error = "rio_gp_create: spill_rounds above the solver limit (8)";
        delete h;
        return RIO_GP_EINVAL;
    }
    int rc = RIO_GP_OK;
    auto bail = [&](int code) {
        g_create_error = h->err;
        rio_gp_destroy(h);
        return code;
    };
    if (hipSetDevice(h->device) != hipSuccess) { h->err = "hipSetDevice failed"; return bail(RIO_GP_EUPSTREAM); }
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess || !(h->stream = h->own_stream) ||
        hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
        h->err = "stream/event creation failed";
        return bail(RIO_GP_EUPSTREAM);
    }
    if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); h->side = nullptr; }
    for (int q = 0; q < kRing && h->side; ++q)
        if (hipEventCreate(&h->ev_scan[q]) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_res[q], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipStreamDestroy(h->side);
            h->side = nullptr;  // (no overlap: everything else works)
        }
    if (h->side && (hipStreamCreateWithFlags(&h->scan2, hipStreamNonBlocking) != hipSuccess ||
                    hipEventCreateWithFlags(&h->ev_run, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess)) {
        (void)hipGetLastError();
        if (h->scan2) (void)hipStreamDestroy(h->scan2);
        h->scan2 = nullptr;  // (no chain: everything else works)
    }
    h->chain_ok = h->scan2 && scan_chain_fits((u32)h->cap_nodes);
#ifdef RIO_GP_LAB
    if (const char* e = getenv("RIO_GP_CHAIN_DIAG")) h->chain_diag = atoi(e);  // (1, 2: timing experiments only — the waits are what makes the chain correct)
    if (const char* e = getenv("RIO_GP_CHAIN_PER_WAVE")) h->chain_per_wave = (u32)atoi(e);
    if (const char* e = getenv("RIO_GP_CHAIN_INLINE_BELOW")) h->inline_below = strtoull(e, nullptr, 10);
    if (const char* e = getenv("RIO_GP_OVERLAP_MIN_ROWS")) h->overlap_min_rows = h->overlap_event_min_rows = strtoull(e, nullptr, 10);
#endif
    const size_t R = h->cap_rows, M = h->cap_nodes, W = (size_t)kMaxBlocks * kWaves;
    // the balanced pack columns (k_rebal) have uniform wave ranges: up to a tile per wave range more than the table; the
    // all-NONE column stands in for their `cur` column as well
    const size_t R2 = std::max((size_t)rebal_rows(h->cap_obj) + 8 * kTile, R);
#define A(ptr, cnt) if ((rc = dalloc(h, &(ptr), (cnt))) != RIO_GP_OK) return bail(rc)
    A(h->assign[0], R); A(h->assign[1], R); A(h->load, R); A(h->aff, R); A(h->pos, R2);
    A(h->cap, M); A(h->used, M); A(h->alive_bits, (M + 31) / 32 + 4); A(h->dead_bits, (M + 31) / 32 + 4);
    A(h->alive_bytes, M);
    A(h->sb.H, (size_t)((M + 7) / 8) * kMaxBlocks * 16); A(h->sb.blkstat, (size_t)kMaxBlocks * 4);
    {   // the histogram ring of overlapped quiet ticks: as many buffers as ticks may be in flight, within 256 MiB
        const size_t hwords = (size_t)((M + 7) / 8) * kMaxBlocks * 16;
        size_t nb = ((size_t)256 << 20) / (hwords * sizeof(u64));
        nb = nb > (size_t)kRing ? (size_t)kRing : nb < 2 ? 2 : nb;
        h->H_ring[0] = h->sb.H; h->blk_ring[0] = h->sb.blkstat;
        for (size_t q = 1; q < nb; ++q) { A(h->H_ring[q], hwords); A(h->blk_ring[q], (size_t)kMaxBlocks * 4); }
        h->ov_bufs = (u32)nb;
    }
    A(h->sb.partial, (size_t)resolve_blocks((u32)M) * 8 + 8);
    A(h->sb.wsp_sum[0], W); A(h->sb.wsp_sum[1], W); A(h->sb.wsp_cnt[0], W); A(h->sb.wsp_cnt[1], W);
    A(h->sb.bsp_sum[0], (size_t)kMaxBlocks); A(h->sb.bsp_sum[1], (size_t)kMaxBlocks); A(h->sb.bsp_cnt[0], (size_t)kMaxBlocks); A(h->sb.bsp_cnt[1], (size_t)kMaxBlocks);
    A(h->sb.used_kept, M); A(h->sb.used_cur, M); A(h->sb.claim_tot, M); A(h->sb.cutblk, M); A(h->sb.budget, M);
    A(h->sb.admpre, M); A(h->sb.cutidx, M); A(h->dstats, 1); A(h->fx_dev, (size_t)kMaxBlocks * 8);
    A(h->sb.R, (size_t)kMaxBlocks); A(h->sb.RP, (size_t)kMaxBlocks * resolve_blocks((u32)M)); A(h->D, (size_t)kFillRounds * M);
    A(h->pk.idx, R); A(h->pk.load, R); A(h->pk.aff, R); A(h->pk.next, R); A(h->pk.wcnt, W);
    A(h->pk2.idx, R2); A(h->pk2.load, R2); A(h->pk2.aff, R2); A(h->pk2.next, R2); A(h->pk2.wcnt, W);
    A(h->Tg, M * kWaves);
    A(h->chain_flags, (size_t)kMaxBlocks * (1 + kWaves));
    A(h->sh_lkept, M); A(h->sh_lclaim, M); A(h->sh_lcur, M); A(h->sh_lcutblk, M); A(h->sh_lcutidx, M);
    A(h->sh_gprev, M); A(h->sh_gfinal, M); A(h->sh_rank_base, 2); A(h->sh_verdict, 8); A(h->sh_forced, (M + 31) / 32 + 4);
#undef A
    h->sb.stats = h->dstats;
    if (hipHostMalloc(reinterpret_cast<void**>(&h->h_stats), sizeof(DevStats) * kRing, hipHostMallocMapped) !=
        hipSuccess) {
        h->err = "hipHostMalloc failed";
        return bail(RIO_GP_ENOMEM);
    }
    if (hipHostMalloc(reinterpret_cast<void**>(&h->h_chain_err), 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_chain_err), h->h_chain_err, 0) != hipSuccess) {
        h->err = "hipHostMalloc(chain error word) failed";
        return bail(RIO_GP_ENOMEM);
    }
    memset(h->h_chain_err, 0, 64);
    h->slot_rows = resolve_blocks(h->cap_nodes);
    if (hipHostMalloc(reinterpret_cast<void**>(&h->h_slots), (size_t)2 * kRing * h->slot_rows * 8 * sizeof(u64),
                      hipHostMallocMapped) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&h->h_fx), (size_t)(1 + kRing) * kMaxBlocks * 8 * sizeof(u64), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_fx), h->h_fx, 0) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_slots), h->h_slots, 0) != hipSuccess) {
        h->err = "hipHostMalloc(mapped verdict slots) failed";
        return bail(RIO_GP_ENOMEM);
    }
    memset(h->h_slots, 0, (size_t)2 * kRing * h->slot_rows * 8 * sizeof(u64));
    memset(h->h_fx, 0, (size_t)(1 + kRing) * kMaxBlocks * 8 * sizeof(u64));
    h->cs_words = (((size_t)h->cap_nodes + 31) / 32 + 8 + 1) & ~(size_t)1;  // bitmap words, then the u64 count (8 B aligned)
    if (hipHostMalloc(reinterpret_cast<void**>(&h->h_cs), (h->cs_words + 4) * sizeof(u32), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_cs), h->h_cs, 0) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&h->cs_cnt), 9 * 16 * sizeof(u64)) != hipSuccess ||  // k_clean: total + 8 group counters, a line each
        hipMalloc(reinterpret_cast<void**>(&h->cs_ticket), sizeof(unsigned int)) != hipSuccess ||
        hipMemset(h->cs_cnt, 0, 9 * 16 * sizeof(u64)) != hipSuccess || hipMemset(h->cs_ticket, 0, sizeof(unsigned int)) != hipSuccess) {
        h->err = "clean_server staging allocation failed";
        return bail(RIO_GP_ENOMEM);
    }
    h->allocs.push_back(h->cs_cnt);
    h->allocs.push_back(h->cs_ticket);
    h->alive_slot_words = (((u32)h->cap_nodes + 31) / 32 + 4 + 31) & ~31u;
    if (hipHostMalloc(reinterpret_cast<void**>(&h->h_alive_ring), (size_t)kAliveSlots * h->alive_slot_words * sizeof(u32),
                      hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_alive_ring), h->h_alive_ring, 0) != hipSuccess) {
        h->err = "hipHostMalloc(mapped liveness ring) failed";
        (void)hipGetLastError();
        rio_gp_destroy(h);
        return RIO_GP_EUPSTREAM;
    }
    if (hipHostMalloc(reinterpret_cast<void**>(&h->h_small), (size_t)6 * kSmallBatch * sizeof(u32), hipHostMallocMapped) !=
            hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_small), h->h_small, 0) != hipSuccess) {
        h->err = "hipHostMalloc(mapped micro-batch staging) failed";
        return bail(RIO_GP_ENOMEM);
    }
    memset(h->h_small, 0, (size_t)6 * kSmallBatch * sizeof(u32));  // completion word: 0 = no call yet (sequence numbers start at 1)
    if (hipHostMalloc(reinterpret_cast<void**>(&h->h_mid), (size_t)4 * kMidBatch * sizeof(u32), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_mid), h->h_mid, 0) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&h->mid_ticket), sizeof(unsigned int)) != hipSuccess ||
        hipMemset(h->mid_ticket, 0, sizeof(unsigned int)) != hipSuccess) {
        h->err = "hipHostMalloc(mapped medium-batch staging) failed";
        return bail(RIO_GP_ENOMEM);
    }
    h->allocs.push_back(h->mid_ticket);
    if (hipHostMalloc(reinterpret_cast<void**>(&h->h_req), (size_t)4 * kReqBatch * sizeof(u32), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_req), h->h_req, 0) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&h->pp_bad), 64) != hipSuccess || hipMemset(h->pp_bad, 0, 64) != hipSuccess) {
        h->err = "hipHostMalloc(mapped request staging) failed";
        return bail(RIO_GP_ENOMEM);
    }
    h->allocs.push_back(h->pp_bad);
    if ((rc = dalloc(h, &h->pp_claim, (size_t)h->cap_nodes + 1)) != RIO_GP_OK) return bail(rc);
    if (hipMalloc(&h->pp_stage, pp_stage_bytes()) != hipSuccess || hipMemset(h->pp_stage, 0, pp_stage_bytes()) != hipSuccess) {
        h->err = "request staging allocation failed";
        return bail(RIO_GP_ENOMEM);
    }
    h->allocs.push_back(h->pp_stage);
    // every row starts unplaced; the position scratch is all-ones between calls
    launch_fill_u32(h->assign[0], R, kNone, h->stream);
    launch_fill_u32(h->assign[1], R, kNone, h->stream);
    launch_fill_u32(h->pos, R2, kNone, h->stream);
    launch_fill_u32(h->load, R, 0, h->stream);
    launch_fill_u32(h->aff, R, kNone, h->stream);
    (void)hipMemsetAsync(h->used, 0, M * sizeof(u64), h->stream);
    (void)hipMemsetAsync(h->alive_bits, 0, ((M + 31) / 32 + 4) * sizeof(u32), h->stream);
    (void)hipMemsetAsync(h->dstats, 0, sizeof(DevStats), h->stream);
    (void)hipMemsetAsync(h->D, 0, (size_t)kFillRounds * M * sizeof(u64), h->stream);
    (void)hipMemsetAsync(h->sb.R, 0, (size_t)kMaxBlocks * sizeof(u64), h->stream);
    (void)hipMemsetAsync(h->chain_flags, 0, (size_t)kMaxBlocks * (1 + kWaves) * sizeof(u32), h->stream);
    if (hipStreamSynchronize(h->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
        h->err = "initial fill failed (no gfx950 code object loaded?)";
        return bail(RIO_GP_EUPSTREAM);
    }
    *out = h;
    return RIO_GP_OK;
}

void rio_gp_destroy(rio_gp_t* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->own_stream && h->own_stream != h->stream) (void)hipStreamSynchronize(h->own_stream);
    shard_comm_free(h);
    for (void* p : h->allocs) (void)hipFree(p);
    for (auto& b : h->vt) if (b.p) (void)hipFree(b.p);
    for (auto& b : h->stage) if (b.p) (void)hipFree(b.p);
    for (auto& b : h->rq) if (b.p) (void)hipFree(b.p);
    if (h->vrec.p) (void)hipFree(h->vrec.p);
    if (h->part.p) (void)hipFree(h->part.p);
    if (h->h_stats) (void)hipHostFree(h->h_stats);
    if (h->h_chain_err) (void)hipHostFree(h->h_chain_err);
    if (h->h_slots) (void)hipHostFree(h->h_slots);
    if (h->h_fx) (void)hipHostFree(h->h_fx);
    if (h->h_small) (void)hipHostFree(h->h_small);
    if (h->h_mid) (void)hipHostFree(h->h_mid);
    if (h->h_req) (void)hipHostFree(h->h_req);
    if (h->h_cs) (void)hipHostFree(h->h_cs);
    if (h->h_alive_ring) (void)hipHostFree(h->h_alive_ring);
    chain_end(h);
    if (h->scan2) { (void)hipStreamSynchronize(h->scan2); (void)hipStreamDestroy(h->scan2); }
    if (h->ev_run) (void)hipEventDestroy(h->ev_run);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->side) { (void)hipStreamSynchronize(h->side); (void)hipStreamDestroy(h->side); }
    for (int q = 0; q < kRing; ++q) {
        if (h->ev_scan[q]) (void)hipEventDestroy(h->ev_scan[q]);
        if (h->ev_res[q]) (void)hipEventDestroy(h->ev_res[q]);
    }
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev2) (void)hipEventDestroy(h->ev2);
    if (h->ev3) (void)hipEventDestroy(h->ev3);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int rio_gp_set_flags(rio_gp_t* h, uint32_t flags) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    const uint32_t fixed = h->lifecycle ? RIO_GP_CFG_ROW_LIFECYCLE : 0u;
    if ((flags & ~RIO_GP_CFG_REF_SELF_ASSIGN) != fixed) return fail(h, RIO_GP_EINVAL, "rio_gp_set_flags: only RIO_GP_CFG_REF_SELF_ASSIGN may change");
    const u32 sa = (flags & RIO_GP_CFG_REF_SELF_ASSIGN) ? 1u : 0u;
    if (sa && (h->p2p || h->sc)) return fail(h, RIO_GP_EINVAL, "rio_gp_set_flags: row-sharded handles do not implement RIO_GP_CFG_REF_SELF_ASSIGN");
    if (sa != h->sa) { h->sa = sa; h->have_solved = false; ++h->mut_epoch; }
    return RIO_GP_OK;
}

int rio_gp_sync(rio_gp_t* h) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RIO_GP_OK;
}

uint64_t rio_gp_num_objects(rio_gp_t* h) { return h ? h->n : 0; }
uint32_t rio_gp_num_nodes(rio_gp_t* h) { return h ? h->m : 0; }

// ---- node table ---------------------------------------------------------------------------

int rio_gp_set_nodes(rio_gp_t* h, uint32_t m, const uint64_t* cap, const uint8_t* alive) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    if (m > h->cap_nodes) return fail(h, RIO_GP_EINVAL, "rio_gp_set_nodes: m exceeds max_nodes");
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<u64> c(m ? m : 1, RIO_GP_CAP_INF);
    if (cap) memcpy(c.data(), cap, (size_t)m * sizeof(u64));
    h->h_alive.assign(m, 1);
    if (alive) for (uint32_t j = 0; j < m; ++j) h->h_alive[j] = alive[j] ? 1 : 0;
    if (m) {
        HIPCHK(h, hipMemcpyAsync(h->cap, c.data(), (size_t)m * sizeof(u64), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->alive_bytes, h->h_alive.data(), m, hipMemcpyHostToDevice, h->stream));
    }
    launch_pack_alive(h->alive_bytes, m, h->alive_bits, h->stream);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->alive_dirty = false;
    h->all_alive = true;
    for (uint32_t j = 0; j < m; ++j) h->all_alive = h->all_alive && h->h_alive[j];
    if (m != h->m) { h->used_valid = false; h->used_parts = false; }
    h->m = m;
    h->have_solved = false; ++h->mut_epoch;
    return RIO_GP_OK;
}

// Liveness push without a host wait and without a launch: the bitmap is packed into a ring slot of mapped pinned memory;
// the next whole-table scan reads it from there (scan_nodes), anything else that needs the device array first makes
// flush_alive deliver it in the arguments of one tiny kernel.
static int push_alive_bits(rio_gp* h) {
    static_assert(RIO_GP_MAX_NODES <= 256 * 32, "WordPack holds RIO_GP_MAX_NODES bits");
    const u32 words = (h->m + 31) / 32;
    if (!h->alive_dirty) h->alive_slot = (h->alive_slot + 1) % kAliveSlots;  // (an unconsumed push is simply overwritten)
    u32* w = h->h_alive_ring + (size_t)h->alive_slot * h->alive_slot_words;
    memset(w, 0, sizeof(u32) * (words ? words : 1));
    h->all_alive = true;
    for (uint32_t j = 0; j < h->m; ++j) {
        if (h->h_alive[j]) w[j >> 5] |= 1u << (j & 31);
        else h->all_alive = false;
    }
    std::atomic_thread_fence(std::memory_order_release);
    h->alive_dirty = true;
    h->have_solved = false; ++h->mut_epoch;
    return RIO_GP_OK;
}

int rio_gp_set_alive_all(rio_gp_t* h, uint32_t m, const uint8_t* alive) {
    if (!h || !alive) return RIO_GP_EINVAL;
    Locked g(h);
    if (m != h->m) return fail(h, RIO_GP_EINVAL, "rio_gp_set_alive_all: m differs from the node table");
    HIPCHK(h, hipSetDevice(h->device));
    for (uint32_t j = 0; j < m; ++j) h->h_alive[j] = alive[j] ? 1 : 0;
    return push_alive_bits(h);
}

int rio_gp_set_alive(rio_gp_t* h, uint32_t node, uint8_t alive) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    if (node >= h->m) return fail(h, RIO_GP_EINVAL, "rio_gp_set_alive: node out of range");
    HIPCHK(h, hipSetDevice(h->device));
    h->h_alive[node] = alive ? 1 : 0;
    return push_alive_bits(h);
}

int rio_gp_get_nodes(rio_gp_t* h, uint32_t m, uint64_t* cap, uint8_t* alive, uint64_t* used) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    if (m != h->m) return fail(h, RIO_GP_EINVAL, "rio_gp_get_nodes: m differs from the node table");
    HIPCHK(h, hipSetDevice(h->device));
    if (used) { int rc = ensure_used(h); if (rc) return rc; }
    if (cap && m) HIPCHK(h, hipMemcpyAsync(cap, h->cap, (size_t)m * sizeof(u64), hipMemcpyDeviceToHost, h->stream));
    if (used && m) HIPCHK(h, hipMemcpyAsync(used, h->used, (size_t)m * sizeof(u64), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (alive) memcpy(alive, h->h_alive.data(), m);
    return RIO_GP_OK;
}

// ---- object table -------------------------------------------------------------------------

static int set_objects_impl(rio_gp_t* h, uint64_t n, const uint32_t* load, const uint32_t* aff, hipMemcpyKind kind) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    if (n > h->cap_obj) return fail(h, RIO_GP_EINVAL, "rio_gp_set_objects: n exceeds max_objects");
    HIPCHK(h, hipSetDevice(h->device));
    if (load) { if (n) HIPCHK(h, hipMemcpyAsync(h->load, load, n * sizeof(u32), kind, h->stream)); }
    else launch_fill_u32(h->load, n, 1u, h->stream);
    if (aff) { if (n) HIPCHK(h, hipMemcpyAsync(h->aff, aff, n * sizeof(u32), kind, h->stream)); }
    else launch_fill_u32(h->aff, n, h->lifecycle ? kAffInactive : kNone, h->stream);
    launch_fill_u32(h->assign[h->cur], h->cap_rows, kNone, h->stream);
    HIPCHK(h, hipMemsetAsync(h->used, 0, (size_t)h->cap_nodes * sizeof(u64), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->n = n;
    h->used_valid = true;
    h->used_parts = false;
    h->have_solved = false; ++h->mut_epoch;
    return RIO_GP_OK;
}
int rio_gp_set_objects(rio_gp_t* h, uint64_t n, const uint32_t* load, const uint32_t* aff) {
    return set_objects_impl(h, n, load, aff, hipMemcpyHostToDevice);
}
int rio_gp_set_objects_dev(rio_gp_t* h, uint64_t n, const uint32_t* d_load, const uint32_t* d_aff) {
    return set_objects_impl(h, n, d_load, d_aff, hipMemcpyDeviceToDevice);
}

static int set_assign_impl(rio_gp_t* h, uint64_t n, const uint32_t* assign, hipMemcpyKind kind) {
    if (!h || !assign) return RIO_GP_EINVAL;
    Locked g(h);
    if (n != h->n) return fail(h, RIO_GP_EINVAL, "rio_gp_set_assign: n differs from the object table");
    if (kind == hipMemcpyHostToDevice)
        for (uint64_t i = 0; i < n; ++i)
            if (assign[i] != RIO_GP_NONE && assign[i] >= h->m)
                return fail(h, RIO_GP_EINVAL, "rio_gp_set_assign: node id out of range");
    HIPCHK(h, hipSetDevice(h->device));
    if (n) HIPCHK(h, hipMemcpyAsync(h->assign[h->cur], assign, n * sizeof(u32), kind, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->used_valid = false;
    h->have_solved = false; ++h->mut_epoch;
    return RIO_GP_OK;
}
int rio_gp_set_assign(rio_gp_t* h, uint64_t n, const uint32_t* a) { return set_assign_impl(h, n, a, hipMemcpyHostToDevice); }
int rio_gp_set_assign_dev(rio_gp_t* h, uint64_t n, const uint32_t* a) { return set_assign_impl(h, n, a, hipMemcpyDeviceToDevice); }

int rio_gp_get_assign(rio_gp_t* h, uint64_t n, uint32_t* out) {
    if (!h || !out) return RIO_GP_EINVAL;
    Locked g(h);
    if (n != h->n) return fail(h, RIO_GP_EINVAL, "rio_gp_get_assign: n differs from the object table");
    HIPCHK(h, hipSetDevice(h->device));
    if (n) HIPCHK(h, hipMemcpyAsync(out, h->assign[h->cur], n * sizeof(u32), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RIO_GP_OK;
}
int rio_gp_get_solved(rio_gp_t* h, uint64_t n, uint32_t* out) {
    if (!h || !out) return RIO_GP_EINVAL;
    Locked g(h);
    if (n != h->n || !h->have_solved) return fail(h, RIO_GP_EINVAL, "rio_gp_get_solved: no solve / size mismatch");
    HIPCHK(h, hipSetDevice(h->device));
    if (n) HIPCHK(h, hipMemcpyAsync(out, h->assign[h->cur ^ 1], n * sizeof(u32), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RIO_GP_OK;
}
int rio_gp_get_objects(rio_gp_t* h, uint64_t n, uint32_t* out_load, uint32_t* out_aff) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    if (n != h->n) return fail(h, RIO_GP_EINVAL, "rio_gp_get_objects: n differs from the object table");
    HIPCHK(h, hipSetDevice(h->device));
    if (n && out_load) HIPCHK(h, hipMemcpyAsync(out_load, h->load, n * sizeof(u32), hipMemcpyDeviceToHost, h->stream));
    if (n && out_aff) HIPCHK(h, hipMemcpyAsync(out_aff, h->aff, n * sizeof(u32), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RIO_GP_OK;
}
int rio_gp_set_object_attrs(rio_gp_t* h, uint64_t n, const uint32_t* idx, const uint32_t* load, const uint32_t* aff) {
    if (!h || (n && !idx)) return RIO_GP_EINVAL;
    Locked g(h);
    for (uint64_t k = 0; k < n; ++k)
        if (idx[k] >= h->n) return fail(h, RIO_GP_EINVAL, "rio_gp_set_object_attrs: object index out of range");
    if (!n || (!load && !aff)) return RIO_GP_OK;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    for (int q = 0; q < 3; ++q)
        if ((rc = ensure(h, h->stage[q], n * sizeof(u32)))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->stage[0].p, idx, n * sizeof(u32), hipMemcpyHostToDevice, h->stream));
    if (load) HIPCHK(h, hipMemcpyAsync(h->stage[1].p, load, n * sizeof(u32), hipMemcpyHostToDevice, h->stream));
    if (aff) HIPCHK(h, hipMemcpyAsync(h->stage[2].p, aff, n * sizeof(u32), hipMemcpyHostToDevice, h->stream));
    if ((rc = zero_stats(h))) return rc;
    launch_set_attrs(h->load, h->aff, h->n, (const u32*)h->stage[0].p, load ? (const u32*)h->stage[1].p : nullptr,
                     aff ? (const u32*)h->stage[2].p : nullptr, n, h->dstats, h->stream);
    if (load) h->used_valid = false;
    h->have_solved = false; ++h->mut_epoch;
    return read_stats(h);
}

int rio_gp_count_placed(rio_gp_t* h, uint64_t* out) {
    if (!h || !out) return RIO_GP_EINVAL;
    Locked g(h);
    HIPCHK(h, hipSetDevice(h->device));
    int rc = zero_stats(h);
    if (rc) return rc;
    launch_count_placed(h->assign[h->cur], h->n, h->dstats, h->stream);
    if ((rc = read_stats(h))) return rc;
    *out = h->h_stats[0].evicted_clean;
    return RIO_GP_OK;
}

int rio_gp_set_num_objects(rio_gp_t* h, uint64_t n) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    if (n > h->cap_obj) return fail(h, RIO_GP_EINVAL, "rio_gp_set_num_objects: n exceeds max_objects");
    // rows keep their contents: rows >= n simply take no part (and are rejected as indices) until n grows again — but a
    // placed row that drops out (or comes back) changes what `used` must count, so the vector is rebuilt before its next use
    if (n != h->n) h->used_valid = false;
    h->n = n;
    h->have_solved = false; ++h->mut_epoch;
    h->last_pending_valid = false;
    return RIO_GP_OK;
}

const uint32_t* rio_gp_assign_dev(rio_gp_t* h) { return h ? h->assign[h->cur] : nullptr; }
const uint32_t* rio_gp_solved_dev(rio_gp_t* h) { return h ? h->assign[h->cur ^ 1] : nullptr; }

// ---- CRUD ---------------------------------------------------------------------------------

int rio_gp_lookup_batch_dev(rio_gp_t* h, uint64_t n, const uint32_t* d_idx, uint32_t* d_out) {
    if (!h || (n && (!d_idx || !d_out))) return RIO_GP_EINVAL;
    Locked g(h);
    HIPCHK(h, hipSetDevice(h->device));
    int rc = zero_stats(h);
    if (rc) return rc;
    launch_lookup(h->assign[h->cur], h->n, d_idx, n, d_out, h->dstats, h->stream);
    if ((rc = read_stats(h))) return rc;
    if (h->h_stats[0].err) return fail(h, RIO_GP_EINVAL, "rio_gp_lookup_batch: object index out of range");
    return RIO_GP_OK;
}

int rio_gp_lookup_batch(rio_gp_t* h, uint64_t n, const uint32_t* idx, uint32_t* out_node) {
    if (!h || (n && (!idx || !out_node))) return RIO_GP_EINVAL;
    Locked g(h);
    for (uint64_t k = 0; k < n; ++k)
        if (idx[k] >= h->n) return fail(h, RIO_GP_EINVAL, "rio_gp_lookup_batch: object index out of range");
    if (!n) return RIO_GP_OK;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if (n <= (uint64_t)kSmallBatch) {  // micro-batch: the gather reads and writes mapped pinned memory, one launch + wait
        SmallInline inl;
        const bool in_args = small_inline(&inl, n, idx, nullptr);  // n <= 4: the requests ride in the kernel arguments
        if (!in_args) memcpy(h->h_small, idx, n * sizeof(u32));
        const u32 seq = small_begin(h);
        launch_lookup_small(h->assign[h->cur], h->n, h->d_small, (u32)n, h->d_small + 2 * kSmallBatch, h->dstats, h->stream,
                            small_done_dev(h), seq, in_args ? &inl : nullptr);
        if ((rc = small_wait(h, seq))) return rc;
        memcpy(out_node, h->h_small + 2 * kSmallBatch, n * sizeof(u32));
        return RIO_GP_OK;
    }
    if (n <= (uint64_t)kMidBatch) {
        // medium batch: indices and results in mapped pinned memory, a few workgroups, the last one stores the completion
        // word — no staging copies through the runtime, no stream wait (1 000 lookups: 29.5 -> ~13 us per call)
        memcpy(h->h_mid, idx, n * sizeof(u32));
        const u32 seq = small_begin(h);
        launch_lookup(h->assign[h->cur], h->n, h->d_mid, n, h->d_mid + kMidBatch, h->dstats, h->stream, small_done_dev(h), seq,
                      h->mid_ticket);
        if ((rc = small_wait(h, seq))) return rc;
        memcpy(out_node, h->h_mid + kMidBatch, n * sizeof(u32));
        return RIO_GP_OK;
    }
    if ((rc = ensure(h, h->stage[0], n * sizeof(u32))) || (rc = ensure(h, h->stage[1], n * sizeof(u32)))) return rc;
    u32 *d_idx = (u32*)h->stage[0].p, *d_out = (u32*)h->stage[1].p;
    HIPCHK(h, hipMemcpyAsync(d_idx, idx, n * sizeof(u32), hipMemcpyHostToDevice, h->stream));
    launch_lookup(h->assign[h->cur], h->n, d_idx, n, d_out, h->dstats, h->stream);
    HIPCHK(h, hipMemcpyAsync(out_node, d_out, n * sizeof(u32), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return RIO_GP_OK;
}

// End of a synchronous call of the window-partitioned kernels: the error counter and the completion word arrive in mapped host
// memory (k_finish_err); no counter copy-back, no hipStreamSynchronize.
static int finish_err(rio_gp* h, const char* what) {
    const u32 seq = small_begin(h);
    h->h_small[4 * kSmallBatch] = 0;
    launch_finish_err(h->dstats, h->d_small + 4 * kSmallBatch, small_done_dev(h), seq, h->stream);
    int rc = small_wait(h, seq);
    if (rc) return rc;
    if (h->h_small[4 * kSmallBatch]) return fail(h, RIO_GP_EINVAL, what);
    return RIO_GP_OK;
}

static int update_dev_locked(rio_gp* h, uint64_t n, const uint32_t* d_idx, const uint32_t* d_node) {
    int rc = zero_stats(h);
    if (rc) return rc;
    if (h->part_mode != 2 && part_applicable(h->n, n, d_idx, d_node)) {  // big batch: binned by row window, elected in LDS, written densely
        if ((rc = ensure(h, h->part, part_scratch_words(h->n, n) * sizeof(u32)))) return rc;
        launch_update_part(h->assign[h->cur], h->n, h->m, d_idx, d_node, n, (u32*)h->part.p, h->dstats, h->stream, aff_life(h));
        h->used_valid = false;
        h->have_solved = false; ++h->mut_epoch;
        return finish_err(h, "rio_gp_update_batch: invalid entries were skipped");
    } else {
        launch_update(h->assign[h->cur], h->n, h->m, d_idx, d_node, n, h->pos, h->dstats, h->stream, aff_life(h));
    }
    h->used_valid = false;
    h->have_solved = false; ++h->mut_epoch;
    if ((rc = read_stats(h))) return rc;
    if (h->h_stats[0].err) return fail(h, RIO_GP_EINVAL, "rio_gp_update_batch: invalid entries were skipped");
    return RIO_GP_OK;
}

int rio_gp_update_batch_dev(rio_gp_t* h, uint64_t n, const uint32_t* d_idx, const uint32_t* d_node) {
    if (!h || (n && (!d_idx || !d_node))) return RIO_GP_EINVAL;
    Locked g(h);
    HIPCHK(h, hipSetDevice(h->device));
    return update_dev_locked(h, n, d_idx, d_node);
}

int rio_gp_update_batch(rio_gp_t* h, uint64_t n, const uint32_t* idx, const uint32_t* node) {
    if (!h || (n && (!idx || !node))) return RIO_GP_EINVAL;
    Locked g(h);
    for (uint64_t k = 0; k < n; ++k)
        if (idx[k] >= h->n || (node[k] != RIO_GP_NONE && node[k] >= h->m))
            return fail(h, RIO_GP_EINVAL, "rio_gp_update_batch: index or node out of range");
    if (!n) return RIO_GP_OK;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if (n <= (uint64_t)kSmallBatch) {
        // micro-batch (one first-touch update per activation in the reference flow, service.rs:244-252): the entries were
        // validated above, so the kernels read them from mapped pinned memory and nothing is copied, zeroed or read back
        SmallInline inl;
        const bool in_args = small_inline(&inl, n, idx, node);
        if (!in_args) {
            memcpy(h->h_small, idx, n * sizeof(u32));
            memcpy(h->h_small + kSmallBatch, node, n * sizeof(u32));
        }
        const u32 seq = small_begin(h);
        // `used` follows the writes when it is valid (k_remove_small does the same): a server that mixes single updates with
        // policy calls does not re-stream the whole table before every place_pending
        fold_used(h);
        launch_update_small(h->assign[h->cur], h->d_small, h->d_small + kSmallBatch, (u32)n, h->stream, aff_life(h),
                            small_done_dev(h), seq, in_args ? &inl : nullptr, h->used_valid ? h->used : nullptr, h->load, h->m);
        h->have_solved = false; ++h->mut_epoch;
        return small_wait(h, seq);
    }
    if (n <= (uint64_t)kMidBatch) {
        // medium batch, validated above: the election and the apply kernel read the entries from mapped pinned memory, the
        // apply kernel's last workgroup stores the completion word — no staging copies, no counter read-back, no stream wait
        memcpy(h->h_mid, idx, n * sizeof(u32));
        memcpy(h->h_mid + kMidBatch, node, n * sizeof(u32));
        const u32 seq = small_begin(h);
        fold_used(h);
        launch_update(h->assign[h->cur], h->n, h->m, h->d_mid, h->d_mid + kMidBatch, n, h->pos, h->dstats, h->stream, aff_life(h),
                      h->mid_ticket, small_done_dev(h), seq, h->used_valid ? h->used : nullptr, h->load);
        h->have_solved = false; ++h->mut_epoch;
        return small_wait(h, seq);
    }
    if ((rc = ensure(h, h->stage[0], n * sizeof(u32))) || (rc = ensure(h, h->stage[1], n * sizeof(u32)))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->stage[0].p, idx, n * sizeof(u32), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->stage[1].p, node, n * sizeof(u32), hipMemcpyHostToDevice, h->stream));
    return update_dev_locked(h, n, (const u32*)h->stage[0].p, (const u32*)h->stage[1].p);
}

static int remove_dev_locked(rio_gp* h, uint64_t n, const uint32_t* d_idx) {
    int rc = zero_stats(h);
    if (rc) return rc;
    fold_used(h);
    if (h->part_mode != 2 && part_applicable(h->n, n, d_idx, nullptr)) {
        if ((rc = ensure(h, h->part, part_scratch_words(h->n, n) * sizeof(u32)))) return rc;
        launch_remove_part(h->assign[h->cur], h->n, h->m, h->load, d_idx, n, (u32*)h->part.p, h->used_valid ? h->used : nullptr,
                           h->dstats, h->stream, aff_life(h));
        h->have_solved = false; ++h->mut_epoch;
        return finish_err(h, "rio_gp_remove_batch: invalid entries were skipped");
    } else {
        launch_remove(h->assign[h->cur], h->n, h->m, h->load, d_idx, n, h->used_valid ? h->used : nullptr, h->dstats,
                      h->stream, aff_life(h));
    }
    h->have_solved = false; ++h->mut_epoch;
    if ((rc = read_stats(h))) return rc;
    if (h->h_stats[0].err) return fail(h, RIO_GP_EINVAL, "rio_gp_remove_batch: invalid entries were skipped");
    return RIO_GP_OK;
}

int rio_gp_remove_batch_dev(rio_gp_t* h, uint64_t n, const uint32_t* d_idx) {
    if (!h || (n && !d_idx)) return RIO_GP_EINVAL;
    Locked g(h);
    HIPCHK(h, hipSetDevice(h->device));
    return remove_dev_locked(h, n, d_idx);
}

int rio_gp_remove_batch(rio_gp_t* h, uint64_t n, const uint32_t* idx) {
    if (!h || (n && !idx)) return RIO_GP_EINVAL;
    Locked g(h);
    for (uint64_t k = 0; k < n; ++k)
        if (idx[k] >= h->n) return fail(h, RIO_GP_EINVAL, "rio_gp_remove_batch: object index out of range");
    if (!n) return RIO_GP_OK;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if (n <= (uint64_t)kSmallBatch) {  // micro-batch: validated above, read from mapped pinned memory, one launch + one wait
        SmallInline inl;
        const bool in_args = small_inline(&inl, n, idx, nullptr);
        if (!in_args) memcpy(h->h_small, idx, n * sizeof(u32));
        const u32 seq = small_begin(h);
        fold_used(h);
        launch_remove_small(h->assign[h->cur], h->m, h->load, h->d_small, (u32)n, h->used_valid ? h->used : nullptr, h->stream,
                            aff_life(h), small_done_dev(h), seq, in_args ? &inl : nullptr);
        h->have_solved = false; ++h->mut_epoch;
        return small_wait(h, seq);
    }
    if (n <= (uint64_t)kMidBatch) {  // medium batch, validated above: as update_batch
        memcpy(h->h_mid, idx, n * sizeof(u32));
        const u32 seq = small_begin(h);
        fold_used(h);
        launch_remove(h->assign[h->cur], h->n, h->m, h->load, h->d_mid, n, h->used_valid ? h->used : nullptr, h->dstats,
                      h->stream, aff_life(h), small_done_dev(h), seq, nullptr, h->mid_ticket);
        h->have_solved = false; ++h->mut_epoch;
        return small_wait(h, seq);
    }
    if ((rc = ensure(h, h->stage[0], n * sizeof(u32)))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->stage[0].p, idx, n * sizeof(u32), hipMemcpyHostToDevice, h->stream));
    return remove_dev_locked(h, n, (const u32*)h->stage[0].p);
}

int rio_gp_clean_servers(rio_gp_t* h, const uint64_t* dead_bitmap, uint64_t* evicted) {
    if (!h || !dead_bitmap) return RIO_GP_EINVAL;
    Locked g(h);
    HIPCHK(h, hipSetDevice(h->device));
    // one launch + one wait: the kernel reads the bitmap from mapped pinned memory and its last workgroup writes the
    // evicted count back into it (no staging copy, no counter memset, no copy-back)
    const u32 words32 = (h->m + 31) / 32;
    u64* h_count = reinterpret_cast<u64*>(h->h_cs + h->cs_words);
    *h_count = 0;
    if (evicted) *evicted = 0;
    if (!words32) return RIO_GP_OK;
    bool any = false;
    for (u32 w = 0; w < words32; ++w) h->h_cs[w] = 0;
    for (u32 j = 0; j < h->m; ++j)
        if ((dead_bitmap[j >> 6] >> (j & 63)) & 1ull) { h->h_cs[j >> 5] |= 1u << (j & 31); any = true; }
    h->have_solved = false; ++h->mut_epoch;
    if (!any || h->n == 0) return RIO_GP_OK;  // retain() with a predicate nothing matches, or over an empty map
    const u32 seq = (small_begin(h) & 0xFFFFFFu) | 0x800000u;  // 24 bits, never 0
    fold_used(h);  // (k_clean zeroes the dead nodes' entries: what the last solve's rounds admitted there must be in first)
    launch_clean(h->assign[h->cur], h->n, h->m, h->d_cs, h->used_valid ? h->used : nullptr, h->dstats, h->stream,
                 h->cs_cnt, h->cs_ticket, reinterpret_cast<u64*>(h->d_cs + h->cs_words), aff_life(h), seq);
    {   // the last workgroup stores total | seq << 40 into mapped pinned memory: spin on the tag, ask the stream after 50 ms
        volatile u64* w = h_count;
        const auto t0 = std::chrono::steady_clock::now();
        u32 spins = 0;
        while ((*w >> 40) != seq) {
            __builtin_ia32_pause();
            if ((++spins & 0xFFFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
        }
        if ((*w >> 40) != seq) {
            HIPCHK(h, hipStreamSynchronize(h->stream));
            HIPCHK(h, hipGetLastError());
            if ((*w >> 40) != seq) return fail(h, RIO_GP_EUPSTREAM, "rio_gp_clean_servers: the kernel left no total");
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (evicted) *evicted = *h_count & ((1ull << 40) - 1);
    return RIO_GP_OK;
}

int rio_gp_clean_server(rio_gp_t* h, uint32_t node, uint64_t* evicted) {
    if (!h) return RIO_GP_EINVAL;
    u32 m;
    {
        Locked g(h);
        m = h->m;
        if (node >= m) {  // an address nothing was ever placed on: retain() removes nothing (local.rs:56)
            if (evicted) *evicted = 0;
            return RIO_GP_OK;
        }
    }
    std::vector<uint64_t> bm((m + 63) / 64 + 1, 0);
    bm[node >> 6] |= 1ull << (node & 63);
    return rio_gp_clean_servers(h, bm.data(), evicted);
}

// ---- policy -------------------------------------------------------------------------------

// The general path of place_pending (any batch the one-workgroup kernels do not finish): the window-sorted form for big dense
// batches, else k_ppm_first -> [k_clean] -> k_ppm_gather -> solve of the virtual table against the committed `used` (same
// kernels, VIRT) -> [fix-up] -> k_ppm_output, with nothing waiting on the host in between (placement_kernels.hip).
// d_idx / d_req / d_out / d_flag: device-resident arrays, or (host_io) rows of mapped pinned host memory — then the first
// kernel leaves device copies of the requests for the kernels behind it and the last one writes the results over PCIe.
// Synchronous: returns when the results are in d_out / d_flag.  dev_api: the entries were NOT validated on the host.
static int place_pending_general(rio_gp* h, uint64_t n, const u32* d_idx, const u32* d_req, u32* d_out, u32* d_flag,
                                 bool host_io, bool dev_api) {
    flush_alive(h);  // the request kernels read the device's liveness bitmap
    int rc;
    const size_t bytes = n * sizeof(u32);
    for (int q = 0; q < 4; ++q)
        if ((rc = ensure(h, h->vt[q], bytes))) return rc;
    u32 *vcur = (u32*)h->vt[0].p, *vload = (u32*)h->vt[1].p, *vaff = (u32*)h->vt[2].p, *vnext = (u32*)h->vt[3].p;
    u32* assign = h->assign[h->cur];
    h->sb.fx = FxRows{};  // nobody reads the fix-up counters of the virtual-table solve: plain DevStats atomics, no pinned slot touched
    if ((rc = ensure_used(h))) return rc;
    const char* const who = dev_api ? "rio_gp_place_pending_dev" : "rio_gp_place_pending";
    if (h->part_mode != 2 && !host_io && pp_win_applicable(h->n, n, d_idx, d_req) &&
        (((uintptr_t)d_out | (uintptr_t)d_flag) & 15u) == 0) {  // (its output kernels store whole vectors)
        // Big batch: sorted by row window once, the row-side step out of LDS (k_pp_win_gather), the decisions written into the
        // real column by the solve itself: two random accesses per request instead of nine.  The entries are validated by the
        // binning kernel, which changes nothing.
        if ((rc = ensure(h, h->part, part_scratch_words(h->n, n) * sizeof(u32))) || (rc = ensure(h, h->vrec, n * sizeof(uint2)))) return rc;
        // An invalid entry makes the call fail with nothing changed, and nobody waits to find out: the binning kernel raises a
        // flag in mapped host memory and a device counter, every kernel enqueued behind it looks at the counter and does
        // nothing (k_scan solves an empty table), and the host reads the flag when it picks up the verdict.
        if ((rc = zero_stats(h))) return rc;
        u32* const h_bad = h->h_cs + h->cs_words + 2;
        *h_bad = 0;
        u32* const h_status = h->h_small + 4 * kSmallBatch;
        *h_status = 0;
        launch_pp_bin(h->n, h->m, d_idx, d_req, n, (u32*)h->part.p, h->dstats, h->d_cs + h->cs_words + 2, h->stream, h->dead_bits,
                      h->pp_claim);
        // The window kernel answers every request it can by itself — sticky hits, first touches on requesters that are active
        // members, later requests of an object — and records what the first touches ask of every requester; when every total
        // fits (k_pp_win_verdict) the answers are final and k_pp_win_split hands them out: no virtual table, no solve.
        launch_pp_win_gather(assign, h->load, h->n, h->m, h->alive_bits, n, (const u32*)h->part.p, vaff, vnext, h->dead_bits,
                             aff_life(h), h->dstats, h->pp_claim, h->stream);
        if (!h->all_alive)  // service.rs:227-237: every object of a dead node a request ran into is un-placed
            launch_clean(assign, h->n, h->m, h->dead_bits, h->used, h->dstats, h->stream, nullptr, nullptr, nullptr, aff_life(h));
        launch_pp_win_verdict(h->m, h->cap, h->alive_bits, h->used, h->pp_claim, h->dstats, h->pp_bad + 1, h->d_small + 4 * kSmallBatch,
                              h->stream);
        // (the answers' two words sit in vaff / vnext, in the sorted order, until here: the solve below writes vnext afterwards)
        launch_pp_win_unsort(h->n, (const u32*)h->part.p, vaff, vnext, n, (uint2*)h->vrec.p, d_out, d_flag, h->pp_bad + 1, h->stream);
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipGetLastError());
        if (*h_bad || *h_status == 3) return fail(h, RIO_GP_EINVAL, std::string(who) + ": object index or requester out of range (nothing was changed)");
        if (*h_status == 1) {  // final: `used` has taken the claims in place
            h->have_solved = false; ++h->mut_epoch;
            return RIO_GP_OK;
        }
        if (*h_status != 2) return fail(h, RIO_GP_EUPSTREAM, std::string(who) + ": the window kernels left no verdict");
        // The batch needs the solve (a requester would run full, a dead node was in the way, a requester is not an active
        // member): over the same records, the window kernel's placements standing as the optimistic ones.  The solve's kernels
        // read whole tiles of the object and requester columns: they get padded copies (the caller's arrays end where they end).
        if ((rc = ensure(h, h->stage[0], bytes)) || (rc = ensure(h, h->stage[1], bytes))) return rc;
        HIPCHK(h, hipMemcpyAsync(h->stage[0].p, d_idx, bytes, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->stage[1].p, d_req, bytes, hipMemcpyDeviceToDevice, h->stream));
        d_idx = (const u32*)h->stage[0].p;
        d_req = (const u32*)h->stage[1].p;
        Plan vp = hplan(h, n);
        const u64 seq = ++h->wait_seq;
        vp.mark = seq;
        Table vtab{vcur, vload, d_req /* the requesters ARE the affinity column of the virtual table */, vnext};
        vtab.pk_idx = d_idx;      // virtual row -> real row: every decision of the solve also goes to assign[d_idx[k]]
        vtab.real_next = assign;  // (in place: the rows that change are pending, nothing else reads them before the outputs)
        vtab.vrec = (const uint2*)h->vrec.p;  // the scan reads the records and writes vcur / vload for the kernels behind it
        vtab.prewritten = true;               // k_pp_win_gather has put the alive requesters' first touches into the column
        const NodeTab vnt{h->cap, h->alive_bits, h->used};
        h->sb.D = h->D;
        launch_scan(vp, vtab, vnt, h->sb, true, h->all_alive, h->stream);
        launch_resolve(vp, vnt, h->sb, slot_dev(h, 0), h->stream);
        enqueue_slow(h, vp, vtab, vnt, true, false);  // ahead of the verdict: its kernels guard themselves on the device
        launch_pp_win_output(d_idx, d_req, n, vcur, vload, vnext, h->alive_bits, h->sb.cutidx, h->m, d_out, d_flag, aff_life(h), h->dstats,
                             h->stream, h->sa, (const uint2*)h->vrec.p);
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipGetLastError());
        const DevStats v = reduce_slot(h, 0, h->m);
        const bool vslow = v.n_cut > 0 || v.spillcand > 0;
        std::swap(h->used, h->sb.used_cur);
        h->used_parts = vslow;
        h->parts_rounds = h->rounds;
        h->have_solved = false; ++h->mut_epoch;
        return RIO_GP_OK;
    }
    // The library's own (padded) copies of the requests, made by the first kernel on its way: requests in mapped host memory
    // are read over PCIe once — and a caller's device arrays end where they end, while the solve's kernels read whole tiles
    // (the requesters are the affinity column of the virtual table): past the end of an exact-size array that is somebody
    // else's page (found by the fuzz: a memory access fault at sizes a few hundred bytes short of a page).
    u32 *s_idx = nullptr, *s_req = nullptr, *vflag = nullptr;
    if ((rc = ensure(h, h->stage[0], bytes)) || (rc = ensure(h, h->stage[1], bytes))) return rc;
    s_idx = (u32*)h->stage[0].p; s_req = (u32*)h->stage[1].p;
    const bool mark = !h->all_alive;  // (1) service.rs:227-237 — requested rows on dead nodes trigger clean_server of those nodes
    if (mark) {
        if ((rc = ensure(h, h->stage[2], bytes))) return rc;
        vflag = (u32*)h->stage[2].p;
    }
    u32* const h_status = h->h_small + 4 * kSmallBatch;
    u32* const d_status = h->d_small + 4 * kSmallBatch;
    *h_status = 2;  // neither 0, 1 nor 3: the output kernel must write it
    launch_ppm_first(assign, h->n, h->m, h->alive_bits, d_idx, d_req, n, h->pos, s_idx, s_req, mark ? h->dead_bits : nullptr, vflag,
                     h->pp_bad, h->stream);
    const u32* const k_idx = s_idx;
    const u32* const k_req = s_req;
    (void)host_io;
    if (mark) launch_clean(assign, h->n, h->m, h->dead_bits, h->used, h->dstats, h->stream, nullptr, nullptr, nullptr, aff_life(h), 0, h->pp_bad);
    // (2)(3) the virtual table (rows = requests): the first request per row decides
    launch_ppm_gather(assign, h->load, k_idx, n, h->pos, vcur, vload, vaff /* position of the row's first request */, h->pp_bad, h->stream);
    // (4) solve it against the committed `used`
    Plan vp = hplan(h, n);
    const u64 mk = ++h->wait_seq;
    vp.mark = mk;
    Table vtab{vcur, vload, k_req /* the requesters ARE the affinity column of the virtual table */, vnext};
    vtab.skip_if = h->pp_bad;
    const NodeTab vnt{h->cap, h->alive_bits, h->used};  // (ensure_used above folded whatever the last solve's rounds had left)
    h->sb.D = h->D;
    launch_scan(vp, vtab, vnt, h->sb, true, h->all_alive, h->stream);
    launch_resolve(vp, vnt, h->sb, slot_dev(h, 0), h->stream);
    // the fix-up ahead of the verdict when the last batch needed it (its kernels guard themselves on the device); otherwise the
    // output kernel finds out on the device and hands back status 1 having changed nothing
    bool fixed = h->spec_mode != 2 && (h->spec_mode == 1 || h->pp_last_slow);
    if (fixed) enqueue_slow(h, vp, vtab, vnt, true, false);
    // (5) publish, outputs, scratch reset, completion word
    u32 seq = small_begin(h);
    launch_ppm_output(assign, h->n, k_idx, k_req, n, vcur, vnext, vaff, vflag, h->pos, h->alive_bits, h->sb, vp, d_out, d_flag,
                      aff_life(h), h->pp_bad, fixed, d_status, h->mid_ticket, small_done_dev(h), seq, h->stream);
    if ((rc = small_wait(h, seq))) return rc;
    if (*h_status == 1 && !fixed) {
        enqueue_slow(h, vp, vtab, vnt, true, false);
        fixed = true;
        *h_status = 2;
        seq = small_begin(h);
        launch_ppm_output(assign, h->n, k_idx, k_req, n, vcur, vnext, vaff, vflag, h->pos, h->alive_bits, h->sb, vp, d_out, d_flag,
                          aff_life(h), h->pp_bad, true, d_status, h->mid_ticket, small_done_dev(h), seq, h->stream);
        if ((rc = small_wait(h, seq))) return rc;
    }
    HIPCHK(h, hipGetLastError());
    if (*h_status == 3) return fail(h, RIO_GP_EINVAL, std::string(who) + ": object index or requester out of range (nothing was changed)");
    if (*h_status != 0) return fail(h, RIO_GP_EUPSTREAM, std::string(who) + ": the output kernel left no status");
    const DevStats v = reduce_slot(h, 0, h->m);  // (k_resolve's pinned rows landed before the completion word)
    const bool vslow = v.n_cut > 0 || v.spillcand > 0;
    h->pp_last_slow = vslow;
    std::swap(h->used, h->sb.used_cur);  // the solve's `used` vector becomes the committed one (as commit does): no copy
    h->used_parts = vslow && h->sb.D != nullptr;  // + what the water-fill rounds admitted (D rows), folded in later
    h->parts_rounds = h->rounds;
    h->have_solved = false; ++h->mut_epoch;
    return RIO_GP_OK;
}

// One pass over a caller's array: copy it into the mapped pinned row the kernels read, and find its largest entry on the way
// (a branch-free loop the compiler vectorises: the validation of a request batch costs no pass of its own and no early-exit
// branch per entry — 262 143 requests: a scalar check-then-memcpy was most of the call's host time).
static inline u32 copy_max(u32* __restrict__ dst, const u32* __restrict__ src, size_t n) {
    u32 mx = 0;
    for (size_t k = 0; k < n; ++k) {
        const u32 v = src[k];
        dst[k] = v;
        mx = v > mx ? v : mx;
    }
    return mx;
}
static inline u32 max_of(const u32* __restrict__ src, size_t n) {
    u32 mx = 0;
    for (size_t k = 0; k < n; ++k) mx = src[k] > mx ? src[k] : mx;
    return mx;
}

// rio_gp_place_pending with the handle locked; the entries are validated here (on their way into the pinned rows) unless
// `validated` says the caller has; micro_tried: the one-workgroup kernel has already run over this batch and handed it over
// untouched (rio_gp_mixed_batch)
static int place_pending_host_locked(rio_gp* h, uint64_t n, const uint32_t* idx, const uint32_t* requester, uint32_t* out_node,
                                     uint32_t* out_flag, bool micro_tried, bool validated = true) {
    static const char* const kRange = "rio_gp_place_pending: object index or requester out of range";
    flush_alive(h);
    int rc;
    if (!validated && n <= (uint64_t)kSmallBatch) {
        if (max_of(idx, n) >= h->n || max_of(requester, n) >= h->m) return fail(h, RIO_GP_EINVAL, kRange);
        validated = true;
    }
    if (n <= (uint64_t)kSmallBatch && !micro_tried) {
        // micro-batch: one workgroup, one launch, request/result arrays in mapped pinned memory (no staging copies)
        if ((rc = ensure_used(h))) return rc;
        u32 *hs = h->h_small, *ds = h->d_small;
        SmallInline inl;
        const bool in_args = small_inline(&inl, n, idx, requester);
        if (!in_args) {
            memcpy(hs, idx, n * sizeof(u32));
            memcpy(hs + kSmallBatch, requester, n * sizeof(u32));
        }
        hs[4 * kSmallBatch] = 2;  // neither 0 nor 1: the kernel must write it
        const u32 seq = small_begin(h);
        launch_pp_one(h->assign[h->cur], h->load, h->m, h->cap, h->alive_bits, h->used, ds, ds + kSmallBatch,
                      (u32)n, ds + 2 * kSmallBatch, ds + 3 * kSmallBatch, ds + 4 * kSmallBatch, h->stream, aff_life(h),
                      small_done_dev(h), seq, in_args ? &inl : nullptr, 0, nullptr, nullptr, h->sa);
        if ((rc = small_wait(h, seq))) return rc;
        const u32 status = hs[4 * kSmallBatch];
        if (status == 0) {
            memcpy(out_node, hs + 2 * kSmallBatch, n * sizeof(u32));
            if (out_flag) memcpy(out_flag, hs + 3 * kSmallBatch, n * sizeof(u32));
            h->have_solved = false; ++h->mut_epoch;
            return RIO_GP_OK;
        }
        if (status != 1) return fail(h, RIO_GP_EUPSTREAM, "rio_gp_place_pending: micro-batch kernel left no status");
        // status 1: a dead node / dead or full requester is involved — nothing was changed, take the general path
    }
    const size_t bytes = n * sizeof(u32);
    if (n <= (uint64_t)kMidBatch / 4) {
        // medium batch (<= 4 096 requests): requests and results in mapped pinned memory, the last kernel's last workgroup
        // stores the completion word: no staging copies, no stream wait
        u32* hm = h->h_mid;
        u32* dm = h->d_mid;
        const u32 mi = copy_max(hm, idx, n), mr = copy_max(hm + kMidBatch, requester, n);
        if (!validated && (mi >= h->n || mr >= h->m)) return fail(h, RIO_GP_EINVAL, kRange);
        if (n > (uint64_t)kSmallBatch) {
            // first the one-workgroup kernel (k_pp_one, 1024 threads x 4 requests): sticky hits and first touches that fit —
            // the whole call is ONE launch and one wait (4 096 requests: 46 -> ~15 us); anything heavier hands over untouched
            if ((rc = ensure_used(h))) return rc;
            h->h_small[4 * kSmallBatch] = 2;
            const u32 seq1 = small_begin(h);
            launch_pp_one(h->assign[h->cur], h->load, h->m, h->cap, h->alive_bits, h->used, dm, dm + kMidBatch, (u32)n,
                          dm + 2 * kMidBatch, dm + 3 * kMidBatch, h->d_small + 4 * kSmallBatch, h->stream, aff_life(h),
                          small_done_dev(h), seq1, nullptr, 0, h->pp_stage, h->mid_ticket, h->sa, true);
            if ((rc = small_wait(h, seq1))) return rc;
            const u32 status = h->h_small[4 * kSmallBatch];
            if (status == 0) {
                memcpy(out_node, hm + 2 * kMidBatch, bytes);
                if (out_flag) memcpy(out_flag, hm + 3 * kMidBatch, bytes);
                h->have_solved = false; ++h->mut_epoch;
                return RIO_GP_OK;
            }
            if (status != 1) return fail(h, RIO_GP_EUPSTREAM, "rio_gp_place_pending: one-workgroup kernel left no status");
        }
        if ((rc = place_pending_general(h, n, dm, dm + kMidBatch, dm + 2 * kMidBatch, dm + 3 * kMidBatch, true, false))) return rc;
        memcpy(out_node, hm + 2 * kMidBatch, bytes);
        if (out_flag) memcpy(out_flag, hm + 3 * kMidBatch, bytes);
        return RIO_GP_OK;
    }
    if (n <= (uint64_t)kReqBatch) {
        // up to 65 536 requests: requests and results in mapped pinned memory — the first kernel of the general path reads them
        // over PCIe once (and leaves device copies), the last one writes the results back; no staging copies through the runtime
        u32* hq = h->h_req;
        u32* dq = h->d_req;
        const u32 mi = copy_max(hq, idx, n), mr = copy_max(hq + kReqBatch, requester, n);
        if (!validated && (mi >= h->n || mr >= h->m)) return fail(h, RIO_GP_EINVAL, kRange);
        if ((rc = place_pending_general(h, n, dq, dq + kReqBatch, dq + 2 * kReqBatch, dq + 3 * kReqBatch, true, false))) return rc;
        memcpy(out_node, hq + 2 * kReqBatch, bytes);
        if (out_flag) memcpy(out_flag, hq + 3 * kReqBatch, bytes);
        return RIO_GP_OK;
    }
    // Bigger batches: the runtime's copies out of and into the caller's pageable arrays (staged through its own pinned buffers:
    // ~106 us per MB and direction on this driver, tools/reg_probe.py).  The entries are validated on the DEVICE (the first kernel
    // raises a word every later kernel looks at: an invalid entry changes nothing) — no pass of the host over 2 MB.
    // Round 6 registered the caller's four arrays for the duration of the call instead (hipHostRegister / hipHostUnregister:
    // 55 us per MB, 262 143 requests 249 us instead of ~370) — and two of eight processes of the parity test aborted inside a LATER,
    // unrelated hipMemcpy into a fresh numpy array (none of eight without the registration): the runtime does not survive user pages
    // that are registered, unregistered, freed and mapped again at the same address.  Removed (docs/LESSONS.md 39); a caller that
    // wants the copy engine's rate hands over device arrays (rio_gp_place_pending_dev) or its own pinned ones.
    for (int q = 0; q < 2; ++q)
        if ((rc = ensure(h, h->rq[q], bytes)) || (rc = ensure(h, h->rq[2 + q], bytes))) return rc;
    u32 *d_idx = (u32*)h->rq[0].p, *d_req = (u32*)h->rq[1].p, *d_out = (u32*)h->rq[2].p, *d_flag = (u32*)h->rq[3].p;
    HIPCHK(h, hipMemcpyAsync(d_idx, idx, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_req, requester, bytes, hipMemcpyHostToDevice, h->stream));
    if ((rc = place_pending_general(h, n, d_idx, d_req, d_out, d_flag, false, false))) return rc;
    HIPCHK(h, hipMemcpyAsync(out_node, d_out, bytes, hipMemcpyDeviceToHost, h->stream));
    if (out_flag) HIPCHK(h, hipMemcpyAsync(out_flag, d_flag, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return RIO_GP_OK;
}

int rio_gp_place_pending(rio_gp_t* h, uint64_t n, const uint32_t* idx, const uint32_t* requester,
                         uint32_t* out_node, uint32_t* out_flag) {
    if (!h || (n && (!idx || !requester || !out_node))) return RIO_GP_EINVAL;
    Locked g(h);
    if (!n) return RIO_GP_OK;
    if (n > 0x7FFFF000ull) return fail(h, RIO_GP_EINVAL, "rio_gp_place_pending: batch too large");
    HIPCHK(h, hipSetDevice(h->device));
    // (validated on the way into the pinned rows, before anything is enqueued: an invalid entry changes nothing)
    return place_pending_host_locked(h, n, idx, requester, out_node, out_flag, false, false);
}

// Up to kSmallBatch entries of EACH of update / remove / lookup / place_pending, executed in that order, as ONE launch and ONE
// host wait: one workgroup runs the parts one after the other (k_pp_one<256, 1> with the other parts in front of the requests,
// k_crud_small when there are no requests), a device-scope fence and a barrier between them, one completion word.  What the string layer's combiner sends when the callers of one generation
// asked for different things (a server's connections mix lookups, first touches and removals: service.rs:193-254,
// server.rs:292-304): one round trip instead of one per kind.
int rio_gp_mixed_batch(rio_gp_t* h, rio_gp_mixed* ops) {
    if (!h || !ops || ops->struct_size < sizeof(rio_gp_mixed)) return RIO_GP_EINVAL;
    const uint32_t nu = ops->n_update, nr = ops->n_remove, nl = ops->n_lookup, np = ops->n_place;
    if ((nu && (!ops->update_idx || !ops->update_node)) || (nr && !ops->remove_idx) ||
        (nl && (!ops->lookup_idx || !ops->lookup_out)) || (np && (!ops->place_idx || !ops->place_requester || !ops->place_node)))
        return RIO_GP_EINVAL;
    Locked g(h);
    if (nu > (uint32_t)kSmallBatch || nr > (uint32_t)kSmallBatch || nl > (uint32_t)kSmallBatch || np > (uint32_t)kSmallBatch)
        return fail(h, RIO_GP_EINVAL, "rio_gp_mixed_batch: at most 256 entries of each kind");
    // every kind is validated before anything is enqueued; a kind with an invalid entry is skipped as a whole (its own call
    // would have changed nothing either) and says so in rc[], the others run
    bool run[4] = {nu != 0, nr != 0, nl != 0, np != 0};
    for (int k = 0; k < 4; ++k) ops->rc[k] = RIO_GP_OK;
    for (uint32_t k = 0; k < nu && run[0]; ++k)
        if (ops->update_idx[k] >= h->n || (ops->update_node[k] != RIO_GP_NONE && ops->update_node[k] >= h->m)) {
            ops->rc[0] = fail(h, RIO_GP_EINVAL, "rio_gp_update_batch: index or node out of range");
            run[0] = false;
        }
    for (uint32_t k = 0; k < nr && run[1]; ++k)
        if (ops->remove_idx[k] >= h->n) {
            ops->rc[1] = fail(h, RIO_GP_EINVAL, "rio_gp_remove_batch: object index out of range");
            run[1] = false;
        }
    for (uint32_t k = 0; k < nl && run[2]; ++k)
        if (ops->lookup_idx[k] >= h->n) {
            ops->rc[2] = fail(h, RIO_GP_EINVAL, "rio_gp_lookup_batch: object index out of range");
            run[2] = false;
        }
    for (uint32_t k = 0; k < np && run[3]; ++k)
        if (ops->place_idx[k] >= h->n || ops->place_requester[k] >= h->m) {
            ops->rc[3] = fail(h, RIO_GP_EINVAL, "rio_gp_place_pending: object index or requester out of range");
            run[3] = false;
        }
    const int kinds = (int)run[0] + (int)run[1] + (int)run[2] + (int)run[3];
    if (!kinds) return RIO_GP_OK;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    // staging: the place_pending entries where its own call has them (h_small); update / remove / lookup in the first row of the
    // medium-batch area (mapped pinned memory as well) — [0] update idx | [1] update node | [2] remove idx | [3] lookup idx |
    // [4] lookup out
    u32 *hm = h->h_mid, *dm = h->d_mid;
    u32 *hs = h->h_small, *ds = h->d_small;
    const u32 seq = small_begin(h);
    if (kinds >= 2) {
        // ONE launch: the update / remove / lookup parts run in front of the requests inside the one-workgroup request kernel
        // (k_pp_one<256, 1>), or alone (k_crud_small) when nobody asked for a placement
        SmallInline iu, ir, il, ip;
        CrudSmallArgs c;
        c.n_obj = h->n; c.st = h->dstats;
        fold_used(h);
        if (run[3]) {
            flush_alive(h);
            if ((rc = ensure_used(h))) return rc;
        }
        if (run[0]) {
            c.nu = nu; c.u_idx = dm; c.u_node = dm + kSmallBatch;
            if (small_inline(&iu, nu, ops->update_idx, ops->update_node)) c.u_inl = &iu;
            else {
                memcpy(hm, ops->update_idx, nu * sizeof(u32));
                memcpy(hm + kSmallBatch, ops->update_node, nu * sizeof(u32));
            }
        }
        if (run[1]) {
            c.nr = nr; c.r_idx = dm + 2 * kSmallBatch;
            if (small_inline(&ir, nr, ops->remove_idx, nullptr)) c.r_inl = &ir;
            else memcpy(hm + 2 * kSmallBatch, ops->remove_idx, nr * sizeof(u32));
        }
        if (run[2]) {
            c.nl = nl; c.l_idx = dm + 3 * kSmallBatch; c.l_out = dm + 4 * kSmallBatch;
            if (small_inline(&il, nl, ops->lookup_idx, nullptr)) c.l_inl = &il;
            else memcpy(hm + 3 * kSmallBatch, ops->lookup_idx, nl * sizeof(u32));
        }
        if (run[0] || run[1] || run[3]) { h->have_solved = false; ++h->mut_epoch; }
        if (run[3]) {
            const bool in_args = small_inline(&ip, np, ops->place_idx, ops->place_requester);
            if (!in_args) {
                memcpy(hs, ops->place_idx, np * sizeof(u32));
                memcpy(hs + kSmallBatch, ops->place_requester, np * sizeof(u32));
            }
            hs[4 * kSmallBatch] = 2;  // neither 0 nor 1: the kernel must write it
            launch_pp_one(h->assign[h->cur], h->load, h->m, h->cap, h->alive_bits, h->used, ds, ds + kSmallBatch, np,
                          ds + 2 * kSmallBatch, ds + 3 * kSmallBatch, ds + 4 * kSmallBatch, h->stream, aff_life(h), small_done_dev(h), seq,
                          in_args ? &ip : nullptr, 0, nullptr, nullptr, h->sa, false, &c);
        } else {
            launch_crud_small(c, h->assign[h->cur], h->load, h->m, h->used_valid ? h->used : nullptr, aff_life(h), small_done_dev(h), seq,
                              h->stream);
        }
    } else {
    SmallInline inl;
    if (run[0]) {
        const bool in_args = small_inline(&inl, nu, ops->update_idx, ops->update_node);
        if (!in_args) {
            memcpy(hm, ops->update_idx, nu * sizeof(u32));
            memcpy(hm + kSmallBatch, ops->update_node, nu * sizeof(u32));
        }
        fold_used(h);
        launch_update_small(h->assign[h->cur], dm, dm + kSmallBatch, nu, h->stream, aff_life(h), small_done_dev(h),
                            seq, in_args ? &inl : nullptr, h->used_valid ? h->used : nullptr, h->load, h->m);
        h->have_solved = false; ++h->mut_epoch;
    }
    if (run[1]) {
        const bool in_args = small_inline(&inl, nr, ops->remove_idx, nullptr);
        if (!in_args) memcpy(hm + 2 * kSmallBatch, ops->remove_idx, nr * sizeof(u32));
        fold_used(h);
        launch_remove_small(h->assign[h->cur], h->m, h->load, dm + 2 * kSmallBatch, nr, h->used_valid ? h->used : nullptr, h->stream,
                            aff_life(h), small_done_dev(h), seq, in_args ? &inl : nullptr);
        h->have_solved = false; ++h->mut_epoch;
    }
    if (run[2]) {
        const bool in_args = small_inline(&inl, nl, ops->lookup_idx, nullptr);
        if (!in_args) memcpy(hm + 3 * kSmallBatch, ops->lookup_idx, nl * sizeof(u32));
        launch_lookup_small(h->assign[h->cur], h->n, dm + 3 * kSmallBatch, nl, dm + 4 * kSmallBatch, h->dstats, h->stream,
                            small_done_dev(h), seq, in_args ? &inl : nullptr);
    }
    if (run[3]) {
        flush_alive(h);
        if ((rc = ensure_used(h))) return rc;
        const bool in_args = small_inline(&inl, np, ops->place_idx, ops->place_requester);
        if (!in_args) {
            memcpy(hs, ops->place_idx, np * sizeof(u32));
            memcpy(hs + kSmallBatch, ops->place_requester, np * sizeof(u32));
        }
        hs[4 * kSmallBatch] = 2;  // neither 0 nor 1: the kernel must write it
        launch_pp_one(h->assign[h->cur], h->load, h->m, h->cap, h->alive_bits, h->used, ds, ds + kSmallBatch, np,
                      ds + 2 * kSmallBatch, ds + 3 * kSmallBatch, ds + 4 * kSmallBatch, h->stream, aff_life(h), small_done_dev(h), seq,
                      in_args ? &inl : nullptr, 0, nullptr, nullptr, h->sa);
    }
    }
    if ((rc = small_wait(h, seq))) return rc;
    if (run[2]) memcpy(ops->lookup_out, hm + 4 * kSmallBatch, nl * sizeof(u32));  // (stored by the kernel whose fence precedes the word)
    if (run[3]) {
        const u32 status = hs[4 * kSmallBatch];
        if (status == 0) {
            memcpy(ops->place_node, hs + 2 * kSmallBatch, np * sizeof(u32));
            if (ops->place_flag) memcpy(ops->place_flag, hs + 3 * kSmallBatch, np * sizeof(u32));
            h->have_solved = false; ++h->mut_epoch;
        } else if (status == 1) {
            // a dead node / a dead or full requester is involved: nothing of the place_pending part was changed, the general path
            ops->rc[3] = place_pending_host_locked(h, np, ops->place_idx, ops->place_requester, ops->place_node, ops->place_flag, true);
        } else {
            return fail(h, RIO_GP_EUPSTREAM, "rio_gp_mixed_batch: micro-batch kernel left no status");
        }
    }
    return RIO_GP_OK;
}

int rio_gp_place_pending_dev(rio_gp_t* h, uint64_t n, const uint32_t* d_idx, const uint32_t* d_requester,
                             uint32_t* d_out_node, uint32_t* d_out_flag) {
    if (!h || (n && (!d_idx || !d_requester || !d_out_node))) return RIO_GP_EINVAL;
    Locked g(h);
    if (!n) return RIO_GP_OK;
    if (n > 0x7FFFF000ull) return fail(h, RIO_GP_EINVAL, "rio_gp_place_pending_dev: batch too large");
    // an empty table (or no nodes): every entry is out of range, and the one-workgroup kernel's "0 rows = the host has
    // validated the entries" convention must not be reached with entries nobody has looked at
    if (h->n == 0 || h->m == 0)
        return fail(h, RIO_GP_EINVAL, "rio_gp_place_pending_dev: object index or requester out of range (nothing was changed)");
    HIPCHK(h, hipSetDevice(h->device));
    flush_alive(h);
    int rc;
    if (n <= (uint64_t)kMidBatch / 4 && h->n < (1ull << 31) &&
        (((uintptr_t)d_idx | (uintptr_t)d_requester | (uintptr_t)d_out_node | (uintptr_t)d_out_flag) & 15u) == 0) {
        // up to 4 096 requests: the one-workgroup kernel first, reading the caller's device arrays in place (it validates the
        // entries itself) — ONE launch and one wait when the batch is sticky hits and first touches that fit
        if ((rc = ensure_used(h))) return rc;
        h->h_small[4 * kSmallBatch] = 2;
        const u32 seq = small_begin(h);
        launch_pp_one(h->assign[h->cur], h->load, h->m, h->cap, h->alive_bits, h->used, d_idx, d_requester, (u32)n, d_out_node,
                      d_out_flag ? d_out_flag : h->d_mid + 3 * kMidBatch, h->d_small + 4 * kSmallBatch, h->stream, aff_life(h),
                      small_done_dev(h), seq, nullptr, (u32)h->n, h->pp_stage, h->mid_ticket, h->sa);
        if ((rc = small_wait(h, seq))) return rc;
        const u32 status = h->h_small[4 * kSmallBatch];
        if (status == 0) {
            h->have_solved = false; ++h->mut_epoch;
            return RIO_GP_OK;
        }
        if (status == 3)
            return fail(h, RIO_GP_EINVAL, "rio_gp_place_pending_dev: object index or requester out of range (nothing was changed)");
        if (status != 1) return fail(h, RIO_GP_EUPSTREAM, "rio_gp_place_pending_dev: one-workgroup kernel left no status");
    }
    return place_pending_general(h, n, d_idx, d_requester, d_out_node, d_out_flag, false, true);
}

int rio_gp_solve(rio_gp_t* h, rio_gp_stats* stats) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    HIPCHK(h, hipSetDevice(h->device));
    return solve_locked(h, stats);
}

int rio_gp_commit(rio_gp_t* h) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    HIPCHK(h, hipSetDevice(h->device));
    int rc = commit_locked(h);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RIO_GP_OK;
}

int rio_gp_tick(rio_gp_t* h, rio_gp_stats* stats) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    HIPCHK(h, hipSetDevice(h->device));
    return solve_locked(h, stats, true);
}

int rio_gp_tick_async(rio_gp_t* h) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h, false);  // (a quiet tick does not wait for the previous tick's k_resolve: tick_async_locked)
    HIPCHK(h, hipSetDevice(h->device));
    return tick_async_locked(h);
}

int rio_gp_tick_wait(rio_gp_t* h, rio_gp_stats* out, uint32_t cap, uint32_t* n_out) {
    if (!h || !n_out || (cap && !out)) return RIO_GP_EINVAL;
    Locked g(h);
    HIPCHK(h, hipSetDevice(h->device));
    int rc = harvest_ticks(h);
    if (rc) return rc;
    const size_t n = h->tick_done.size();
    const size_t take = n < cap ? n : cap;
    for (size_t k = 0; k < take; ++k) out[k] = h->tick_done[n - take + k];  // the most recent `take`, oldest first
    *n_out = (uint32_t)n;
    h->tick_done.clear();
    return RIO_GP_OK;
}

int rio_gp_solve_async(rio_gp_t* h) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    HIPCHK(h, hipSetDevice(h->device));  // a host with several handles (one per GPU) calls from any thread
    if (h->sh_tick_n) return fail(h, RIO_GP_EINVAL, "rio_gp_solve_async: row-sharded ticks are in flight (call rio_gp_shard_tick_wait)");
    // the verdict ring holds kRing solves: fold the oldest slot's verdict into the running count before it is overwritten
    // (rio_gp_solve_wait reports how many of ALL the solves since the last wait took the fix-up path)
    if (h->ring_n >= (u32)kRing) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        for (u32 k = 0; k < h->ring_n; ++k) {
            const DevStats v = reduce_slot(h, k, h->m);
            h->ring_slow += (v.n_cut > 0 || v.spillcand > 0);
        }
        h->ring_n = 0;
    }
    reset_inplace(h);
    h->plan = hplan(h, h->n);
    use_fx_slot(h, 0);
    const Table t = real_table(h);
    const NodeTab nt = scan_nodes(h);
    h->ca_now = cut_apply_for(h, false);
    enqueue_scan_resolve(h, t, nt, false, slot_dev(h, h->ring_n));
    HIPCHK(h, hipGetLastError());
    h->ring_n++;
    h->ring_any = true;
    h->have_solved = false; ++h->mut_epoch;
    return RIO_GP_OK;
}

int rio_gp_solve_wait(rio_gp_t* h, rio_gp_stats* stats, uint32_t* n_slow) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    if (!h->ring_any) return fail(h, RIO_GP_EINVAL, "rio_gp_solve_wait: nothing enqueued");
    uint32_t slow = h->ring_slow;  // solves whose slots were recycled (rio_gp_solve_async folds them in)
    DevStats last;
    memset(&last, 0, sizeof last);
    if (h->ring_n == 0) last = reduce_slot(h, kRing - 1, h->m);  // exactly a multiple of kRing: the last solve sits in the last slot
    for (u32 k = 0; k < h->ring_n; ++k) {
        last = reduce_slot(h, k, h->m);
        slow += (last.n_cut > 0 || last.spillcand > 0);
    }
    h->ring_slow = 0;
    h->ring_any = false;
    if (last.n_cut > 0 || last.spillcand > 0) {
        enqueue_slow(h, h->plan, real_table(h), real_nodes(h), false, false);
        int rc = merge_slow(h, &last);
        if (rc) return rc;
        HIPCHK(h, hipGetLastError());
    }
    fill_stats(last, h->n, stats);
    if (n_slow) *n_slow = slow;
    h->ring_n = 0;
    h->have_solved = true;
    return RIO_GP_OK;
}

int rio_gp_solve_profiled(rio_gp_t* h, float* scan_ms, float* resolve_ms) {
    if (!h || !scan_ms || !resolve_ms) return RIO_GP_EINVAL;
    Locked g(h);
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->ev2) { HIPCHK(h, hipEventCreate(&h->ev2)); HIPCHK(h, hipEventCreate(&h->ev3)); }
    reset_inplace(h);
    h->plan = hplan(h, h->n);
    use_fx_slot(h, 0);
    const Table t = real_table(h);
    const NodeTab nt = scan_nodes(h);
    // hipExtLaunchKernelGGL start/stop events = the dispatch's own begin/end timestamps
    fold_used(h);
    h->sb.D = h->D;
    h->solve_used_D = h->sb.D != nullptr;
    launch_scan(h->plan, t, nt, h->sb, false, h->all_alive, h->stream, h->ev0, h->ev1);
    launch_resolve(h->plan, nt, h->sb, slot_dev(h, 0), h->stream, h->ev2, h->ev3);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventElapsedTime(scan_ms, h->ev0, h->ev1));
    HIPCHK(h, hipEventElapsedTime(resolve_ms, h->ev2, h->ev3));
    h->have_solved = false; ++h->mut_epoch;
    h->ring_n = 0; h->ring_slow = 0; h->ring_any = false;
    const DevStats v = reduce_slot(h, 0, h->m);
    if (v.n_cut > 0 || v.spillcand > 0)
        return fail(h, RIO_GP_EINVAL, "rio_gp_solve_profiled: this table needs the cut/spill fix-up");
    return RIO_GP_OK;
}

// ---- row-sharded solve across GPUs ------------------------------------------

static int p2p_check(rio_gp* h);

int rio_gp_set_stream(rio_gp_t* h, void* hip_stream) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return RIO_GP_OK;
}

uint32_t rio_gp_shard_words1(rio_gp_t* h) { return h ? (uint32_t)shard_words1(h->m) : 0; }
uint32_t rio_gp_shard_words2(rio_gp_t* h) { return h ? (uint32_t)shard_words2(h->m) : 0; }

static SolveBufs local_bufs(rio_gp* h) {  // where the local sums of a shard scan go
    SolveBufs b = h->sb;
    b.RP = nullptr;
    b.R = nullptr;  // the row-sharded solve keeps the cut step and the rounds apart (the Y exchange sits between them):
    b.D = nullptr;  // no per-block rejected loads, admitted load straight into used_cur, a ranking launch per round
    b.used_kept = h->sh_lkept;
    b.claim_tot = h->sh_lclaim;
    b.used_cur = h->sh_lcur;
    b.cutblk = h->sh_lcutblk;
    b.cutidx = h->sh_lcutidx;
    return b;
}

static SolveBufs shard_bufs(rio_gp* h) {
    SolveBufs b = h->sb;
    b.RP = nullptr;
    b.R = nullptr;
    b.D = nullptr;
    b.used_snap = h->sh_gprev;  // the global `used` as the last exchange left it: what a round orders the nodes by
    b.forced_bits = h->sh_forced;
    b.rank_base = h->sh_rank_base;
    b.pending_global = h->sh_verdict;  // [0] = rows pending on all ranks, written by k_shard_import_delta
    return b;
}

int rio_gp_shard_scan(rio_gp_t* h, uint64_t* d_x) {
    if (!h || !d_x) return RIO_GP_EINVAL;
    Locked g(h);
    if (h->sa) return fail(h, RIO_GP_EINVAL, "row-sharded solves do not implement RIO_GP_CFG_REF_SELF_ASSIGN (single-GPU handles only)");
    reset_inplace(h);
    h->plan = hplan(h, h->n);
    h->sb.fx = FxRows{};  // row-sharded solve: the fix-up counters are summed in DevStats (rio_gp_shard_finish reads them)
    fold_used(h);
    h->solve_used_D = false;
    const Table t = real_table(h);
    const NodeTab nt = real_nodes(h);
    launch_scan(h->plan, t, nt, h->sb, false, h->all_alive, h->stream);
    const SolveBufs lb = local_bufs(h);
    launch_resolve(h->plan, nt, lb, nullptr, h->stream);  // used_base = nullptr: purely local sums
    launch_shard_pack1(h->plan, lb, reinterpret_cast<u64*>(d_x), h->stream);
    h->have_solved = false; ++h->mut_epoch;
    h->sh_state = 1;
    return RIO_GP_OK;
}

int rio_gp_shard_resolve(rio_gp_t* h, uint32_t rank, uint32_t n_ranks, const uint64_t* d_xg, void* on_stream) {
    if (!h || !d_xg || n_ranks == 0 || rank >= n_ranks) return RIO_GP_EINVAL;
    Locked g(h);
    if (h->sh_state != 1) return fail(h, RIO_GP_EINVAL, "rio_gp_shard_resolve: call rio_gp_shard_scan first");
    h->sh_rank = rank;
    h->sh_R = n_ranks;
    h->sh_slot = h->ring_n;
    h->sh_rows = 1;
    hipStream_t st = on_stream ? static_cast<hipStream_t>(on_stream) : h->stream;
    h->sh_side = on_stream ? st : nullptr;
    launch_shard_import(h->plan, real_nodes(h), shard_bufs(h), reinterpret_cast<const u64*>(d_xg), rank, n_ranks,
                        h->sh_gprev, h->sh_gfinal, h->sh_verdict, slot_dev(h, h->ring_n), st);
    h->ring_n++;
    h->sh_state = 2;
    return RIO_GP_OK;
}

int rio_gp_shard_verdict(rio_gp_t* h, rio_gp_shard_info* out, uint32_t* n_slow) {
    if (!h || !out) return RIO_GP_EINVAL;
    Locked g(h);
    if (h->sh_state != 2) return fail(h, RIO_GP_EINVAL, "rio_gp_shard_verdict: call rio_gp_shard_resolve first");
    HIPCHK(h, hipSetDevice(h->device));
    if (h->sh_side) HIPCHK(h, hipStreamSynchronize(h->sh_side));  // the exchange stream of a pipelined caller
    h->sh_side = nullptr;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->p2p && h->p2p->d_peers) { int rc = p2p_check(h); if (rc) return rc; }
    HIPCHK(h, hipGetLastError());
    uint32_t slow = 0;
    const u32 lo = h->ring_n > (u32)kRing ? h->ring_n - kRing : 0;
    auto fold = [&](u32 k, u64* x) {  // one verdict row (k_shard_import) or one partial row per workgroup (k_resolve_xchg)
        const u64* rows = h->h_slots + (size_t)(k % kRing) * h->slot_rows * 8;
        for (int c = 0; c < 8; ++c) x[c] = 0;
        for (u32 r = 0; r < h->sh_rows; ++r)
            for (int c = 0; c < 8; ++c) x[c] += rows[(size_t)r * 8 + c];
    };
    u64 x[8];
    for (u32 k = lo; k < h->ring_n; ++k) {
        fold(k, x);
        slow += (x[0] > 0 || x[1] > 0);
    }
    fold(h->sh_slot, x);
    out->cut_nodes = x[0]; out->spill_rows = x[1]; out->local_fixup = x[2]; out->kept = x[3];
    out->evicted = x[4]; out->claimants = x[5]; out->load_kept = x[6]; out->load_claim = x[7];
    if (n_slow) *n_slow = slow;
    h->sh_slow = (x[0] > 0 || x[1] > 0);
    h->ring_n = 0;
    if (!h->sh_slow)  // fast path: the committed `used` is the global kept + claimed load
        HIPCHK(h, hipMemcpyAsync(h->sb.used_cur, h->sh_gfinal, (size_t)(h->m ? h->m : 1) * sizeof(u64),
                                 hipMemcpyDeviceToDevice, h->stream));
    return RIO_GP_OK;
}

int rio_gp_shard_cut(rio_gp_t* h, int run_local_fixup, uint64_t* d_y) {
    if (!h || !d_y) return RIO_GP_EINVAL;
    Locked g(h);
    if (h->sh_state != 2 || !h->sh_slow) return fail(h, RIO_GP_EINVAL, "rio_gp_shard_cut: no fix-up pending");
    HIPCHK(h, hipSetDevice(h->device));
    const SolveBufs b = shard_bufs(h);
    if (run_local_fixup) {
        launch_cut_find(h->plan, real_table(h), real_nodes(h), b, false, h->stream, false);
        launch_fill(h->plan, real_table(h), real_nodes(h), b, false, true, false, 0, false, h->stream);
    }
    launch_shard_export_delta(h->plan, b, h->sb.used_kept, 0, reinterpret_cast<u64*>(d_y), h->stream);
    h->sh_state = 3;
    return RIO_GP_OK;
}

int rio_gp_shard_merge(rio_gp_t* h, const uint64_t* d_yg, uint64_t* pending_rows, uint64_t* pending_load) {
    if (!h || !d_yg) return RIO_GP_EINVAL;
    Locked g(h);
    if (h->sh_state != 3 && h->sh_state != 5) return fail(h, RIO_GP_EINVAL, "rio_gp_shard_merge: nothing exported");
    HIPCHK(h, hipSetDevice(h->device));
    launch_shard_import_delta(h->plan, shard_bufs(h), reinterpret_cast<const u64*>(d_yg), h->sh_rank, h->sh_R,
                              h->sh_gprev, h->sh_verdict, slot_dev(h, 0), h->stream);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    const u64* x = h->h_slots;
    if (pending_rows) *pending_rows = x[0];
    if (pending_load) *pending_load = x[1];
    h->sh_state = 4;
    return RIO_GP_OK;
}

int rio_gp_shard_spill(rio_gp_t* h, uint32_t round, int last, uint64_t* d_y) {
    if (!h || !d_y) return RIO_GP_EINVAL;
    Locked g(h);
    if (h->sh_state != 4) return fail(h, RIO_GP_EINVAL, "rio_gp_shard_spill: call rio_gp_shard_merge first");
    HIPCHK(h, hipSetDevice(h->device));
    const SolveBufs b = shard_bufs(h);
    launch_fill(h->plan, real_table(h), real_nodes(h), b, false, false, true, (int)round, last != 0, h->stream);
    launch_shard_export_delta(h->plan, b, h->sh_gprev, (int)((round & 1) ^ 1), reinterpret_cast<u64*>(d_y), h->stream);
    h->sh_state = 5;
    return RIO_GP_OK;
}

int rio_gp_shard_finish(rio_gp_t* h, rio_gp_stats* local_stats) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    if (h->sh_state != 2 && h->sh_state != 4)
        return fail(h, RIO_GP_EINVAL, "rio_gp_shard_finish: solve not resolved / last exchange not merged");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(h->h_stats, h->dstats, sizeof(DevStats), hipMemcpyDeviceToHost, h->stream));
    // local row counters of this shard: fold k_resolve's per-workgroup partial rows on the host
    std::vector<u64> part((size_t)resolve_blocks(h->m) * 8);
    HIPCHK(h, hipMemcpyAsync(part.data(), h->sb.partial, part.size() * sizeof(u64), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    DevStats v;
    memset(&v, 0, sizeof v);
    for (size_t r = 0; r < part.size() / 8; ++r) {
        const u64* x = part.data() + r * 8;
        v.load_kept += x[0]; v.load_claim_tot += x[1];
        v.kept += x[3]; v.evicted += x[4]; v.claimants += x[5]; v.spillcand += x[6];
    }
    if (h->sh_slow) {
        const DevStats& d = h->h_stats[0];
        v.rejected = d.rejected; v.load_rejected = d.load_rejected;
        v.spilled = d.spilled; v.load_spilled = d.load_spilled;
        v.unplaced = d.unplaced; v.load_unplaced = d.load_unplaced;
    }
    fill_stats(v, h->n, local_stats);  // cut_nodes / slow_path / rounds_run are global: the caller has them
    h->have_solved = true;
    h->ring_n = 0;
    h->sh_state = 0;
    return RIO_GP_OK;
}

// ---- peer-to-peer exchange over xGMI (preferred): no collective call on the data path at all ----

int rio_gp_shard_p2p_export(rio_gp_t* h, uint32_t n_ranks, void* out_handle64) {
    if (!h || !out_handle64 || n_ranks == 0 || n_ranks > 32) return RIO_GP_EINVAL;
    Locked g(h);
    if (h->p2p) return fail(h, RIO_GP_EINVAL, "rio_gp_shard_p2p_export: window already exported");
    HIPCHK(h, hipSetDevice(h->device));
    P2P* q = new P2P();
    q->R = n_ranks;
    q->W = (shard_words1(h->cap_nodes) + 7) & ~(size_t)7;
    q->Wx = (shard_xchg_words(h->cap_nodes) + 7) & ~(size_t)7;
    const size_t bytes = q->total_words() * sizeof(u64);
    void* w = nullptr;
    // uncached first (what RCCL uses for its own flag/LL buffers on gfx94x/95x), fine-grained second; ordinary cached
    // device memory is NOT acceptable: a peer's store would sit behind this GPU's stale L2 lines
    if (hipExtMallocWithFlags(&w, bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        w = nullptr;
        if (hipExtMallocWithFlags(&w, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            delete q;
            return fail(h, RIO_GP_EUPSTREAM, "rio_gp_shard_p2p_export: no uncached / fine-grained device memory");
        }
    }
    q->win = static_cast<u64*>(w);
    h->p2p = q;
    HIPCHK(h, hipMemsetAsync(q->win, 0, bytes, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    hipIpcMemHandle_t ih;
    HIPCHK(h, hipIpcGetMemHandle(&ih, q->win));
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    memcpy(out_handle64, &ih, 64);
    return RIO_GP_OK;
}

int rio_gp_shard_p2p_connect(rio_gp_t* h, uint32_t rank, uint32_t n_ranks, const void* handles) {
    if (!h || !handles || n_ranks == 0 || rank >= n_ranks) return RIO_GP_EINVAL;
    Locked g(h);
    P2P* q = h->p2p;
    if (!q || q->R != n_ranks || q->d_peers) return fail(h, RIO_GP_EINVAL, "rio_gp_shard_p2p_connect: export first / rank count differs");
    HIPCHK(h, hipSetDevice(h->device));
    q->rank = rank;
    q->opened.assign(n_ranks, nullptr);
    std::vector<u64*> bases(n_ranks, nullptr);
    for (uint32_t r = 0; r < n_ranks; ++r) {
        if (r == rank) { bases[r] = q->win; continue; }
        hipIpcMemHandle_t ih;
        memcpy(&ih, static_cast<const char*>(handles) + (size_t)r * 64, 64);
        void* o = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&o, ih, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(h, RIO_GP_EUPSTREAM, std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e));
        }
        q->opened[r] = o;
        bases[r] = static_cast<u64*>(o);
    }
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&q->d_peers), n_ranks * sizeof(u64*)));
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&q->d_err), sizeof(u64)));
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&q->scratch), q->W * sizeof(u64)));
    HIPCHK(h, hipMemcpy(q->d_peers, bases.data(), n_ranks * sizeof(u64*), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemset(q->d_err, 0, sizeof(u64)));
    // handshake: every rank stores a token into every peer's hello line and waits for all of theirs (3 s limit).  Next to
    // the token travels the identity of the device the rank runs on (hash of its PCI bus id): ranks that share a GPU — a test
    // box, or a deployment that packs several shards on one device — must keep their spinning exchange kernels small enough
    // to be co-resident (launch_resolve_xchg).
    const u64 token = 0xC0FFEE0000000001ull;
    u64 devid = 1469598103934665603ull;
    {
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, (int)sizeof bus - 1, h->device) != hipSuccess) {
            (void)hipGetLastError();
            snprintf(bus, sizeof bus, "device-%d", h->device);
        }
        for (const char* c = bus; *c; ++c) devid = (devid ^ (u64)(unsigned char)*c) * 1099511628211ull;
        devid |= 1ull;
    }
    HIPCHK(h, hipMemcpyAsync(q->scratch, &devid, sizeof devid, hipMemcpyHostToDevice, h->stream));
    launch_p2p_put(q->scratch, 1, q->d_peers, n_ranks, q->hello_off(rank) + 1, q->hello_off(rank), token, h->stream);
    launch_p2p_wait_copy(q->win, q->W, n_ranks, 0, q->win + q->hello_off(0), token, q->d_err, nullptr, h->stream);
    u64 err = 0;
    std::vector<u64> hello((size_t)n_ranks * 8, 0);
    HIPCHK(h, hipMemcpyAsync(&err, q->d_err, sizeof err, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(hello.data(), q->win + q->hello_off(0), hello.size() * sizeof(u64), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    if (err) return fail(h, RIO_GP_EUPSTREAM, "rio_gp_shard_p2p_connect: a peer's handshake store never became visible");
    q->co_resident = 0;
    for (uint32_t r = 0; r < n_ranks; ++r) q->co_resident += hello[(size_t)r * 8 + 1] == devid;
    if (q->co_resident == 0) q->co_resident = 1;
    return RIO_GP_OK;
}

int rio_gp_shard_p2p_ready(rio_gp_t* h) { return h && h->p2p && h->p2p->d_peers ? 1 : 0; }

int rio_gp_shard_p2p_close(rio_gp_t* h) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    (void)hipGetLastError();
    p2p_free(h);
    h->sh_state = 0;
    h->ring_n = 0;
    return RIO_GP_OK;
}

static int p2p_check(rio_gp* h) {  // after a wait on the stream: did any in-kernel wait time out?
    u64 err = 0;
    HIPCHK(h, hipMemcpyAsync(&err, h->p2p->d_err, sizeof err, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (err) {
        // a peer never delivered (it is gone, or minutes behind): whatever was enqueued behind the wait ran on records that are
        // not there, and this rank's sequence numbers are ahead of anything the peers will see — the session is over
        h->p2p->out_of_step = true;
        return fail(h, RIO_GP_EUPSTREAM, "peer-to-peer exchange timed out waiting for a rank's record (the session is out of step: "
                                         "rio_gp_shard_p2p_close, then fresh windows or another exchange path)");
    }
    return RIO_GP_OK;
}

// ---- native RCCL exchange (optional): the library issues the all-gathers itself --------------

static bool rccl_load(RcclApi* a, const char* path, std::string* err) {
    const char* cands[] = {path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int pass = 0; pass < 2 && !a->lib; ++pass)      // pass 0: a copy already in the process (torch's), pass 1: load
        for (const char* c : cands)
            if (c && *c && (a->lib = dlopen(c, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0)))) break;
    if (!a->lib) { *err = std::string("dlopen(librccl) failed: ") + (dlerror() ? dlerror() : "?"); return false; }
    a->GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(a->lib, "ncclGetUniqueId"));
    a->CommInitRank = reinterpret_cast<int (*)(void**, int, RioGpNcclId, int)>(dlsym(a->lib, "ncclCommInitRank"));
    a->CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(a->lib, "ncclCommDestroy"));
    a->AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(a->lib, "ncclAllGather"));
    a->GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(a->lib, "ncclGetErrorString"));
    if (!a->GetUniqueId || !a->CommInitRank || !a->CommDestroy || !a->AllGather) { *err = "librccl lacks a required symbol"; return false; }
    return true;
}

int rio_gp_shard_comm_unique_id(void* out128, const char* rccl_path) {
    if (!out128) return RIO_GP_EINVAL;
    RcclApi a;
    std::string err;
    if (!rccl_load(&a, rccl_path, &err)) { g_create_error = err; return RIO_GP_EUPSTREAM; }
    RioGpNcclId id;
    memset(&id, 0, sizeof id);
    const int rc = a.GetUniqueId(&id);
    if (rc != 0) { g_create_error = "ncclGetUniqueId failed"; return RIO_GP_EUPSTREAM; }
    memcpy(out128, &id, sizeof id);
    return RIO_GP_OK;
}

int rio_gp_shard_comm_init(rio_gp_t* h, uint32_t rank, uint32_t n_ranks, const void* id128, const char* rccl_path) {
    if (!h || !id128 || n_ranks == 0 || rank >= n_ranks) return RIO_GP_EINVAL;
    Locked g(h);
    if (h->sc) return fail(h, RIO_GP_EINVAL, "rio_gp_shard_comm_init: communicator already set up");
    HIPCHK(h, hipSetDevice(h->device));
    ShardComm* sc = new ShardComm();
    std::string err;
    if (!rccl_load(&sc->api, rccl_path, &err)) { delete sc; return fail(h, RIO_GP_EUPSTREAM, err); }
    RioGpNcclId id;
    memcpy(&id, id128, sizeof id);
    const int rc = sc->api.CommInitRank(&sc->comm, (int)n_ranks, id, (int)rank);
    if (rc != 0) {
        const std::string m = std::string("ncclCommInitRank: ") + (sc->api.GetErrorString ? sc->api.GetErrorString(rc) : "failed");
        delete sc;
        return fail(h, RIO_GP_EUPSTREAM, m);
    }
    sc->rank = rank;
    sc->R = n_ranks;
    h->sc = sc;  // from here on rio_gp_destroy releases whatever was created
    HIPCHK(h, hipStreamCreateWithFlags(&sc->side, hipStreamNonBlocking));
    const size_t w1 = shard_words1(h->cap_nodes);
    for (int q = 0; q < kShardRing; ++q) {
        HIPCHK(h, hipEventCreateWithFlags(&sc->ready[q], hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&sc->done[q], hipEventDisableTiming));
        HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&sc->X[q]), w1 * sizeof(u64)));
        HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&sc->XG[q]), w1 * n_ranks * sizeof(u64)));
    }
    return RIO_GP_OK;
}

uint32_t rio_gp_shard_comm_ranks(rio_gp_t* h) { return h && h->sc ? h->sc->R : 0; }

// One whole fast-path step of the row-sharded solve, nothing waits on the host:
//   stream:      k_scan, k_resolve (local sums), pack X      -> event
//   side stream: ncclAllGather(X) over xGMI, k_shard_import  -> event (frees the ring slot)
// Back-to-back calls overlap solve k's exchange with solve k+1's scan.
int rio_gp_shard_solve_async(rio_gp_t* h) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    if (h->sa) return fail(h, RIO_GP_EINVAL, "row-sharded solves do not implement RIO_GP_CFG_REF_SELF_ASSIGN (single-GPU handles only)");
    if (h->sh_tick_n) return fail(h, RIO_GP_EINVAL, "rio_gp_shard_solve_async: row-sharded ticks are in flight (call rio_gp_shard_tick_wait)");
    if (h->p2p && h->p2p->d_peers) {
        // peer-to-peer, ONE stream, two launches, no collective call and no host wait: k_scan -> k_resolve_xchg.
        // Stream order is the flow control: a rank's record j+1 leaves only after it consumed everyone's record j,
        // so none of the 4 window slots (P2P::xslot_n) is overwritten while its owner still reads it.  (Running the exchange on a
        // second stream under the next scan was measured SLOWER on gfx950: two event records + two stream waits per
        // solve cost more than the 5 us they hide.)
        P2P* q = h->p2p;
        if (q->out_of_step) return fail(h, RIO_GP_EUPSTREAM, kOutOfStep);
        StepGuard step{q};
        const u64 seq = ++q->seq;
        const u32 slot = (u32)(q->xslot_n++ % kP2PSlots);
        reset_inplace(h);
        h->plan = hplan(h, h->n);
        h->sb.fx = FxRows{};
        fold_used(h);
        h->solve_used_D = false;
        const Table t = real_table(h);
        const NodeTab nt = real_nodes(h);
        launch_scan(h->plan, t, nt, h->sb, false, h->all_alive, h->stream);
        h->sh_rank = q->rank;
        h->sh_R = q->R;
        h->sh_slot = h->ring_n;
        h->sh_side = nullptr;
        // ONE launch behind the scan: every workgroup exchanges and resolves its own node group (k_resolve_xchg);
        // the verdict arrives as resolve_blocks(m) partial rows in the pinned slot
        SolveBufs xb = shard_bufs(h);
        xb.H = h->sb.H;
        xb.blkstat = h->sb.blkstat;
        launch_resolve_xchg(h->plan, nt, xb, q->d_peers, q->R, q->rank, q->xdata_off(slot, q->rank),
                            q->win + q->xdata_off(slot, 0), q->Wx, seq, q->d_err, h->sh_gprev, h->sh_gfinal,
                            slot_dev(h, h->ring_n), q->co_resident, h->stream);
        h->sh_rows = resolve_blocks(h->m);
        h->ring_n++;
        h->have_solved = false; ++h->mut_epoch;
        h->sh_state = 2;
        HIPCHK(h, hipGetLastError());
        step.done = true;
        return RIO_GP_OK;
    }
    ShardComm* sc = h->sc;
    if (!sc) return fail(h, RIO_GP_EINVAL, "rio_gp_shard_solve_async: set up rio_gp_shard_p2p_connect or rio_gp_shard_comm_init first");
    const int q = (int)(sc->k++ % kShardRing);
    if (sc->done_valid[q]) HIPCHK(h, hipStreamWaitEvent(h->stream, sc->done[q], 0));
    reset_inplace(h);
    h->plan = hplan(h, h->n);
    h->sb.fx = FxRows{};
    fold_used(h);
    h->solve_used_D = false;
    const Table t = real_table(h);
    const NodeTab nt = real_nodes(h);
    launch_scan(h->plan, t, nt, h->sb, false, h->all_alive, h->stream);
    const SolveBufs lb = local_bufs(h);
    launch_resolve(h->plan, nt, lb, nullptr, h->stream);
    launch_shard_pack1(h->plan, lb, sc->X[q], h->stream);
    HIPCHK(h, hipEventRecord(sc->ready[q], h->stream));
    HIPCHK(h, hipStreamWaitEvent(sc->side, sc->ready[q], 0));
    const int rc = sc->api.AllGather(sc->X[q], sc->XG[q], shard_words1(h->m), kNcclUint64, sc->comm, sc->side);
    if (rc != 0) return fail(h, RIO_GP_EUPSTREAM, std::string("ncclAllGather: ") + (sc->api.GetErrorString ? sc->api.GetErrorString(rc) : "failed"));
    h->sh_rank = sc->rank;
    h->sh_R = sc->R;
    h->sh_slot = h->ring_n;
    h->sh_rows = 1;
    h->sh_side = sc->side;
    launch_shard_import(h->plan, nt, shard_bufs(h), sc->XG[q], sc->rank, sc->R, h->sh_gprev, h->sh_gfinal, h->sh_verdict,
                        slot_dev(h, h->ring_n), sc->side);
    HIPCHK(h, hipEventRecord(sc->done[q], sc->side));
    sc->done_valid[q] = true;
    h->ring_n++;
    h->have_solved = false; ++h->mut_epoch;
    h->sh_state = 2;
    return RIO_GP_OK;
}

// ---- asynchronous committed tick of the row-sharded table (peer-to-peer windows) ----
// Record of tick k in pinned memory: k_resolve_xchg's partial verdict rows in the tick ring's slot k (d_slots), and in slot
// 1 + k of the fix-up counter rows (h_fx; the row-sharded solve does not use them otherwise): rows 0-1 = this rank's counters
// (k_shard_tick_stats, 16 words), row 2 + e = {rows, load} pending on all ranks after exchange e (k_shard_import_delta).
static void shard_exchange_y(rio_gp* h, const SolveBufs& b, const u64* base, int wsp_sel, u64* verdict_host) {
    P2P* q = h->p2p;
    const u64 seq = ++q->seq;
    const u32 slot = (u32)(q->yslot_n++ % kP2PSlots);
    launch_shard_export_put(h->plan, b, base, wsp_sel, q->d_peers, q->R, q->data_off(slot, q->rank), q->flag_off(slot, q->rank), seq,
                            h->stream);
    launch_shard_wait_import(h->plan, b, q->win + q->data_off(slot, 0), q->W, q->win + q->flag_off(slot, 0), seq, q->d_err,
                             h->sh_rank, h->sh_R, h->sh_gprev, h->sh_gfinal, h->sh_verdict, verdict_host, h->stream);
}

int rio_gp_shard_tick_async(rio_gp_t* h) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    if (h->sa) return fail(h, RIO_GP_EINVAL, "row-sharded solves do not implement RIO_GP_CFG_REF_SELF_ASSIGN (single-GPU handles only)");
    P2P* q = h->p2p;
    if (!q || !q->d_peers) return fail(h, RIO_GP_EINVAL, "rio_gp_shard_tick_async: peer-to-peer windows only (rio_gp_shard_p2p_connect first)");
    if (h->ring_n || h->tick_n) return fail(h, RIO_GP_EINVAL, "rio_gp_shard_tick_async: other asynchronous solves are in flight");
    if (h->sh_tick_n == (u32)kRing) return fail(h, RIO_GP_EINVAL, "rio_gp_shard_tick_async: 64 ticks in flight (call rio_gp_shard_tick_wait)");
    if (q->out_of_step) return fail(h, RIO_GP_EUPSTREAM, kOutOfStep);
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    const u32 k = h->sh_tick_n;
    // (1) the fast path: k_scan -> k_resolve_xchg, verdict rows into this tick's slot of the tick ring
    StepGuard step{q};  // (every sequence number this tick takes — its own and its exchanges' — is taken before any check below)
    const u64 seq = ++q->seq;
    const u32 slot = (u32)(q->xslot_n++ % kP2PSlots);
    reset_inplace(h);
    h->plan = hplan(h, h->n);
    h->sb.fx = FxRows{};
    fold_used(h);
    h->solve_used_D = false;
    const Table t = real_table(h);
    const NodeTab nt = real_nodes(h);
    launch_scan(h->plan, t, nt, h->sb, false, h->all_alive, h->stream);
    h->sh_rank = q->rank;
    h->sh_R = q->R;
    h->sh_side = nullptr;
    SolveBufs b = shard_bufs(h);
    b.H = h->sb.H;
    b.blkstat = h->sb.blkstat;
    u64* rows = h->d_slots + (size_t)(kTickSlot0 + k) * h->slot_rows * 8;
    u64* rec = h->d_fx + (size_t)(1 + k) * kMaxBlocks * 8;
    launch_resolve_xchg(h->plan, nt, b, q->d_peers, q->R, q->rank, q->xdata_off(slot, q->rank), q->win + q->xdata_off(slot, 0),
                        q->Wx, seq, q->d_err, h->sh_gprev, h->sh_gfinal, rows, q->co_resident, h->stream);
    // (2) exact cut on this rank: k_cutblk + k_cut_find guard themselves on the cut flag, the re-marking pass on the number
    //     of nodes k_resolve_xchg found to need it here
    b.run_if = &h->dstats->local_fixup;
    launch_cut_find(h->plan, t, nt, b, false, h->stream, false);
    launch_fill(h->plan, t, nt, b, false, true, false, 0, false, h->stream);
    // (3) what this rank admitted, everyone's, the global `used`, this rank's spill base
    shard_exchange_y(h, b, h->sb.used_kept, 0, rec + 16);
    // (4) one water-fill round per spill round (a no-op on the device when nothing is pending anywhere), each with its exchange
    for (u32 r = 0; r < h->rounds; ++r) {
        launch_fill(h->plan, t, nt, b, false, false, true, (int)r, r + 1 == h->rounds, h->stream);
        shard_exchange_y(h, b, h->sh_gprev, (int)((r & 1) ^ 1), rec + 8 * (size_t)(3 + r));
    }
    // (5) this rank's counters, (6) publication
    h->sh_tick_mark[k] = (1ull << 41) | ++h->wait_seq;
    launch_shard_tick_stats(h->plan, b, rec, h->sh_tick_mark[k], h->stream);
    HIPCHK(h, hipGetLastError());
    h->have_solved = true;
    if ((rc = commit_enqueue(h))) return rc;
    h->sh_tick_n = k + 1;
    h->sh_state = 0;
    step.done = true;
    return RIO_GP_OK;
}

int rio_gp_shard_tick_wait(rio_gp_t* h, rio_gp_shard_tick_info* out, uint32_t cap, uint32_t* n_out) {
    if (!h || !n_out || (cap && !out)) return RIO_GP_EINVAL;
    Locked g(h);
    *n_out = 0;
    if (!h->sh_tick_n) return RIO_GP_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    const u32 n = h->sh_tick_n;
    h->sh_tick_n = 0;
    if (h->p2p && h->p2p->d_peers) { int rc = p2p_check(h); if (rc) return rc; }
    const u32 nb = resolve_blocks(h->m);
    const u32 take = n < cap ? n : cap;
    for (u32 k = n - take; k < n; ++k) {
        rio_gp_shard_tick_info& o = out[k - (n - take)];
        const u64* rows = h->h_slots + (size_t)(kTickSlot0 + k) * h->slot_rows * 8;
        const u64* rec = h->h_fx + (size_t)(1 + k) * kMaxBlocks * 8;
        if (rec[15] != h->sh_tick_mark[k]) return fail(h, RIO_GP_EUPSTREAM, "rio_gp_shard_tick_wait: a tick left no record");
        u64 x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (u32 r = 0; r < nb; ++r)
            for (int c = 0; c < 8; ++c) x[c] += rows[(size_t)r * 8 + c];
        DevStats v;
        memset(&v, 0, sizeof v);
        v.load_kept = rec[0]; v.load_claim_tot = rec[1];
        v.kept = rec[3]; v.evicted = rec[4]; v.claimants = rec[5]; v.spillcand = rec[6];
        const bool slow = x[0] > 0 || x[1] > 0;
        if (slow) {
            v.rejected = rec[8]; v.load_rejected = rec[9];
            v.spilled = rec[10]; v.load_spilled = rec[11];
            v.unplaced = rec[12]; v.load_unplaced = rec[13];
        }
        fill_stats(v, h->n, &o.local);
        o.cut_nodes = x[0];
        o.spill_rows = x[1];
        o.slow_path = slow ? 1u : 0u;
        o.rounds_run = 0;
        for (u32 r = 0; slow && r < h->rounds; ++r) o.rounds_run += rec[8 * (size_t)(2 + r)] > 0;  // rows pending BEFORE round r
    }
    *n_out = n;
    return RIO_GP_OK;
}

// all-gather of `words` u64 per rank on the handle's stream (the Y records of the fix-up path, counters)
int rio_gp_shard_exchange(rio_gp_t* h, const uint64_t* d_in, uint64_t* d_out, uint64_t words) {
    if (!h || !d_in || !d_out) return RIO_GP_EINVAL;
    Locked g(h);
    if (h->p2p && h->p2p->d_peers) {
        P2P* q = h->p2p;
        if (words > q->W) return fail(h, RIO_GP_EINVAL, "rio_gp_shard_exchange: record larger than the window row");
        if (q->out_of_step) return fail(h, RIO_GP_EUPSTREAM, kOutOfStep);
        StepGuard step{q};  // (a failure behind the sequence number / slot taken here leaves the ranks out of step: marked)
        const u64 seq = ++q->seq;
        const u32 slot = (u32)(q->yslot_n++ % kP2PSlots);
        launch_p2p_put(reinterpret_cast<const u64*>(d_in), (u32)words, q->d_peers, q->R, q->data_off(slot, q->rank),
                       q->flag_off(slot, q->rank), seq, h->stream);
        launch_p2p_wait_copy(q->win + q->data_off(slot, 0), q->W, q->R, (u32)words, q->win + q->flag_off(slot, 0), seq,
                             q->d_err, reinterpret_cast<u64*>(d_out), h->stream);
        const int rc = p2p_check(h);
        step.done = rc == RIO_GP_OK;
        return rc;
    }
    if (!h->sc) return fail(h, RIO_GP_EINVAL, "rio_gp_shard_exchange: call rio_gp_shard_comm_init first");
    const int rc = h->sc->api.AllGather(d_in, d_out, (size_t)words, kNcclUint64, h->sc->comm, h->stream);
    if (rc != 0) return fail(h, RIO_GP_EUPSTREAM, "ncclAllGather failed");
    return RIO_GP_OK;
}

#ifdef RIO_GP_LAB
// ---- lab build only (librio_gp_lab.so, include/rio_gpu_placement_debug.h): policy knobs for the parity tests and A/B
//      runs, the streaming / host round-trip probes.  None of it is in the product library.
void rio_gp_debug_set_scan_nt(int mode) { set_scan_nt(mode); }
void rio_gp_debug_set_part_shift(int shift) { set_part_shift(shift); }
uint64_t rio_gp_debug_wave_row_lo(uint64_t n_objects, uint32_t n_nodes, uint32_t wave, uint32_t* n_waves) {
    return plan_wave_row_lo(n_objects, n_nodes, wave, n_waves);
}

int rio_gp_debug_set_compact(rio_gp_t* h, int mode) {
    if (!h || mode < 0 || mode >= 8192 || (mode & 15) > 2 || ((mode >> 5) & 3) == 3) return RIO_GP_EINVAL;  // nothing is changed
    Locked g(h);
    h->part_mode = (mode & 16) ? 2 : 0;  // bit 4: big update / remove batches through the plain kernels (A/B runs, parity tests)
    h->cutpack_mode = (mode >> 5) & 3;   // bits 5-6: packing at the cut pass of whole-table solves, 0 auto | 1 always | 2 never
    h->compact_mode = mode & 15;
    h->inc_mode = (mode >> 7) & 3;       // bits 7-8: in-place scan of committed ticks, 0 auto | 1 whatever the table's size | 2 never
    if (h->inc_mode == 3) h->inc_mode = 0;
    h->cutapply_mode = (mode >> 9) & 3;  // bits 9-10: 0 = k_cut_apply when the solve packs at the cut pass | 1 = always | 2 = never
    if (h->cutapply_mode == 3) h->cutapply_mode = 0;
    h->overlap_mode = (mode & 2048) ? 2 : 0;  // bit 11: quiet ticks do not overlap (k_resolve on the main stream, as before round 6)
    h->chain_mode = (mode & 4096) ? 2 : 0;    // bit 12: overlapped quiet ticks are not chained (every scan on the main stream)
    return RIO_GP_OK;
}

uint64_t rio_gp_debug_chained_scans(rio_gp_t* h) {
    if (!h) return 0;
    Locked g(h);
    return h->chain_total;
}

int rio_gp_debug_set_speculate(rio_gp_t* h, int speculate) {
    if (!h || speculate < 0 || speculate > 2) return RIO_GP_EINVAL;
    Locked g(h);
    h->spec_mode = speculate;
    return RIO_GP_OK;
}

int rio_gp_debug_ktrace(rio_gp_t* h, int enable, int table, uint64_t* out2048) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (out2048 && ktrace_read(table, reinterpret_cast<u64*>(out2048)) != 0) return fail(h, RIO_GP_EUPSTREAM, "ktrace read failed");
    ktrace_enable(enable);
    return RIO_GP_OK;
}

int rio_gp_debug_stream_probe(rio_gp_t* h, int mode, int reps, float* ms) {
    if (!h || !ms || reps < 1) return RIO_GP_EINVAL;
    Locked g(h);
    HIPCHK(h, hipSetDevice(h->device));
    if (mode >= 20 && mode <= 23) {  // host round-trip probes: microseconds per call / 1000
        *ms = sync_probe(mode, reps, h->stream);
        if (*ms < 0) return fail(h, RIO_GP_EUPSTREAM, "sync probe failed");
        return RIO_GP_OK;
    }
    // same columns the solve streams: cur/load/aff in, the ping-pong column out (an uncommitted solve is lost)
    *ms = stream_probe(mode, h->assign[h->cur], h->load, h->aff, h->assign[h->cur ^ 1], h->n, reps, h->stream, h->ev0,
                       h->ev1);
    h->have_solved = false; ++h->mut_epoch;
    if (*ms < 0) return fail(h, RIO_GP_EUPSTREAM, "stream probe failed");
    return RIO_GP_OK;
}
#endif  // RIO_GP_LAB

int rio_gp_timer_begin(rio_gp_t* h) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    h->timer_stopped = false;
    return RIO_GP_OK;
}

int rio_gp_timer_stop(rio_gp_t* h) {
    if (!h) return RIO_GP_EINVAL;
    Locked g(h);
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    h->timer_stopped = true;
    return RIO_GP_OK;
}

int rio_gp_timer_end(rio_gp_t* h, float* ms) {
    if (!h || !ms) return RIO_GP_EINVAL;
    Locked g(h);
    if (!h->timer_stopped) HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    h->timer_stopped = false;
    HIPCHK(h, hipEventSynchronize(h->ev1));
    HIPCHK(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return RIO_GP_OK;
}

}  // extern "C"
