// placement_kernels.h — launch interface between the C ABI (rio_gp_capi.hip) and the gfx950
// kernels (placement_kernels.hip).  Internal; the public boundary is include/rio_gpu_placement.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace riogp {

typedef uint32_t u32;
typedef unsigned long long u64;

constexpr u32 kNone = 0xFFFFFFFFu;       // unplaced (RIO_GP_NONE)
constexpr u32 kSpillMark = 0xFFFFFFFEu;  // pending, waiting for the water-fill (never survives a solve)
constexpr u32 kSkipMark = 0xFFFFFFFDu;   // virtual-table row that is a duplicate request (place_pending)
constexpr u32 kAffInactive = 0xFFFFFFFEu;  // AFFINITY of a row that is not an object (RIO_GP_AFF_INACTIVE): never placed
constexpr u32 kFlagReplaced = 0x10u;  // RIO_GP_FLAG_REPLACED (rio_gpu_placement.h)
constexpr u32 kNoCut = 0xFFFFFFFFu;

constexpr int kWaves = 16;           // waves per workgroup of the streaming kernels
constexpr int kBlock = kWaves * 64;  // 1024 threads: ONE workgroup per CU owns one LDS histogram
constexpr int kTile = 256;           // objects per wave-iteration: 64 lanes x dwordx4
constexpr u32 kMaxBlocks = 256;      // = CUs; rows of the per-block histogram table
constexpr u32 kMaxSubs = 256;
constexpr int kMidBatch = 16384;     // lookups of up to this many entries go through mapped pinned memory and a completion word
constexpr int kReqBatch = 131072;    // place_pending batches from host buffers of up to this many requests: mapped pinned memory the kernels
                                     // read and write over PCIe themselves; bigger ones: the caller's arrays registered for the call, DMA copies
                                     // (measured per size, profiles/round6_pp_sizes.txt: 65 536 requests 125 us through the pinned rows against
                                     //  137-182 through registered arrays; 262 143: 398 against 249-373)
constexpr int kSmallBatch = 256;     // place_pending / lookup micro-batches served by one workgroup and one launch        // sub-chunks per block for the exact-cut refinement

// Work decomposition of a table of n rows.  Index order is the only order that matters:
// the table is `tiles` tiles of kTile rows; wave `gw` (of nw = G*kWaves) owns the contiguous tiles
// [gw*tiles/nw, (gw+1)*tiles/nw) (balanced to +-1 tile), block b owns kWaves consecutive wave
// ranges, sub-chunk t of block b is [block_row_lo(b) + t*sub, +sub).
struct Plan {
    u64 n;       // rows
    u64 tiles;   // ceil(n / kTile)
    u64 tq;      // tiles / nw   } the balanced split without a division on the device
    u64 div_magic;  // ceil(2^38 / nw)
    u32 tr;      // tiles % nw
    u32 nw;      // wave ranges = G * kWaves
    u32 G;       // blocks (<= kMaxBlocks)
    u32 sub;     // rows per sub-chunk (multiple of kTile)
    u32 subs;    // sub-chunks per block (<= kMaxSubs)
    u32 m;       // nodes
    u32 mwords;  // ceil(m/32)
    u32 trace;   // lab build: phase traces of the fix-up kernels (0 in the product)
    u32 sa;      // RIO_GP_CFG_REF_SELF_ASSIGN: a pending row claims its affinity node (a request: its requester) whether or not
                 // membership marks that node active, against the node's whole capacity (service.rs:244-252 self-assigns
                 // unconditionally); the water-fill still places on live nodes only.  0: a claim needs a live node
    u64 mark;    // what k_resolve stores in column 7 of its partial rows ("row present"): 1, or the sequence number the
                 // host spins on instead of waiting for the stream
    // packed fix-up (see PackOut): when set, wave gw's rows are only the first wcnt[gw] positions of its range
    const u32* wcnt;
    // k_scan only (launch_scan sets it from NodeTab::alive_src): the liveness bitmap the kernel was given is a fresh one in
    // mapped host memory; workgroup 0 copies it here, the device array the kernels behind the scan read
    u32* alive_dst;
};
Plan make_plan(u64 n, u32 m, u32 max_blocks);
u64 plan_wave_row_lo(u64 n, u32 m, u32 gw, u32* nw_out);  // first row of wave range gw, as the device computes it (debug / tests)

// Accumulators of one solve, device resident (all 64-bit so they can be atomically added).
struct DevStats {
    u64 kept, evicted, claimants, spillcand;            // k_scan   (rows)
    u64 load_kept, load_claim_tot, n_cut;               // k_resolve
    u64 rejected, load_rejected;                        // k_apply_cut
    u64 spilled, load_spilled, unplaced, load_unplaced; // k_spill_apply
    u64 rounds_run;                                     // k_spill_rank
    u64 evicted_clean;                                  // k_clean
    u64 err;                                            // invalid entries seen by batch kernels
    u64 local_fixup;                                    // row-sharded solve: nodes whose claimants of THIS rank need re-marking (k_resolve_xchg)
    u64 global_slow;                                    // row-sharded solve: != 0 iff the solve needs the fix-up on ANY rank (the same
                                                        // value on every rank: it is computed from the all-gathered sums)
};

// Per-workgroup rows of the fix-up counters (a whole-table solve whose counters the host wants): workgroup b of the
// streaming grid owns row b of `dev` across the kernels of one solve — k_scan zeroes it, the cut pass and every water-fill
// round add to it with plain stores, and each round also stores the row into `host` (pinned, mapped): the host folds the
// rows when it reads the verdict.  No contended atomics on eight global words (a 255-to-1 fan-in is ~3 us at the tail of
// three kernels per tick) and no copy kernel for the totals.  Both nullptr: the kernels add into DevStats atomically
// (row-sharded solve, place_pending).
// row = { rejected rows, rejected load, spilled rows, spilled load, unplaced rows, unplaced load, rounds run (row 0), - }
struct FxRows {
    u64* dev = nullptr;   // [kMaxBlocks][8]
    u64* host = nullptr;  // [kMaxBlocks][8], device address of pinned host memory
    u64 seq = 0;          // != 0: the LAST water-fill round stores it in word 7 of every row (also when the round is a
                          // no-op): the host spins on the pinned rows instead of waiting for the stream
};

// Scratch of one solve over one table (real table or the virtual table of place_pending).
struct SolveBufs {
    u64* H;          // [ceil(m/8)][G][16] per-block load histograms, node-group major: kept-by-cur x8 | claim-by-aff x8
                     //   (one 128-byte line per (group of 8 nodes, block): k_resolve reads G contiguous lines per workgroup)
    u64* partial;    // [ceil(m/8)][8] k_resolve per-workgroup partial counters (device copy)
    u64* blkstat;    // [G][4]   kept, evicted, claimants rows per block
    u64* wsp_sum[2]; // [G*kWaves] spill-candidate load per wave range (ping-pong over rounds)
    u32* wsp_cnt[2]; // [G*kWaves]
    u64* bsp_sum[2]; // [G] the same per workgroup (what k_spill_apply's prologue folds: G words instead of G*kWaves)
    u32* bsp_cnt[2]; // [G]
    u64* used_kept;  // [m] load of kept rows (+ used_base for the virtual table)
    u64* used_cur;   // [m] used_kept + admitted claims + admitted spills
    u64* claim_tot;  // [m]
    u32* cutblk;     // [m] block containing the cut or kNoCut
    u64* budget;     // [m] free capacity left at the start of the cut block
    u64* admpre;     // [m] claim load admitted before the cut block
    u32* cutidx;     // [m] row index of the first rejected claimant or kNoCut
    // row-sharded solve only (nullptr otherwise): nodes whose claim prefix overflowed on a lower rank, and the
    // spill load pending on lower ranks
    u32* forced_bits;        // [mwords]
    u64* rank_base;          // [1]
    const u64* pending_global;  // [1] rows still pending on ALL ranks (k_shard_import_delta), nullptr = local count
    const u64* run_if = nullptr;  // the re-marking pass (k_fill<APPLY> without FILL) returns at once when *run_if == 0
                                  // (asynchronous row-sharded tick: DevStats::local_fixup; nullptr = always run)
    // The claim load the cuts reject, for the ordered spill prefix of k_fill's round 0 (no pass over rows):
    u64* RP;         // [node groups][G] by node group, in the blocks BEFORE block b: k_scan zeroes, a k_resolve workgroup that owns
                     //   cut nodes stores its row (plain stores, whole lines); nullptr: not maintained
    u64* R;          // [G] correction of the cut blocks themselves (what a cut block admits of its nodes), as a negative
                     //   number: k_scan zeroes, k_cut_find adds (k_resolve<SEARCH> folds it into RP instead)
    u64* D;          // [kFillRounds][m] load admitted per water-fill round, kept apart from used_cur so that no round reads
                     //   and writes the same vector: k_resolve zeroes it, round r of k_fill orders the nodes by
                     //   used_cur + D[0..r) and adds into D[r]; the committed `used` is used_cur + sum D (launch_used_fold).
                     //   nullptr (row-sharded solve): the rounds read used_snap and add straight into used_cur
    const u64* used_snap;  // row-sharded solve: the global `used` vector as the last exchange left it (nobody writes it during a round)
    u64* Tg = nullptr;     // [m][16] k_cut_apply's per-wave claim sums of the undecided rows: k_resolve zeroes the rows of the nodes
                           //   it finds a cut for (nullptr: the solve's fix-up does not use k_cut_apply)
    DevStats* stats;
    FxRows fx;
};
constexpr u32 kFillRounds = 8;  // rows of SolveBufs::D = the largest number of water-fill rounds per solve

// Packed pending rows (k_scan<COMPACT>): wave gw copies its PENDING rows, in index order, to the front of its own
// row range in these scratch columns and records how many (wcnt[gw]).  The fix-up kernels then run unchanged over
// {cur = an all-NONE column, load, aff, next} with Plan::wcnt set, so their passes cost O(pending rows) instead of
// O(rows); k_pk_scatter writes the decisions back through `idx`.  Index order is preserved: (wave, packed position).
struct PackOut {
    u32* idx;
    u32* load;
    u32* aff;
    u32* next;
    u32* wcnt;  // [G*kWaves]
};

// table columns; for the real table cur/next are the ping-pong assignment columns
struct Table {
    const u32* cur;
    const u32* load;
    const u32* aff;
    u32* next;
    // packed fix-up only (else nullptr): the water-fill also writes each decision to real_next[pk_idx[position]],
    // which makes the separate k_pk_scatter launch unnecessary when at least one spill round runs
    const u32* pk_idx = nullptr;
    u32* real_next = nullptr;
    // the packing pass has already stored NONE into the real row of every packed row (k_cut_apply_rank<PACK>): the last
    // round then writes only the rows it places, not one scattered NONE per row it could not place
    bool none_prewritten = false;
    // virtual table of a window-sorted request batch (k_pp_win_gather): its rows arrive as 8-byte answer records {node | flag |
    // later, load} — the scan reads them and leaves cur / load as columns for the kernels behind it — and the alive requesters' first touches
    // are already in the real column (the scan stores only those whose requester is not alive: RIO_GP_CFG_REF_SELF_ASSIGN)
    const uint2* vrec = nullptr;
    bool prewritten = false;
    // plain virtual table (general request path): when *skip_if != 0 the batch held an invalid entry and the table was not
    // built — the scan solves an empty one
    const u32* skip_if = nullptr;
};

struct NodeTab {
    const u64* cap;         // [m]
    const u32* alive_bits;  // [mwords]
    const u64* used_base;   // [m] or nullptr (virtual table: the committed `used`)
    // A liveness push that has not reached alive_bits yet (else nullptr): the bitmap in mapped pinned host memory.  The
    // scan of the next solve reads it from there and brings alive_bits up to date on its way — the push costs no launch.
    const u32* alive_src = nullptr;
};

// Chained quiet ticks (rio_gp_tick_async over a table nothing has changed in): the scans of consecutive ticks alternate between
// two streams and hand their rows over wave range by wave range (per workgroup in the other form) — workgroup b of tick k + 1 reads what workgroup b of tick k
// wrote (and writes what it read), so it waits for THAT workgroup's flag instead of for the whole launch; the ramp-down of
// one scan and the ramp-up of the next overlap.  flags[b] = `set` of the last chained scan whose workgroup b is through.
struct ScanChain {
    u32* flags;   // [kMaxBlocks] per workgroup, then [kMaxBlocks * kWaves] per wave range; device memory
    u32* err;     // one word of mapped host memory: raised when a wait gave up (the tables are then stale)
    u32 wait;     // sequence number of the scan to wait for (0: none — the stream orders this scan behind what it depends on)
    u32 set;      // this scan's sequence number
    u32 per_wave; // the hand-over is per wave range (a flag per wave, no barrier on its path) instead of per workgroup
};

bool scan_chain_fits(u32 m);  // two workgroups of the chained scan per CU (what makes the in-kernel wait deadlock-free)

// --- solve pipeline ---
void launch_scan(const Plan& p, const Table& t, const NodeTab& nt, const SolveBufs& b, bool virt, bool all_alive,
                 hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, const PackOut* pack = nullptr,
                 const ScanChain* chain = nullptr);
// host_partial: pinned host rows [resolve_blocks(m)][8] = load_kept, load_claim_tot, n_cut, kept, evicted,
// claimants, spillcand, present — the caller adds the rows up (no atomics / copy kernel on the stream).
// search (with p.wcnt set): the packed pending rows — k_resolve also finds the exact cut rows (no k_cut_find launch).
// fold_into: the committed `used` vector still waiting for the D rows of the previous committed solve (folded in before
// b.D is zeroed), fold_rounds = that solve's rounds.
// kept_from: the scan built no kept histogram (launch_inc_scan) — the kept load of a node is the committed vector's entry
// where the node is alive (read after the fold when it is fold_into itself).
void launch_resolve(const Plan& p, const NodeTab& nt, const SolveBufs& b, u64* host_partial, hipStream_t s,
                    hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, const PackOut* search = nullptr,
                    u64* fold_into = nullptr, u32 fold_rounds = 0, const u64* kept_from = nullptr);
// The scan of a COMMITTED tick over a mostly-placed table (k_inc_scan): cur / load / aff are streamed like k_scan does, the
// assignment column is updated in place and only where a row's value changes, no histogram is built; the pending rows ->
// pack (per wave range of p).  launch_rebal deals them out evenly over rebal_plan(p) into `dst` and builds the fix-up's
// per-block histograms and spill totals there.
bool inc_scan_fits(u32 m);
void launch_inc_scan(const Plan& p, u32* assign, const u32* load, const u32* aff, const NodeTab& nt, const SolveBufs& b,
                     const PackOut& pack, hipStream_t s);
Plan rebal_plan(const Plan& p);
u64 rebal_rows(u64 n);  // rows of the balanced columns for a table of n rows (+ the usual padding)
void launch_rebal(const Plan& p, const Plan& pv, const PackOut& src, const NodeTab& nt, const PackOut& dst, const SolveBufs& b,
                  hipStream_t s);
unsigned resolve_blocks(u32 m);
void set_scan_nt(int mode);  // 0 by table size | 1 always | 2 never: non-temporal column streams in k_scan
// The exact cut search as a launch of its own (k_cut_find; guards itself on the device).  have_cutblk: launch_resolve of
// the same solve (same bufs) has already written cutblk / budget / admpre — else k_cutblk runs first (row-sharded path).
void launch_cut_find(const Plan& p, const Table& t, const NodeTab& nt, const SolveBufs& b, bool virt, hipStream_t s,
                     bool have_cutblk);
// One launch of k_fill: apply = re-mark the rejected claimants (+ pack: copy the rows that go on to the water-fill into the
// pack columns, per-wave counts in pack->wcnt; real table only), fill = one water-fill round.  apply && fill is round 0
// of a solve, fill alone a later round, apply alone the row-sharded solve's cut step.  The workgroups order the nodes
// themselves (by capacity class, in LDS): there is no ranking launch.
void launch_fill(const Plan& p, const Table& t, const NodeTab& nt, const SolveBufs& b, bool virt, bool apply, bool fill,
                 int round, bool last, hipStream_t s, const PackOut* pack = nullptr);
bool fill_can_pack(u32 m);
// k_cut_apply (real table of one GPU, whole-table solves): the exact cut search and the re-marking pass as ONE pass over the
// rows — replaces launch_cut_find + the apply half of round 0; the rounds behind it are launch_fill(apply = false) from round 0
// on, over pk (pack: Plan::wcnt = pk.wcnt, Table::none_prewritten) or over the table (pk is scratch for the undecided rows)
bool cut_apply_fits(u32 m);
void launch_cut_apply(const Plan& p, const Table& t, const NodeTab& nt, const SolveBufs& b, const PackOut& pk, const PackOut& ul,
                      u64* Tg, bool pack, bool all_alive, hipStream_t s);
// used[j] += D[0][j] + ... + D[rounds-1][j]
void launch_used_fold(u64* used, const u64* D, u32 m, u32 rounds, hipStream_t s);
#ifdef RIO_GP_LAB
int ktrace_enable(int on);
int ktrace_read(int table, u64* out /*[kMaxBlocks*8]*/);  // 0 k_resolve<search> | 1 k_fill round 0 | 2 k_fill later rounds
float sync_probe(int mode, int reps, hipStream_t s);  // host round-trip probes (stream_probe.hip)
float stream_probe(int mode, const u32* a, const u32* b, const u32* c, u32* o, u64 n, int reps, hipStream_t s,
                   hipEvent_t e0, hipEvent_t e1);
#endif

// --- CRUD over the assignment column ---
// the requests of a call of at most 4 entries, passed in the kernel arguments (a = object indices, b = nodes / requesters)
struct SmallInline { u32 a[4]; u32 b[4]; };
void launch_lookup_small(const u32* assign, u64 n_obj, const u32* idx, u32 n, u32* out, DevStats* st, hipStream_t s,
                         u32* done, u32 seq, const SmallInline* inl = nullptr);
// done / seq (lookup, update_small, remove, pp_small): when the call is ONE workgroup, its last act is to store seq into
// *done (mapped pinned memory) — the host spins on the word instead of waiting for the stream; nullptr = no word
void launch_lookup(const u32* assign, u64 n_obj, const u32* idx, u64 n, u32* out, DevStats* st, hipStream_t s,
                   u32* done = nullptr, u32 seq = 0, unsigned int* ticket = nullptr);  // ticket (device word, 0 between calls):
                   // several workgroups may share the completion word — the last one to finish stores it
// aff_life (every CRUD launcher below): the affinity column when the handle tracks the row lifecycle (rows that are
// written become objects, rows that are removed / deleted / dropped by clean_server stop being objects), else nullptr
// used / load != nullptr (medium batches): the per-node load vector follows the writes (a few global atomics)
void launch_update(u32* assign, u64 n_obj, u32 m, const u32* idx, const u32* node, u64 n, u32* pos_scratch,
                   DevStats* st, hipStream_t s, u32* aff_life = nullptr, unsigned int* ticket = nullptr, u32* done = nullptr,
                   u32 seq = 0, u64* used = nullptr, const u32* load = nullptr);
// n <= kSmallBatch validated entries (may be mapped host memory): last writer wins inside the batch, one launch
void launch_update_small(u32* assign, const u32* idx, const u32* node, u32 n, hipStream_t s, u32* aff_life = nullptr,
                         u32* done = nullptr, u32 seq = 0, const SmallInline* inl = nullptr, u64* used = nullptr,
                         const u32* load = nullptr, u32 m = 0);
// n <= kSmallBatch validated entries (may be mapped host memory / kernel arguments): one small workgroup
void launch_remove_small(u32* assign, u32 m, const u32* load, const u32* idx, u32 n, u64* used_or_null, hipStream_t s,
                         u32* aff_life, u32* done, u32 seq, const SmallInline* inl);
void launch_remove(u32* assign, u64 n_obj, u32 m, const u32* load, const u32* idx, u64 n, u64* used_or_null,
                   DevStats* st, hipStream_t s, u32* aff_life = nullptr, u32* done = nullptr, u32 seq = 0,
                   const SmallInline* inl = nullptr, unsigned int* ticket = nullptr);
// big random batches, partitioned by row window first (k_part_bin ...): part_applicable says whether a batch qualifies,
// scratch = part_scratch_words(n_obj, n) u32 words of device memory
bool part_applicable(u64 n_obj, u64 n, const void* idx, const void* node_or_null);
size_t part_scratch_words(u64 n_obj, u64 n);
// end of a synchronous call: the error counter (as 0 / 1) into a mapped host word, then the completion word (the host spins on it)
void launch_finish_err(const DevStats* st, u32* host_err, u32* done, u32 seq, hipStream_t s);
void set_part_shift(int shift);  // rows per window = 1 << shift, 12..14 (A/B runs)
void launch_update_part(u32* assign, u64 n_obj, u32 m, const u32* idx, const u32* node, u64 n, u32* scratch, DevStats* st,
                        hipStream_t s, u32* aff_life = nullptr);
void launch_remove_part(u32* assign, u64 n_obj, u32 m, const u32* load, const u32* idx, u64 n, u32* scratch,
                        u64* used_or_null, DevStats* st, hipStream_t s, u32* aff_life = nullptr);
// counter == nullptr: accumulate into st->evicted_clean.  ticket/host_out: self-resetting counter + total written to
// mapped host memory by the last workgroup (dead_bits may itself be mapped host memory).
void launch_clean(u32* assign, u64 n_obj, u32 m, const u32* dead_bits, u64* used_or_null, DevStats* st, hipStream_t s,
                  u64* counter = nullptr, unsigned int* ticket = nullptr, u64* host_out = nullptr, u32* aff_life = nullptr,
                  u32 seq = 0 /* != 0: *host_out = total | seq << 40, the word the host spins on */,
                  const u32* skip_if = nullptr /* asynchronous form: do nothing when *skip_if != 0 */);
void launch_recompute_used(const u32* assign, const u32* load, u64 n_obj, u32 m, u64* used, hipStream_t s);
void launch_fill_u32(u32* p, u64 n, u32 v, hipStream_t s);
void launch_set_attrs(u32* load, u32* aff, u64 n_obj, const u32* idx, const u32* nload, const u32* naff, u64 n,
                      DevStats* st, hipStream_t s);
void launch_count_placed(const u32* assign, u64 n_obj, DevStats* st, hipStream_t s);
void launch_pack_alive(const uint8_t* alive_bytes, u32 m, u32* alive_bits, hipStream_t s);
struct WordPack { u32 w[256]; };  // 8 192 bits = RIO_GP_MAX_NODES, passed by value as a kernel argument
void launch_store_words(const WordPack& pack, u32 nwords, u32* dst, hipStream_t s);

// --- place_pending, batches of up to kOneBatch requests: ONE workgroup, one launch; idx/req/out_* may be mapped host
//     memory; *status = 1 means "needs the general path", nothing was changed ---
constexpr int kOneBatch = 4096;
// stage + ticket (device memory: pp_stage_bytes() of scratch, one zeroed word): batches of more than kPpStagedFrom requests
// from HOST buffers (host_io; device-resident batches: more than 1 024) go through three launches — requests and their rows
// gathered by many workgroups, the decision in one, results and table stores by many again (k_pp_stage / k_pp_decide /
// k_pp_apply) — instead of one workgroup doing all of it
constexpr int kPpStagedFrom = 256;
inline size_t pp_stage_bytes() { return (size_t)kOneBatch * (16 + 16 + 8) + 64; }
// the update / remove / lookup parts of a mixed micro-batch (<= kSmallBatch validated entries each; *_inl: n <= 4 entries in
// the kernel arguments): run by the one workgroup of k_pp_one<kSmallBatch, 1> in front of the requests, or alone
struct CrudSmallArgs {
    u32 nu = 0, nr = 0, nl = 0;
    const u32 *u_idx = nullptr, *u_node = nullptr, *r_idx = nullptr, *l_idx = nullptr;
    u32* l_out = nullptr;
    const SmallInline *u_inl = nullptr, *r_inl = nullptr, *l_inl = nullptr;
    u64 n_obj = 0;
    DevStats* st = nullptr;
};
void launch_pp_one(u32* assign, const u32* load, u32 m, const u64* cap, const u32* alive_bits, u64* used,
                   const u32* idx, const u32* req, u32 n, u32* out_node, u32* out_flag, u32* status, hipStream_t s,
                   u32* aff_life = nullptr, u32* done = nullptr, u32 seq = 0, const SmallInline* inl = nullptr, u32 n_obj_chk = 0,
                   void* stage = nullptr, unsigned int* ticket = nullptr, u32 sa = 0, bool host_io = false,
                   const CrudSmallArgs* crud = nullptr);
void launch_crud_small(const CrudSmallArgs& c, u32* assign, const u32* load, u32 m, u64* used_or_null, u32* aff_life, u32* done,
                       u32 seq, hipStream_t s);
// --- place_pending, the general request path (k_ppm_first / k_ppm_gather / solve of the virtual table / k_ppm_output) ---
// bad: device word, 0 between calls (raised by k_ppm_first on an invalid entry, put back by k_ppm_output's last workgroup);
// s_idx / s_req: the library's padded device copies of the requests, written on the way (use THEM afterwards: idx / req may be
// mapped host memory, or a caller's exact-size device arrays the solve's tile-wide reads must not run past);
// dead_bits / vflag != nullptr: some node is not alive — nodes that requests run into are marked, REPLACED bits per request
void launch_ppm_first(const u32* assign, u64 n_obj, u32 m, const u32* alive_bits, const u32* idx, const u32* req, u64 n,
                      u32* pos_scratch, u32* s_idx, u32* s_req, u32* dead_bits, u32* vflag, u32* bad, hipStream_t s);
void launch_ppm_gather(const u32* assign, const u32* load, const u32* idx, u64 n, const u32* pos_scratch, u32* vcur, u32* vload,
                       u32* vfirst, const u32* bad, hipStream_t s);
// *status (mapped host memory): 0 done | 1 the solve needs the fix-up and it was not enqueued (fixup_done == false): nothing
// was changed, enqueue it and launch this again with fixup_done | 3 invalid entry: nothing was changed.  Ends with the
// several-workgroup completion word (ticket / done / seq).
void launch_ppm_output(u32* assign, u64 n_obj, const u32* idx, const u32* req, u64 n, const u32* vcur, const u32* vnext,
                       const u32* vfirst, const u32* vflag, u32* pos_scratch, const u32* alive_bits, const SolveBufs& b,
                       const Plan& vp, u32* out_node, u32* out_flag, u32* aff_life, u32* bad, bool fixup_done, u32* status,
                       unsigned int* ticket, u32* done, u32 seq, hipStream_t s);

// place_pending over a window-sorted batch (big batches): see k_pp_win_gather.  scratch = part_scratch_words(n_obj, n) words.
bool pp_win_applicable(u64 n_obj, u64 n, const void* idx, const void* req);
// claim_fast: [m + 1] u64 of device memory — the load the batch's first touches put on every requester + the count of requests
// the window kernel could not answer by itself; cleared by launch_pp_bin, filled by launch_pp_win_gather, judged by
// launch_pp_win_verdict (*verdict: 1 every answer is final, `used` has taken the claims | 2 the batch needs the solve over the
// records, nothing else was changed | 3 invalid entry), handed out by launch_pp_win_unsort
void launch_pp_bin(u64 n_obj, u32 m, const u32* idx, const u32* req, u64 n, u32* scratch, DevStats* st, u32* host_err, hipStream_t s,
                   u32* dead_bits, u64* claim_fast);  // (clears the two)
// ans0 / ans1: the answer records' two words, [n] each, in the SORTED order of launch_pp_bin's records (launch_pp_win_unsort
// carries them back to batch order: the caller's columns on verdict 1, vrec on verdict 2)
void launch_pp_win_gather(u32* assign, const u32* load, u64 n_obj, u32 m, const u32* alive_bits, u64 n, const u32* scratch,
                          u32* ans0, u32* ans1, u32* dead_bits, u32* aff_life, const DevStats* st, u64* claim_fast, hipStream_t s);
void launch_pp_win_unsort(u64 n_obj, const u32* scratch, const u32* ans0, const u32* ans1, u64 n, uint2* vrec, u32* out_node, u32* out_flag,
                          const u32* verdict, hipStream_t s);
void launch_pp_win_verdict(u32 m, const u64* cap, const u32* alive_bits, u64* used, const u64* claim_fast, const DevStats* st,
                           u32* verdict_dev, u32* verdict_host, hipStream_t s);
void launch_pp_win_output(const u32* idx, const u32* req, u64 n, const u32* vcur, const u32* vload, const u32* vnext,
                          const u32* alive_bits, const u32* cutidx, u32 m, u32* out_node, u32* out_flag, u32* aff_life,
                          const DevStats* st, hipStream_t s, u32 sa, const uint2* vrec);

size_t scan_lds_bytes(u32 m);

// --- row-sharded solve: records exchanged between ranks ---
inline size_t shard_words1(u32 m) { return 2 * (size_t)m + 8; }  // X = [kept_local[m] | claim_local[m] | 8 counters]
inline size_t shard_words2(u32 m) { return (size_t)m + 2; }      // Y = [delta[m] | spill load | spill rows]
void launch_shard_pack1(const Plan& p, const SolveBufs& b, u64* X, hipStream_t s);
// Xg[R][shard_words1(m)] = the all-gathered X records (RCCL paths)
void launch_shard_import(const Plan& p, const NodeTab& nt, const SolveBufs& b, const u64* Xg, u32 rank, u32 R,
                         u64* gprev, u64* gfinal, u64* verdict_dev, u64* verdict_host, hipStream_t s);
// peer-to-peer exchange over xGMI (windows IPC-mapped by every rank, R <= 32)
// the whole fast-path exchange in one launch (k_resolve_xchg): window rows of shard_xchg_words(m) words; host_partial =
// pinned rows [resolve_blocks(m)][8] of partial verdicts (the caller adds them up)
inline size_t shard_xchg_words(u32 m) { return 2 * (2 * (size_t)m + 8 * (size_t)((m + 7) / 8)); }  // two tagged words per value
void launch_resolve_xchg(const Plan& p, const NodeTab& nt, const SolveBufs& b, u64* const* d_peers, u32 R, u32 rank,
                         size_t my_row_off, const u64* win_rows, size_t W, u64 seq, u64* p2p_err, u64* gprev, u64* gfinal,
                         u64* host_partial, u32 co_resident, hipStream_t s);
void launch_p2p_put(const u64* src, u32 words, u64* const* d_peers, u32 R, size_t data_off, size_t flag_off, u64 seq,
                    hipStream_t s);
void launch_p2p_wait_copy(const u64* win_slot, size_t W, u32 R, u32 words, const u64* flags, u64 seq, u64* err, u64* out,
                          hipStream_t s);
void launch_shard_export_delta(const Plan& p, const SolveBufs& b, const u64* base, int wsp_sel, u64* Y, hipStream_t s);
// asynchronous row-sharded tick: one exchange of the fix-up record in two launches (export + put | wait + import), both
// no-ops — no store to a peer, no wait for one — when the solve needs no fix-up anywhere (DevStats::global_slow)
void launch_shard_export_put(const Plan& p, const SolveBufs& b, const u64* base, int wsp_sel, u64* const* d_peers, u32 R,
                             size_t data_off, size_t flag_off, u64 seq, hipStream_t s);
void launch_shard_wait_import(const Plan& p, const SolveBufs& b, const u64* win_slot, size_t W, const u64* flags, u64 seq,
                              u64* err, u32 rank, u32 R, u64* gprev, const u64* gfinal, u64* verdict_dev, u64* verdict_host,
                              hipStream_t s);
void launch_shard_tick_stats(const Plan& p, const SolveBufs& b, u64* out_host, u64 mark, hipStream_t s);
void launch_shard_import_delta(const Plan& p, const SolveBufs& b, const u64* Yg, u32 rank, u32 R, u64* gprev,
                               u64* verdict_dev, u64* verdict_host, hipStream_t s);

}  // namespace riogp
