// placement_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the batched object-placement solver.
//
// What is replaced (paths relative to /root/reference):
//   LocalObjectPlacement::{lookup,update,remove,clean_server}  rio-rs/src/object_placement/local.rs:22-68
//   Service::get_or_create_placement / check_address_mismatch  rio-rs/src/service.rs:193-298
// by kernels over a dense table in HBM (object: cur/load/aff u32 columns; node: cap/alive/used).
//
// Everything here is integer/index work bounded by HBM bandwidth — no MFMA.  Design rules used
// (guides: cdna_hip_programming.md §2, §6 G2/G11/G12/G13; MI355X_MICROARCH.md §LDS, §price list):
//   * 64-wide waves, every column access is a coalesced dwordx4 (16 B/lane, 1 KiB per wave-instr);
//   * each wave owns a CONTIGUOUS row range, so "index order" = (block, wave, iteration, lane,
//     element) order and ordered prefix sums never need a block-wide barrier in the streaming loop;
//   * one 1024-thread workgroup per CU (256 workgroups = 256 CUs = 8 XCDs x 32) so a solve has
//     only 256 per-node histograms to reduce; per-node counters live in LDS (ds_add_u64), never
//     in global atomics on the streaming path;
//   * inter-workgroup dependencies are cut at kernel boundaries (per-XCD L2s are not coherent),
//     never hand-rolled grid barriers;
//   * all cross-row decisions are integer sums and index-ordered prefixes: results do not depend
//     on dispatch order, so the output is bit-identical to the sequential oracle.
#include "placement_kernels.h"
#include <cstdlib>

#include <hip/hip_ext.h>

namespace riogp {

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------

// Cross-lane helpers.  Scans and sums run on DPP (row_shr within a row of 16 lanes, row_bcast:15 / :31 across rows —
// gfx9 wave64 only): VALU moves of a few cycles each instead of ds_bpermute round trips through the LDS crossbar
// (12 dependent bpermutes per u64 scan were the latency floor of every ordered-prefix search).
template <int CTRL>
__device__ __forceinline__ u64 dpp64(u64 v) {  // lanes without a valid source read 0
    const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)v, CTRL, 0xf, 0xf, false);
    const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v >> 32), CTRL, 0xf, 0xf, false);
    return ((u64)hi << 32) | lo;
}
template <int CTRL>
__device__ __forceinline__ u32 dpp32(u32 v) {
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ u64 shfl_up64(u64 v, int d) {  // general shuffles (ds_bpermute): off the hot loops
    u32 lo = __shfl_up((u32)v, d, 64), hi = __shfl_up((u32)(v >> 32), d, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_xor64(u64 v, int d) {
    u32 lo = __shfl_xor((u32)v, d, 64), hi = __shfl_xor((u32)(v >> 32), d, 64);
    return ((u64)hi << 32) | lo;
}
// value of a WAVE-UNIFORM lane (src must be the same in every lane: a constant or derived from a ballot)
__device__ __forceinline__ u64 shfl64(u64 v, int src) {
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, src);
    const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), src);
    return ((u64)hi << 32) | lo;
}
// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ u64 wave_incl_scan(u64 v, int lane) {
    const int rl = lane & 15;
    u64 t;
    t = dpp64<0x111>(v); if (rl >= 1) v += t;               // row_shr:1
    t = dpp64<0x112>(v); if (rl >= 2) v += t;               // row_shr:2
    t = dpp64<0x114>(v); if (rl >= 4) v += t;               // row_shr:4
    t = dpp64<0x118>(v); if (rl >= 8) v += t;               // row_shr:8
    t = dpp64<0x142>(v); if ((lane & 31) >= 16) v += t;     // row_bcast:15
    t = dpp64<0x143>(v); if (lane >= 32) v += t;            // row_bcast:31
    return v;
}
__device__ __forceinline__ u64 sat_add(u64 a, u64 b) {
    u64 s = a + b;
    return s < a ? ~0ull : s;
}
__device__ __forceinline__ u64 wave_incl_scan_sat(u64 v, int lane) {
    const int rl = lane & 15;
    u64 t;
    t = dpp64<0x111>(v); if (rl >= 1) v = sat_add(v, t);
    t = dpp64<0x112>(v); if (rl >= 2) v = sat_add(v, t);
    t = dpp64<0x114>(v); if (rl >= 4) v = sat_add(v, t);
    t = dpp64<0x118>(v); if (rl >= 8) v = sat_add(v, t);
    t = dpp64<0x142>(v); if ((lane & 31) >= 16) v = sat_add(v, t);
    t = dpp64<0x143>(v); if (lane >= 32) v = sat_add(v, t);
    return v;
}
// sum over the wave, the same value in every lane (readlane 63 of the scan)
__device__ __forceinline__ u64 wave_sum(u64 v) {
    return shfl64(wave_incl_scan(v, (int)(threadIdx.x & 63)), 63);
}
__device__ __forceinline__ u32 wave_sum32(u32 v) {
    const int lane = (int)(threadIdx.x & 63), rl = lane & 15;
    u32 t;
    t = dpp32<0x111>(v); if (rl >= 1) v += t;
    t = dpp32<0x112>(v); if (rl >= 2) v += t;
    t = dpp32<0x114>(v); if (rl >= 4) v += t;
    t = dpp32<0x118>(v); if (rl >= 8) v += t;
    t = dpp32<0x142>(v); if ((lane & 31) >= 16) v += t;
    t = dpp32<0x143>(v); if (lane >= 32) v += t;
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}
// Balanced, index-ordered split of the table: wave gw owns tiles [gw*tiles/nw, (gw+1)*tiles/nw).
// Division-free: tiles = tq * nw + tr, so gw * tiles / nw = gw * tq + floor(gw * tr / nw); gw <= nw <= 4 096 keeps gw * tr
// below 2^24, where the multiply-shift by div_magic = ceil(2^38 / nw) is exact (error < 2^-14 < 1 / nw).  A 64-bit
// division is ~100 instructions, twice per wave in every kernel's prologue (measured: ~1 us of a 10 us fix-up kernel).
__host__ __device__ __forceinline__ u64 wave_row_lo(const Plan& p, u64 gw) {
    const u64 x = gw * p.tr;
    return (gw * p.tq + ((x * p.div_magic) >> 38)) * kTile;
}
__device__ __forceinline__ void wave_range(const Plan& p, u64 gw, u64& wstart, u64& wend) {
    wstart = wave_row_lo(p, gw);
    wend = wave_row_lo(p, gw + 1);
    if (wend > p.n) wend = p.n;
    if (wstart > wend) wstart = wend;
    if (p.wcnt) {  // packed fix-up: only the first wcnt[gw] positions of the range hold rows
        const u64 e = wstart + p.wcnt[gw];
        if (e < wend) wend = e;
    }
}
// ... of the whole table, whatever Plan::wcnt says (no load behind a branch: see k_cut_apply's step loop)
__device__ __forceinline__ void wave_range_plain(const Plan& p, u64 gw, u64& wstart, u64& wend) {
    wstart = wave_row_lo(p, gw);
    wend = wave_row_lo(p, gw + 1);
    if (wend > p.n) wend = p.n;
    if (wstart > wend) wstart = wend;
}
// the wave whose range contains row position i (inverse of wave_row_lo), and whether i is a live packed position
__device__ __forceinline__ bool packed_live(const Plan& p, u64 i) {
    if (!p.wcnt) return true;
    const u64 tile = i / kTile;
    u64 gw = ((tile + 1) * p.nw - 1) / p.tiles;  // (off the hot paths: the unfused cut search only)
    return i - wave_row_lo(p, gw) < p.wcnt[gw];
}
__device__ __forceinline__ u64 block_row_lo(const Plan& p, u32 b) { return wave_row_lo(p, (u64)b * kWaves); }
constexpr int kSmall = 128;  // bytes of small per-block scratch at the head of the dynamic LDS region
__device__ __forceinline__ bool bit_of(const u32* bits, u32 j) { return (bits[j >> 5] >> (j & 31)) & 1u; }

// Phase traces (lab build only; compiled out of the product): workgroup b stores wall_clock64() (100 MHz) at phase
// boundary `slot` of kernel table `tab` into g_kt[tab][b][slot] when the plan's trace flag is set.
#ifdef RIO_GP_LAB
constexpr int kKtTables = 8;
__device__ u64 g_kt[kKtTables][kMaxBlocks * 8];
static int g_trace_host = 0;
#define RIOGP_KT(pl, tab, slot) do { if (threadIdx.x == 0 && (pl).trace && blockIdx.x < kMaxBlocks) g_kt[tab][(size_t)blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
#define RIOGP_KTF(flag, tab, slot) do { if (threadIdx.x == 0 && (flag) && blockIdx.x < kMaxBlocks) g_kt[tab][(size_t)blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
int ktrace_enable(int on) { g_trace_host = on; return 0; }
int ktrace_read(int table, u64* out) {
    if (table < 0 || table >= kKtTables) return -1;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kt), sizeof(u64) * kMaxBlocks * 8, sizeof(u64) * kMaxBlocks * 8 * (size_t)table);
}
static inline u32 trace_flag() { return (u32)g_trace_host; }
#else
#define RIOGP_KT(pl, tab, slot) do { } while (0)
#define RIOGP_KTF(flag, tab, slot) do { } while (0)
static inline u32 trace_flag() { return 0; }
#endif

Plan make_plan(u64 n, u32 m, u32 max_blocks) {
    Plan p;
    p.n = n;
    p.m = m;
    p.mwords = (m + 31) / 32;
    p.wcnt = nullptr;
    p.alive_dst = nullptr;
    p.trace = trace_flag();
    p.sa = 0;
    p.mark = 1;
    if (max_blocks == 0 || max_blocks > kMaxBlocks) max_blocks = kMaxBlocks;
    u64 tiles = (n + kTile - 1) / kTile;
    if (tiles == 0) tiles = 1;
    u64 g = (tiles + kWaves - 1) / kWaves;
    if (g > max_blocks) g = max_blocks;
    p.tiles = tiles;
    p.G = (u32)g;
    p.nw = p.G * kWaves;
    p.tq = tiles / p.nw;
    p.tr = (u32)(tiles % p.nw);
    p.div_magic = ((1ull << 38) + p.nw - 1) / p.nw;
    // rows of the largest block: ceil(tiles*16/nw) tiles (+1 for the floor/ceil jitter of the split)
    const u64 max_block_tiles = (tiles * kWaves + p.nw - 1) / p.nw + 1;
    const u64 sub_tiles = (max_block_tiles + kMaxSubs - 1) / kMaxSubs;
    p.sub = (u32)(sub_tiles * kTile);
    p.subs = (u32)((max_block_tiles * kTile + p.sub - 1) / p.sub);
    if (p.subs > kMaxSubs) p.subs = kMaxSubs;
    return p;
}

// the split as the kernels compute it, for the host-side check of the multiply-shift against the plain division
u64 plan_wave_row_lo(u64 n, u32 m, u32 gw, u32* nw_out) {
    const Plan p = make_plan(n, m, 0);
    if (nw_out) *nw_out = p.nw;
    return wave_row_lo(p, gw);
}

__host__ __device__ __forceinline__ size_t scan_lds_bytes_dev(u32 m) {
    const u32 mwords = (m + 31) / 32;
    const size_t b = kSmall + ((size_t)2 * m + 2) * sizeof(u64) + (size_t)((mwords + 3) & ~3u) * sizeof(u32) + 64;
    return (b + 15) & ~(size_t)15;
}
size_t scan_lds_bytes(u32 m) { return scan_lds_bytes_dev(m); }
// Row classification shared by every streaming kernel (must be identical everywhere):
//   kept      placed on a live node            -> sticky               (service.rs:241-242)
//   claimant  pending, affinity node is live   -> first touch          (service.rs:244-252)
//   else      pending, goes to the water-fill — unless its affinity is kAffInactive (not an object: class 3, ignored)
// VIRT (virtual table of place_pending): rows are requests; "kept" = already placed (dead nodes
// were evicted beforehand), kSkipMark rows are duplicate requests and take no part.
template <bool VIRT>
__device__ __forceinline__ int classify(u32 c, u32 a, u32 m, const u32* alv, u32 sa = 0) {
    if (VIRT) {
        if (c == kSkipMark) return 3;
        if (c < m) return 0;
    } else {
        if (c < m && bit_of(alv, c)) return 0;
    }
    if (a < m && (sa || bit_of(alv, a))) return 1;
    if (!VIRT && a == kAffInactive) return 3;  // not an object: takes no part
    return 2;
}

// ------------------------------------------------------------------------------------------------
// K1  k_scan — THE streaming kernel: one pass over cur/load/aff (12 B/row read), optimistic
//     write of the new assignment (4 B/row), per-block per-node load histograms in LDS.
//     Algorithmic traffic 16 B/row; everything else is <3 % overhead:
//     H lines 2*m*8 B per block, 3 words per wave.
//     The row body is branch-free (one ds_add_u64 per row, trash bin for rows that add nothing;
//     counters are wave-uniform popcounts of ballots) so the next tile's three dwordx4 loads stay
//     in flight under a counted vmcnt while the current tile is processed.
//     A row whose affinity is kAffInactive is not an object (row lifecycle of the string layer:
//     never interned, removed, or dropped by clean_server): it is kept if it happens to be placed
//     on a live node and takes no part otherwise — neither claimant nor spill candidate.
// ------------------------------------------------------------------------------------------------
// Answer record of one request, 8 bytes at its batch position (k_pp_win_gather -> k_pp_win_split, or -> k_scan<COMPACT 3> when the
// batch needs the solve): word 0 = node (16 bits, 0xFFFF = none) | flag << 16 (8 bits) | later request of its object << 31;
// word 1 = the row's load (first request of an object) or the batch position of the first request (later ones).
__host__ __device__ __forceinline__ u32 pp_ans(u32 node, u32 flag, bool later) {
    return (node == kNone ? 0xFFFFu : (node & 0xFFFFu)) | (flag << 16) | (later ? 0x80000000u : 0u);
}
__host__ __device__ __forceinline__ u32 pp_ans_node(u32 w) { return (w & 0xFFFFu) == 0xFFFFu ? kNone : (w & 0xFFFFu); }
__host__ __device__ __forceinline__ u32 pp_ans_flag(u32 w) { return (w >> 16) & 0xFFu; }
// ... as the `cur` column of the virtual table: a later request takes no part, a row its first request placed (or could not
// place) was pending, anything else is kept where it is
__host__ __device__ __forceinline__ u32 pp_ans_cur(u32 w) {
    return (w >> 31) ? kSkipMark : ((pp_ans_flag(w) & 0xFu) >= 2u ? kNone : pp_ans_node(w));
}

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x4u __attribute__((ext_vector_type(4), aligned(4)));  // four consecutive words at any word address
constexpr u64 kChainTimeoutTicks = 100000000ull;  // wall_clock64 runs at 100 MHz: 1 s, then ScanChain::err is raised
constexpr u32 kStageCap = 320;  // words per column of a wave's packing ring: < 64 left over + one tile (256) of new records
constexpr u32 kIncFlush = 256, kIncCap = 512;  // k_inc_scan's rings: records per flush (four per lane), words per ring column

// NT: non-temporal column streams for tables beyond the 256 MiB Infinity Cache (measured +1-2 % at 40-100 M rows and
// -25 % at 10 M rows, where the cache serves part of every pass: launch_scan picks by table size)
template <bool NT>
__device__ __forceinline__ uint4 ld4(const u32* p) {
    if (NT) {
        const u32x4 r = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        return make_uint4(r.x, r.y, r.z, r.w);
    }
    return *reinterpret_cast<const uint4*>(p);
}
// CHAIN (chained quiet ticks, k_scan): a workgroup's rows of the assignment columns are handed from one LAUNCH to the next
// while both run, to whatever CU and XCD the next tick's workgroup sits on.  Both sides go through to the fabric: 16-byte
// `sc0 sc1` stores (same cost as plain ones; the 8-byte agent atomics the compiler would pick cost 2.7x per byte) and
// `sc0 sc1` loads (never served by a vector L1 or by another XCD's stale line) — buffer instructions, because they are the
// 16-byte accesses that take a cache policy from C++ and are counted in vmcnt by the compiler like any other.
typedef u32 u32x4b __attribute__((ext_vector_type(4)));
constexpr int kAuxSc0Sc1 = 17;  // cache policy operand of the raw buffer builtins on gfx940+: sc0 = 1, nt = 2, sc1 = 16
__device__ __forceinline__ __amdgpu_buffer_rsrc_t col_rsrc(const u32* col) {  // raw buffer over a whole column: byte offsets, no bound
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(col), 0, 0xFFFFFFFFu, 0x00020000);
}
__device__ __forceinline__ uint4 ld4_fabric(__amdgpu_buffer_rsrc_t r, u64 row) {
    const u32x4b v = __builtin_amdgcn_raw_buffer_load_b128(r, (u32)(row * 4u), 0, kAuxSc0Sc1);
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <bool NT>
__device__ __forceinline__ uint4 ld4_o32(const u32* base, u64 row) {  // (chained scan: tables below 2^30 rows — a 32-bit byte offset off a uniform base)
    const u32 off = (u32)row * 4u;
    return ld4<NT>(reinterpret_cast<const u32*>(reinterpret_cast<const char*>(base) + off));
}
__device__ __forceinline__ void st4_fabric(__amdgpu_buffer_rsrc_t r, u64 row, const uint4 v) {
    u32x4b x;
    x.x = v.x; x.y = v.y; x.z = v.z; x.w = v.w;
    __builtin_amdgcn_raw_buffer_store_b128(x, r, (u32)(row * 4u), 0, kAuxSc0Sc1);
}
template <bool NT>
__device__ __forceinline__ void st4(u32* p, const uint4 v) {
    if (NT) {
        u32x4 r;
        r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
        __builtin_nontemporal_store(r, reinterpret_cast<u32x4*>(p));
    } else {
        *reinterpret_cast<uint4*>(p) = v;
    }
}

template <bool VIRT, bool ALLALIVE, bool CHECK, int COMPACT = 0, bool NT = false, bool CHAIN = false>
__device__ __forceinline__ void scan_tile(const uint4 cv, const uint4 av, const uint4 lv, const u64 i0, const u64 wend,
                                          const u32 m, const u32* alv, u64* hist, u32* __restrict__ next,
                                          u64& sp_sum, u32& sp_cnt, u32& kept_cnt, u32& evict_cnt, u32& claim_cnt,
                                          const PackOut* pk = nullptr, u64* pk_pos = nullptr, u32* stage = nullptr,
                                          u32* st_head = nullptr, u32* st_fill = nullptr,
                                          const uint4 xv = make_uint4(0, 0, 0, 0), const u32 sa = 0,
                                          const __amdgpu_buffer_rsrc_t* nrs = nullptr) {
    // COMPACT == 3 (virtual table of a big place_pending batch): rows are requests, xv = the objects they ask for; a claimant's
    // optimistic node also goes straight into the REAL assignment column, pk->next[object] (one scattered store per first touch:
    // the fix-up patches the same rows through the same indices, so no pass carries the decisions back afterwards) — unless
    // the window kernel has already written it there, densely (pk->load != nullptr: k_pp_win_gather writes the requester of
    // every pending row's first request when that requester is ALIVE; a requester that is not — RIO_GP_CFG_REF_SELF_ASSIGN —
    // must wait for clean_server, which runs between the two kernels, and is stored here)
    uint4 ov;
    u64 sp_local = 0;
    u32 any_sp = 0, pm = 0;
#define RIOGP_ROW(C, A, L, O, E, X)                                                                    \
    {                                                                                                  \
        const bool inr = !CHECK || (i0 + E < wend);                                                    \
        const bool cin = C < m, ain = A < m;                                                           \
        const u32 cc = cin ? C : 0, aa = ain ? A : 0;                                                  \
        const bool skip = VIRT && C == kSkipMark;                                                      \
        const bool dead = !VIRT && A == kAffInactive;                                                  \
        const bool kept = inr && cin && (VIRT || ALLALIVE || bit_of(alv, cc));                         \
        const bool cl = inr && !kept && !skip && ain && (ALLALIVE || sa || bit_of(alv, aa));           \
        const bool sp = inr && !kept && !cl && !skip && !dead;                                         \
        const u32 bin = (kept && !VIRT) ? cc : (cl ? m + aa : 2 * m);                                  \
        atomicAdd(&hist[bin], (u64)L);                                                                 \
        O = kept ? C : (cl ? A : (skip ? kSkipMark : (dead ? kNone : kSpillMark)));                    \
        kept_cnt += (u32)__popcll(__ballot(kept));                                                     \
        claim_cnt += (u32)__popcll(__ballot(cl));                                                      \
        if (!VIRT) evict_cnt += (u32)__popcll(__ballot(inr && !kept && C != kNone));                   \
        sp_local += sp ? (u64)L : 0;                                                                   \
        any_sp |= sp;                                                                                  \
        if (COMPACT == 1 || COMPACT == 2) pm |= (u32)(cl | sp) << E;                                   \
        if (COMPACT == 3 && cl && !(pk->load && (ALLALIVE || bit_of(alv, aa)))) pk->next[X] = A;      \
    }
    RIOGP_ROW(cv.x, av.x, lv.x, ov.x, 0, xv.x)
    RIOGP_ROW(cv.y, av.y, lv.y, ov.y, 1, xv.y)
    RIOGP_ROW(cv.z, av.z, lv.z, ov.z, 2, xv.z)
    RIOGP_ROW(cv.w, av.w, lv.w, ov.w, 3, xv.w)
#undef RIOGP_ROW
    if (COMPACT == 1 || COMPACT == 2) {  // pending rows of this tile, in index order (lane-major, then element), to the wave's packed cursor
        const u64 b0 = __ballot(pm & 1u), b1 = __ballot(pm & 2u), b2 = __ballot(pm & 4u), b3 = __ballot(pm & 8u);
        if (b0 | b1 | b2 | b3) {
            const u64 lt = (1ull << (threadIdx.x & 63)) - 1ull;
            const u32 rank = (u32)(__popcll(b0 & lt) + __popcll(b1 & lt) + __popcll(b2 & lt) + __popcll(b3 & lt));
            const u32 cnt = (u32)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));
            if (COMPACT == 2) {
                // Staged: the records go to this wave's LDS ring (four columns of kStageCap words) first; whenever 64 are
                // waiting they leave as four fully coalesced 256-byte stores.  The direct form below issues up to sixteen
                // masked store instructions per tile with a handful of scattered lanes each: measured 37-40 us for the
                // churn-tick scan against 26 us for the plain scan.
                u32 e = *st_head + *st_fill + rank;
#define RIOGP_PK(E, A, L, O)                                                                              \
                if (pm & (1u << E)) {                                                                     \
                    const u32 x = e >= kStageCap ? (e >= 2 * kStageCap ? e - 2 * kStageCap : e - kStageCap) : e;  \
                    stage[x] = (u32)(i0 + E); stage[kStageCap + x] = L; stage[2 * kStageCap + x] = A;     \
                    stage[3 * kStageCap + x] = O; ++e;                                                    \
                }
                RIOGP_PK(0, av.x, lv.x, ov.x)
                RIOGP_PK(1, av.y, lv.y, ov.y)
                RIOGP_PK(2, av.z, lv.z, ov.z)
                RIOGP_PK(3, av.w, lv.w, ov.w)
#undef RIOGP_PK
                *st_fill += cnt;
                __builtin_amdgcn_wave_barrier();
                while (*st_fill >= 64u) {  // wave-uniform
                    u32 x = *st_head + (threadIdx.x & 63);
                    x = x >= kStageCap ? x - kStageCap : x;
                    const u32 v0 = stage[x], v1 = stage[kStageCap + x], v2 = stage[2 * kStageCap + x], v3 = stage[3 * kStageCap + x];
                    const u64 o = *pk_pos + (threadIdx.x & 63);
                    pk->idx[o] = v0; pk->load[o] = v1; pk->aff[o] = v2; pk->next[o] = v3;
                    *st_head = *st_head + 64u >= kStageCap ? *st_head + 64u - kStageCap : *st_head + 64u;
                    *st_fill -= 64u;
                    *pk_pos += 64u;
                    __builtin_amdgcn_wave_barrier();
                }
            } else {
                u64 pos = *pk_pos + rank;
#define RIOGP_PK(E, A, L, O)                                                                     \
                if (pm & (1u << E)) { pk->idx[pos] = (u32)(i0 + E); pk->load[pos] = L; pk->aff[pos] = A; pk->next[pos] = O; ++pos; }
                RIOGP_PK(0, av.x, lv.x, ov.x)
                RIOGP_PK(1, av.y, lv.y, ov.y)
                RIOGP_PK(2, av.z, lv.z, ov.z)
                RIOGP_PK(3, av.w, lv.w, ov.w)
#undef RIOGP_PK
                *pk_pos += cnt;
            }
        }
    }
    const u64 spmask = __ballot(any_sp);
    if (spmask) {  // wave-uniform, never taken on the fast path
        sp_sum += sp_local;
        // exact row count of spill candidates in this tile
        u32 c = 0;
        c += (!CHECK || i0 + 0 < wend) && ov.x == kSpillMark;
        c += (!CHECK || i0 + 1 < wend) && ov.y == kSpillMark;
        c += (!CHECK || i0 + 2 < wend) && ov.z == kSpillMark;
        c += (!CHECK || i0 + 3 < wend) && ov.w == kSpillMark;
        sp_cnt += c;
    }
    if (!CHECK || i0 + 3 < wend) {
        if (CHAIN) st4_fabric(*nrs, i0, ov);
        else st4<NT>(next + i0, ov);
    } else if (CHAIN) {  // (the ragged last tile of the table: single words, agent scope)
        if (i0 + 0 < wend) __hip_atomic_store(next + i0 + 0, ov.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (i0 + 1 < wend) __hip_atomic_store(next + i0 + 1, ov.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (i0 + 2 < wend) __hip_atomic_store(next + i0 + 2, ov.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        if (i0 + 0 < wend) next[i0 + 0] = ov.x;
        if (i0 + 1 < wend) next[i0 + 1] = ov.y;
        if (i0 + 2 < wend) next[i0 + 2] = ov.z;
    }
}

// Where workgroup b's sums for node group g (8 nodes) live in H: one 128-byte line {kept x8 | claim x8}.
__host__ __device__ __forceinline__ size_t h_line(u32 g, u32 b, u32 G) { return ((size_t)g * G + b) * 16; }

// TPI = tiles (of 256 rows) a wave processes per loop iteration; the next TPI tiles are always in
// flight while the current ones are processed.
template <bool VIRT, bool ALLALIVE, int TPI, int COMPACT = 0, bool NT = false, bool CHAIN = false>
__global__ __launch_bounds__(kBlock, CHAIN ? 8 : 1) void k_scan(const u32* __restrict__ cur, const u32* __restrict__ load,
                                                 const u32* __restrict__ aff, u32* __restrict__ next,
                                                 const u32* __restrict__ alive_bits, Plan p, u64* __restrict__ H,
                                                 u64* __restrict__ blkstat, u64* __restrict__ wsp_sum,
                                                 u32* __restrict__ wsp_cnt, DevStats* __restrict__ stats, PackOut pko,
                                                 FxRows fx, u64* __restrict__ bsp_sum, u32* __restrict__ bsp_cnt,
                                                 u64* __restrict__ R, u64* __restrict__ RP, ScanChain ch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 m = p.m;
    u32* bst = reinterpret_cast<u32*>(smem);                 // [4] (first 128 B: small scratch, G17)
    u64* hist = reinterpret_cast<u64*>(smem + kSmall);       // [2m + 2] kept-by-cur | claim-by-aff | trash
    u32* alv = reinterpret_cast<u32*>(hist + 2 * m + 2);     // [mwords]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: row ranges in SGPRs
    RIOGP_KT(p, 3, 0);
    const u64 gw = (u64)blockIdx.x * kWaves + wave;
    u64 wstart, wend;
    wave_range(p, gw, wstart, wend);
    // COMPACT 3: pko.wcnt = the batch's invalid-entry counter; set -> the virtual table was not built: solve an empty one
    // (the plain virtual table of the general request path: the same word, when the caller passes one)
    if ((COMPACT == 3 || (VIRT && COMPACT == 0 && pko.wcnt)) && *reinterpret_cast<const volatile u32*>(pko.wcnt)) wend = wstart;
    const u64 span = wend > wstart ? wend - wstart : 0;
    const u64 wfull = wstart + (span / kTile) * kTile;                  // end of the full tiles
    const u64 wgrp = wstart + (span / (kTile * TPI)) * (kTile * TPI);   // end of the full TPI-tile groups

    // first group of loads goes out BEFORE the LDS set-up and its barrier: HBM latency overlaps both
    u64 it = wstart;
    uint4 cv[TPI], av[TPI], lv[TPI], xv[TPI];  // (xv: COMPACT == 3 only — the objects the rows of a virtual table ask for)
    // COMPACT == 3 with pko.aff set: the virtual table arrives as 8-byte records {cur | load} (k_pp_win_gather's vrec); the
    // two halves of a lane's 32 bytes travel in cv / lv, are told apart where they are used, and leave again as the two
    // columns `cur` / `load` the kernels behind this one read (what a split pass of its own did: 28 us per 10 M requests)
    const bool vrec = COMPACT == 3 && pko.aff != nullptr;
    const u32* const csrc = vrec ? pko.aff : cur;
    const u32* const lsrc = vrec ? pko.aff + 4 : load;
    const int vsh = vrec ? 1 : 0;  // (records: 2 words per row)
    const __amdgpu_buffer_rsrc_t crs = col_rsrc(cur), nrs = col_rsrc(next);  // (CHAIN only: the two assignment columns through the fabric)
    if (it < wgrp) {
#pragma unroll
        for (int q = 0; q < TPI; ++q) {
            const u64 i = it + (u64)q * kTile + (u64)lane * 4;
            if (!CHAIN) cv[q] = ld4<NT>(csrc + (i << vsh));  // (CHAIN: the previous tick's workgroup may still be writing them)
            av[q] = CHAIN ? ld4_o32<NT>(aff, i) : ld4<NT>(aff + i);
            lv[q] = CHAIN ? ld4_o32<NT>(load, i) : ld4<NT>(lsrc + (i << vsh));
            if (COMPACT == 3) xv[q] = ld4<NT>(pko.idx + i);
        }
    }
    auto unrec = [&](uint4& c, uint4& l, const u64 i) {  // records -> columns (and out to the column arrays)
        if (!vrec) return;
        const uint4 a = c, b = l;
        c = make_uint4(pp_ans_cur(a.x), pp_ans_cur(a.z), pp_ans_cur(b.x), pp_ans_cur(b.z));  // (answer records: k_pp_win_gather)
        l = make_uint4(a.y, a.w, b.y, b.w);
        *reinterpret_cast<uint4*>(const_cast<u32*>(cur) + i) = c;
        *reinterpret_cast<uint4*>(const_cast<u32*>(load) + i) = l;
    };

    for (u32 k = tid; k < 2 * m + 2; k += kBlock) hist[k] = 0;
    if (!ALLALIVE || (p.alive_dst && blockIdx.x == 0))
        for (u32 k = tid; k < p.mwords; k += kBlock) {
            const u32 w = alive_bits[k];
            if (!ALLALIVE) alv[k] = w;
            if (p.alive_dst && blockIdx.x == 0) p.alive_dst[k] = w;  // a fresh bitmap (mapped host memory) -> the device array
        }
    if (tid < 4) bst[tid] = 0;
    u64& bsum = *reinterpret_cast<u64*>(smem + 32);          // spill-candidate load of the whole block
    if (tid == 0) bsum = 0;
    if (fx.dev && tid < 8) fx.dev[(size_t)blockIdx.x * 8 + tid] = 0;  // this workgroup's row of the fix-up counters
    // claim load the cuts reject, for k_fill's ordered spill prefix: row blockIdx.x of RP (by node group, written by k_resolve
    // workgroups that own cut nodes) and the cut block's correction R[blockIdx.x] (k_cut_find) start at zero
    if (R && tid == 0) R[blockIdx.x] = 0;
    if (RP) {  // (this workgroup's share of the [node group][block] words: contiguous)
        const u32 ngr = (m + 7) >> 3;
        for (u32 k = tid; k < ngr; k += kBlock) RP[(size_t)blockIdx.x * ngr + k] = 0;
    }
    if (blockIdx.x == 0 && tid == 0) {  // accumulators the fix-up kernels add into
        stats->rejected = 0; stats->load_rejected = 0;
        stats->spilled = 0; stats->load_spilled = 0; stats->unplaced = 0; stats->load_unplaced = 0;
        stats->rounds_run = 0;
        stats->n_cut = 0;  // device-side "a node has a cut" flag (k_resolve / k_cutblk)
        stats->local_fixup = 0;
        stats->global_slow = 0;
    }
    auto chain_wait = [&](const u32* flag) {  // (one lane; bounded: must not hang the device if the predecessor never comes)
        const u64 t0 = wall_clock64();
        u32 tries = 0;
        while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ch.wait) < 0) {
            if ((++tries & 15u) == 0 && wall_clock64() - t0 > kChainTimeoutTicks) {
                __hip_atomic_store(ch.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            if (ch.per_wave) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(1);
        }
    };
    // Chained quiet ticks: this workgroup's rows of `cur` are what workgroup blockIdx.x of the previous tick's scan wrote (the same
    // plan), and its rows of `next` are what that workgroup read — it may still be running, on the other scan stream.  Wait for
    // ITS flag, not for its launch.  (It is resident or finished: two workgroups of this kernel fit a CU and at most two launches
    // of the chain are in flight.)  per_wave: the same per wave range — a flag per wave, no barrier on the hand-over's path.
    if (CHAIN && tid == 0 && ch.wait && !ch.per_wave) chain_wait(ch.flags + blockIdx.x);
    __syncthreads();
    if (CHAIN && ch.per_wave && ch.wait && lane == 0) chain_wait(ch.flags + kMaxBlocks + gw);
    if (CHAIN && it < wgrp) {
#pragma unroll
        for (int q = 0; q < TPI; ++q) cv[q] = ld4_fabric(crs, it + (u64)q * kTile + (u64)lane * 4);
    }

    u64 sp_sum = 0;
    u32 sp_cnt = 0, kept_cnt = 0, evict_cnt = 0, claim_cnt = 0;  // kept/evict/claim are wave-uniform
    u64 pk_pos = wstart;  // COMPACT: this wave's packed write cursor (wave-uniform)
    // COMPACT == 2: this wave's packing ring sits behind the histogram region (launch_scan sizes the dynamic LDS for it)
    u32* stage = COMPACT == 2 ? reinterpret_cast<u32*>(smem + scan_lds_bytes_dev(p.m)) + (size_t)wave * 4 * kStageCap : nullptr;
    u32 st_head = 0, st_fill = 0;

    while (it < wgrp) {
        const u64 nit = it + (u64)kTile * TPI;
        // Software prefetch of the next group — UNCONDITIONAL so the loop body is straight-line and these
        // loads stay in flight while the current group is processed; on the wave's last iteration the
        // address is clamped to the group being processed (an L1/L2 hit, no HBM traffic, no over-read).
        const u64 pit = nit < wgrp ? nit : it;
        uint4 cn[TPI], an[TPI], ln[TPI], xn[TPI];
#pragma unroll
        for (int q = 0; q < TPI; ++q) {
            const u64 i = pit + (u64)q * kTile + (u64)lane * 4;
            cn[q] = CHAIN ? ld4_fabric(crs, i) : ld4<NT>(csrc + (i << vsh));
            an[q] = CHAIN ? ld4_o32<NT>(aff, i) : ld4<NT>(aff + i);
            ln[q] = CHAIN ? ld4_o32<NT>(load, i) : ld4<NT>(lsrc + (i << vsh));
            if (COMPACT == 3) xn[q] = ld4<NT>(pko.idx + i);
        }
#pragma unroll
        for (int q = 0; q < TPI; ++q) {
            if (COMPACT == 3) unrec(cv[q], lv[q], it + (u64)q * kTile + (u64)lane * 4);
            scan_tile<VIRT, ALLALIVE, false, COMPACT, NT, CHAIN>(cv[q], av[q], lv[q], it + (u64)q * kTile + (u64)lane * 4, wend,
                                                          m, alv, hist, next, sp_sum, sp_cnt, kept_cnt, evict_cnt,
                                                          claim_cnt, &pko, &pk_pos, stage, &st_head, &st_fill,
                                                          COMPACT == 3 ? xv[q] : make_uint4(0, 0, 0, 0), p.sa, &nrs);
        }
        it = nit;
#pragma unroll
        for (int q = 0; q < TPI; ++q) { cv[q] = cn[q]; av[q] = an[q]; lv[q] = ln[q]; if (COMPACT == 3) xv[q] = xn[q]; }
    }
    for (; it < wend; it += kTile) {  // leftover full tiles (< TPI) and the ragged last tile of the table
        const u64 i = it + (u64)lane * 4;
        uint4 c1 = CHAIN ? ld4_fabric(crs, i) : *reinterpret_cast<const uint4*>(csrc + (i << vsh));
        const uint4 a1 = *reinterpret_cast<const uint4*>(aff + i);
        uint4 l1 = *reinterpret_cast<const uint4*>(lsrc + (i << vsh));
        const uint4 x1 = COMPACT == 3 ? *reinterpret_cast<const uint4*>(pko.idx + i) : make_uint4(0, 0, 0, 0);
        if (COMPACT == 3) unrec(c1, l1, i);
        if (it < wfull)
            scan_tile<VIRT, ALLALIVE, false, COMPACT, NT, CHAIN>(c1, a1, l1, i, wend, m, alv, hist, next, sp_sum, sp_cnt, kept_cnt,
                                                          evict_cnt, claim_cnt, &pko, &pk_pos, stage, &st_head, &st_fill, x1, p.sa, &nrs);
        else
            scan_tile<VIRT, ALLALIVE, true, COMPACT, NT, CHAIN>(c1, a1, l1, i, wend, m, alv, hist, next, sp_sum, sp_cnt, kept_cnt,
                                                         evict_cnt, claim_cnt, &pko, &pk_pos, stage, &st_head, &st_fill, x1, p.sa, &nrs);
    }
    if (COMPACT == 2 && st_fill) {  // what is left in the ring (< 64 records)
        u32 x = st_head + (u32)lane;
        x = x >= kStageCap ? x - kStageCap : x;
        if ((u32)lane < st_fill) {
            const u64 o = pk_pos + (u32)lane;
            pko.idx[o] = stage[x]; pko.load[o] = stage[kStageCap + x]; pko.aff[o] = stage[2 * kStageCap + x];
            pko.next[o] = stage[3 * kStageCap + x];
        }
        pk_pos += st_fill;
    }

    // per-wave spill-candidate totals (index-ordered prefix over wave ranges comes later)
    sp_sum = wave_sum(sp_sum);
    sp_cnt = wave_sum32(sp_cnt);
    if (CHAIN) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's row stores (through to the fabric) have landed — inline asm: no
                                                          // pass may decide the counter is known to be empty and drop the wait
        if (ch.per_wave && lane == 0) __hip_atomic_store(ch.flags + kMaxBlocks + gw, ch.set, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) {
        wsp_sum[gw] = sp_sum;
        wsp_cnt[gw] = sp_cnt;
        if (COMPACT == 1 || COMPACT == 2) pko.wcnt[gw] = (u32)(pk_pos - wstart);
        atomicAdd(&bst[0], kept_cnt);
        atomicAdd(&bst[1], evict_cnt);
        atomicAdd(&bst[2], claim_cnt);
        atomicAdd(&bst[3], sp_cnt);
        if (sp_sum) atomicAdd(&bsum, sp_sum);
    }
    __syncthreads();
    if (CHAIN && tid == 0 && !ch.per_wave)  // this workgroup's rows are done, read and written (every wave has drained its stores in front of the
                            // barrier, and they went through to the fabric): the next tick's workgroup blockIdx.x may go
        __hip_atomic_store(ch.flags + blockIdx.x, ch.set, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) { bsp_sum[blockIdx.x] = bsum; bsp_cnt[blockIdx.x] = bst[3]; }
    // this block's sums, node-group major: line (g, b) = {kept of nodes 8g..8g+7 | their claim loads} — 16 consecutive
    // threads store one 128-byte line, so k_resolve's workgroup g reads G contiguous lines and nothing else
    const u32 ng = (m + 7) >> 3;
    for (u32 k = tid; k < ng * 16; k += kBlock) {
        const u32 g = k >> 4, c = k & 15, j = g * 8 + (c & 7);
        H[h_line(g, blockIdx.x, p.G) + c] = j < m ? hist[(c >> 3) * m + j] : 0ull;
    }
    if (tid < 4) blkstat[(size_t)blockIdx.x * 4 + tid] = bst[tid];
    RIOGP_KT(p, 3, 7);
}

// ------------------------------------------------------------------------------------------------
// K1i k_inc_scan — the scan of a COMMITTED tick over a table that is mostly placed (the clean_server rebalance stream,
//     BASELINE config 5; local.rs:51-58 + service.rs:227-252 batched).  It reads what k_scan reads — cur, load, aff,
//     12 B/row, coalesced — and differs in what it does NOT do:
//       * the assignment column is updated IN PLACE (the tick is committed: nobody is promised the table as it was), and only
//         where a row's value changes here: a claimant takes its affinity node, an evicted row that is not an object becomes
//         NONE — one 16-byte store per lane that holds such a row.  Kept rows are not rewritten (k_scan: 4 B/row), and a
//         spill candidate's row is left to the water-fill's last round, which writes it whatever becomes of it;
//       * no kept histogram: the kept load of node j is the committed used[j] where j is alive and 0 where it is not
//         (k_resolve takes it from there: ResolveArgs::kept_from — valid whenever the library's `used` vector is), so only
//         the pending rows (a tenth of the table per churn tick) cost an LDS operation at all;
//       * no claim histogram either: the pending rows are packed {row, load, affinity}, in index order, to the front of the
//         wave's range of the pack columns (k_scan<COMPACT>'s rings), and k_rebal deals them out evenly to the fix-up's
//         workgroups and builds the per-block histograms and spill totals over THAT layout.
//     Why the load / affinity columns are streamed and not gathered for the pending rows only: the rows a churn tick evicts
//     are not clustered at the granularity that matters — after 60 ticks of the config-5 stream 73 % of the 128-byte lines
//     and 58 % of the 64-byte segments of a column hold one (97 % / 81 % in the first ticks), and a gather per pending row is
//     two million random accesses at 17-20 ns chip-wide each: measured 34 us for the kernel against 21 for the stream.
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ size_t inc_lds_base(u32 m) {  // small | alive bitmap (k_inc_scan's fixed part of the LDS)
    return (kSmall + (size_t)(((m + 31) / 32 + 3) & ~3u) * sizeof(u32) + 63) & ~(size_t)63;
}
template <int TPI, bool NT>
__global__ __launch_bounds__(kBlock) void k_inc_scan(u32* __restrict__ assign, const u32* __restrict__ load,
                                                     const u32* __restrict__ aff, const u32* __restrict__ alive_bits, Plan p,
                                                     u64* __restrict__ blkstat, DevStats* __restrict__ stats, PackOut pko,
                                                     FxRows fx, u64* __restrict__ R, u64* __restrict__ RP) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 m = p.m;
    u32* bst = reinterpret_cast<u32*>(smem);                 // [4]
    u32* alv = reinterpret_cast<u32*>(smem + kSmall);        // [mwords]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    RIOGP_KT(p, 3, 0);
    const u64 gw = (u64)blockIdx.x * kWaves + wave;
    u64 wstart, wend;
    wave_range(p, gw, wstart, wend);
    const u64 span = wend > wstart ? wend - wstart : 0;
    const u64 wgrp = wstart + (span / (kTile * TPI)) * (kTile * TPI);   // end of the full TPI-tile groups
    u64 it = wstart;
    uint4 cv[TPI], av[TPI], lv[TPI];
    if (it < wgrp) {
#pragma unroll
        for (int q = 0; q < TPI; ++q) {
            const u64 i = it + (u64)q * kTile + (u64)lane * 4;
            cv[q] = ld4<NT>(assign + i);
            av[q] = ld4<NT>(aff + i);
            lv[q] = ld4<NT>(load + i);
        }
    }
    for (u32 k = tid; k < p.mwords; k += kBlock) {
        const u32 w = alive_bits[k];
        alv[k] = w;
        if (p.alive_dst && blockIdx.x == 0) p.alive_dst[k] = w;  // a fresh bitmap (mapped host memory) -> the device array
    }
    if (tid < 4) bst[tid] = 0;
    if (fx.dev && tid < 8) fx.dev[(size_t)blockIdx.x * 8 + tid] = 0;
    if (R && tid == 0) R[blockIdx.x] = 0;
    if (RP) {
        const u32 ngr = (m + 7) >> 3;
        for (u32 k = tid; k < ngr; k += kBlock) RP[(size_t)blockIdx.x * ngr + k] = 0;
    }
    if (blockIdx.x == 0 && tid == 0) {
        stats->rejected = 0; stats->load_rejected = 0;
        stats->spilled = 0; stats->load_spilled = 0; stats->unplaced = 0; stats->load_unplaced = 0;
        stats->rounds_run = 0;
        stats->n_cut = 0;
        stats->local_fixup = 0;
        stats->global_slow = 0;
    }
    __syncthreads();
    RIOGP_KT(p, 3, 1);

    u32 kept_cnt = 0, evict_cnt = 0, claim_cnt = 0, sp_cnt = 0;  // per LANE here (summed over the wave at the end)
    u64 pk_pos = wstart;  // this wave's packed write cursor (wave-uniform)
    // this wave's packing ring: three columns of kStageCap words behind the bitmap
    // (kIncCap words per column: < kIncFlush left over + one tile of new records; a flush is kIncFlush records, four per lane)
    u32* stage = reinterpret_cast<u32*>(smem + inc_lds_base(m)) + (size_t)wave * 3 * kIncCap;
    u32 st_head = 0, st_fill = 0;
    const u64 lt = (1ull << lane) - 1ull;

#ifdef RIO_GP_LAB   // timing experiments only (results are wrong): trace flag bit 1 = no packing, bit 2 = no in-place stores
    const bool dbg_nopack = (p.trace & 2u) != 0, dbg_nostore = (p.trace & 4u) != 0;
#else
    constexpr bool dbg_nopack = false, dbg_nostore = false;
#endif
    // one tile: classify, write the lane's vector back if a row of it changed, pack the rows that go on
    auto tile = [&](const uint4 c, const uint4 a, const uint4 l, const u64 i0) {
        uint4 ov = c;
        u32 km = 0;
        bool chg = false;
#define RIOGP_ROW(C, A, L, O, E)                                                      \
        {                                                                             \
            const bool inr = i0 + E < wend;                                           \
            const bool cin = C < m, ain = A < m;                                      \
            const bool kept = inr && cin && bit_of(alv, cin ? C : 0u);                \
            const bool pend = inr && !kept;                                           \
            const bool ev = pend && C != kNone;                                       \
            const bool dead = A == kAffInactive;                                      \
            const bool cl = pend && ain && (p.sa || bit_of(alv, ain ? A : 0u));       \
            const bool sp = pend && !cl && !dead;                                     \
            O = cl ? A : ((ev && dead) ? kNone : C);                                  \
            chg |= cl | (ev && dead);                                                 \
            kept_cnt += (u32)kept; evict_cnt += (u32)ev;                              \
            claim_cnt += (u32)cl; sp_cnt += (u32)sp;                                  \
            km |= (u32)(cl | sp) << E;                                                \
        }
        RIOGP_ROW(c.x, a.x, l.x, ov.x, 0)
        RIOGP_ROW(c.y, a.y, l.y, ov.y, 1)
        RIOGP_ROW(c.z, a.z, l.z, ov.z, 2)
        RIOGP_ROW(c.w, a.w, l.w, ov.w, 3)
#undef RIOGP_ROW
        if (chg && !dbg_nostore) *reinterpret_cast<uint4*>(assign + i0) = ov;  // (a lane's rows past the end of the table are padding)
        if (dbg_nopack) km = 0;
        const u64 b0 = __ballot(km & 1u), b1 = __ballot(km & 2u), b2 = __ballot(km & 4u), b3 = __ballot(km & 8u);
        if (b0 | b1 | b2 | b3) {  // (wave-uniform) index order = lane-major, then element
            u32 e = st_head + st_fill + (u32)(__popcll(b0 & lt) + __popcll(b1 & lt) + __popcll(b2 & lt) + __popcll(b3 & lt));
#define RIOGP_PK(E, A, L)                                                                                 \
            if (km & (1u << E)) {                                                                         \
                const u32 x = e & (kIncCap - 1u);                                                         \
                stage[x] = (u32)(i0 + E); stage[kIncCap + x] = L; stage[2 * kIncCap + x] = A;             \
                ++e;                                                                                      \
            }
            RIOGP_PK(0, a.x, l.x)
            RIOGP_PK(1, a.y, l.y)
            RIOGP_PK(2, a.z, l.z)
            RIOGP_PK(3, a.w, l.w)
#undef RIOGP_PK
            st_fill += (u32)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));
            __builtin_amdgcn_wave_barrier();
            if (st_fill >= kIncFlush) {  // wave-uniform: 256 records leave as three coalesced 1 KB stores, four records per lane
                // (a flush is a store instruction per column every ~10 tiles instead of one every ~2.5: the stores of a wave
                //  retire in order with its loads, so every flush is a moment at which the prefetch is waited for together
                //  with the stores in front of it — same-run: 6 us of the kernel's span were these moments)
                const u32 x = st_head + 4u * (u32)lane;   // (st_head is 0 or kIncFlush: no wrap inside a flush)
                const u64 o = pk_pos + 4u * (u32)lane;
                *reinterpret_cast<uint4*>(pko.idx + o) = *reinterpret_cast<const uint4*>(stage + x);
                *reinterpret_cast<uint4*>(pko.load + o) = *reinterpret_cast<const uint4*>(stage + kIncCap + x);
                *reinterpret_cast<uint4*>(pko.aff + o) = *reinterpret_cast<const uint4*>(stage + 2 * kIncCap + x);
                st_head = (st_head + kIncFlush) & (kIncCap - 1u);
                st_fill -= kIncFlush;
                pk_pos += kIncFlush;
                __builtin_amdgcn_wave_barrier();
            }
        }
    };

    while (it < wgrp) {
        const u64 nit = it + (u64)kTile * TPI;
        const u64 pit = nit < wgrp ? nit : it;  // (the last iteration re-requests its own group: a hit, no over-read)
        uint4 cn[TPI], an[TPI], ln[TPI];
#pragma unroll
        for (int q = 0; q < TPI; ++q) {
            const u64 i = pit + (u64)q * kTile + (u64)lane * 4;
            cn[q] = ld4<NT>(assign + i);
            an[q] = ld4<NT>(aff + i);
            ln[q] = ld4<NT>(load + i);
        }
#pragma unroll
        for (int q = 0; q < TPI; ++q) tile(cv[q], av[q], lv[q], it + (u64)q * kTile + (u64)lane * 4);
        it = nit;
#pragma unroll
        for (int q = 0; q < TPI; ++q) { cv[q] = cn[q]; av[q] = an[q]; lv[q] = ln[q]; }
    }
    for (; it < wend; it += kTile) {  // leftover tiles (< TPI) and the ragged last tile of the table
        const u64 i = it + (u64)lane * 4;
        const uint4 c1 = *reinterpret_cast<const uint4*>(assign + i);
        const uint4 a1 = *reinterpret_cast<const uint4*>(aff + i);
        const uint4 l1 = *reinterpret_cast<const uint4*>(load + i);
        tile(c1, a1, l1, i);
    }
    if (st_fill) {  // what is left in the ring (< kIncFlush records)
        for (u32 r = (u32)lane; r < st_fill; r += 64u) {
            const u32 x = (st_head + r) & (kIncCap - 1u);
            const u64 o = pk_pos + r;
            pko.idx[o] = stage[x]; pko.load[o] = stage[kIncCap + x]; pko.aff[o] = stage[2 * kIncCap + x];
        }
        pk_pos += st_fill;
    }
    RIOGP_KT(p, 3, 2);
    kept_cnt = wave_sum32(kept_cnt);
    evict_cnt = wave_sum32(evict_cnt);
    claim_cnt = wave_sum32(claim_cnt);
    sp_cnt = wave_sum32(sp_cnt);
    if (lane == 0) {
        pko.wcnt[gw] = (u32)(pk_pos - wstart);
        atomicAdd(&bst[0], kept_cnt);
        atomicAdd(&bst[1], evict_cnt);
        atomicAdd(&bst[2], claim_cnt);
        atomicAdd(&bst[3], sp_cnt);
    }
    __syncthreads();
    if (tid < 4) blkstat[(size_t)blockIdx.x * 4 + tid] = bst[tid];
    RIOGP_KT(p, 3, 7);
}

// ------------------------------------------------------------------------------------------------
// K1r k_rebal — deals the pending rows k_inc_scan<false> extracted out EVENLY, in index order, and is the "scan" of the
//     fix-up's table: the rows sit, per wave range of the real table, at the front of the range (counts in src.wcnt); their
//     exclusive prefix gives every row a dense position d, and wave v of the balanced table pv (uniform wave ranges of
//     ctiles tiles) takes the dense tiles [v T/nw, (v+1) T/nw) of the T = ceil(P/256) tiles.  A churn tick's evicted rows sit
//     in clusters (a node's rows are index runs: the interval water-fill put them there), so a workgroup of the row-range
//     decomposition holds up to four times the mean and every fix-up kernel takes as long as its busiest workgroup; over
//     the balanced table every workgroup of the cut search and of the water-fill rounds holds P/G rows.  Index order =
//     (wave, position) order is the same in both layouts, so the solve is the same solve.
//     Per balanced block: claim histograms -> H, spill totals, optimistic `next` (claimant -> affinity, else spill mark) —
//     what k_scan builds for its own blocks.
// ------------------------------------------------------------------------------------------------
__device__ u64 block_excl_scan_1024(u64 v, bool saturating, u64* lds_part /*[16]*/, u64* total);  // (defined with the cut search)
__host__ __device__ __forceinline__ size_t rebal_lds_bytes(u32 m, u32 nw) {
    return scan_lds_bytes_dev(m) + ((size_t)nw + 8) * sizeof(u32) + 16 * sizeof(u64);
}
__global__ __launch_bounds__(kBlock) void k_rebal(const u32* __restrict__ s_idx, const u32* __restrict__ s_load,
                                                  const u32* __restrict__ s_aff, const u32* __restrict__ s_wcnt, Plan p, Plan pv,
                                                  const u32* __restrict__ alive_bits, u32* __restrict__ d_idx,
                                                  u32* __restrict__ d_load, u32* __restrict__ d_aff, u32* __restrict__ d_next,
                                                  u32* __restrict__ d_wcnt, u64* __restrict__ H, u64* __restrict__ wsp_sum,
                                                  u32* __restrict__ wsp_cnt, u64* __restrict__ bsp_sum, u32* __restrict__ bsp_cnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 m = p.m, nw = p.nw;
    u32* bst = reinterpret_cast<u32*>(smem);                 // [4]
    u64* hist = reinterpret_cast<u64*>(smem + kSmall);       // [2m + 2]
    u32* alv = reinterpret_cast<u32*>(hist + 2 * m + 2);     // [mwords]
    u32* pre = reinterpret_cast<u32*>(smem + scan_lds_bytes_dev(m));  // [nw + 1] rows packed before source wave s
    u64* part = reinterpret_cast<u64*>(pre + nw + 8);        // [16] block-scan partials (nw is a multiple of 16: aligned)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the source counts first (the only dependent global read of the prologue), four per thread
    RIOGP_KT(p, 5, 0);
    const uint4 cq = *reinterpret_cast<const uint4*>(s_wcnt + (size_t)tid * 4);  // (the array holds kMaxBlocks * kWaves words whatever nw is)
    u32 c4[4] = {cq.x, cq.y, cq.z, cq.w};
    for (u32 k = tid; k < 2 * m + 2; k += kBlock) hist[k] = 0;
    for (u32 k = tid; k < p.mwords; k += kBlock) alv[k] = alive_bits[k];
    if (tid < 4) bst[tid] = 0;
    u64& bsum = *reinterpret_cast<u64*>(smem + 32);
    if (tid == 0) bsum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if ((u32)tid * 4u + (u32)k >= nw) c4[k] = 0;
    u64 total = 0;
    u64 ex = block_excl_scan_1024((u64)c4[0] + c4[1] + c4[2] + c4[3], false, part, &total);  // (two barriers inside)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32 g4 = (u32)tid * 4u + (u32)k;
        if (g4 <= nw) pre[g4] = (u32)ex;
        ex += c4[k];
    }
    if (tid == kBlock - 1 && nw == (u32)kBlock * 4u) pre[nw] = (u32)total;
    __syncthreads();
    RIOGP_KT(p, 5, 1);
    const u32 P = (u32)total;
    const u32 T = (P + (u32)kTile - 1u) / (u32)kTile;
    const u32 tq = T / nw, tr = T - tq * nw;
    const u64 gw = (u64)blockIdx.x * kWaves + wave;
    const u64 tlo = gw * tq + ((gw * tr * p.div_magic) >> 38), thi = (gw + 1) * tq + (((gw + 1) * tr * p.div_magic) >> 38);
    const u64 dlo = tlo * kTile < P ? tlo * kTile : P, dhi = thi * kTile < P ? thi * kTile : P;
    const u64 obase = wave_row_lo(pv, gw);  // this wave's range of the balanced columns
    u64 sp_sum = 0;
    u32 sp_cnt = 0;  // per lane
    // source wave of the range's first row: the last s with pre[s] <= dlo (wave-uniform)
    u32 s_lo = 0;
    {
        u32 lo = 0, hi = nw;
        while (hi - lo > 1) {
            const u32 mid = lo + ((hi - lo) >> 1);
            if (pre[mid] <= (u32)dlo) lo = mid; else hi = mid;
        }
        s_lo = lo;
    }
    for (u64 d0 = dlo; d0 < dhi; d0 += kTile) {
        const u32 d = (u32)d0 + (u32)lane * 4u;
        // this lane's source wave: the last s in [s_lo, nw) with pre[s] <= d
        u32 s = s_lo;
        {
            u32 lo = s_lo, hi = nw;
            while (hi - lo > 1) {
                const u32 mid = lo + ((hi - lo) >> 1);
                if (pre[mid] <= d) lo = mid; else hi = mid;
            }
            s = lo;
        }
        u32 ix[4] = {0, 0, 0, 0}, ld[4] = {0, 0, 0, 0}, af[4] = {kNone, kNone, kNone, kNone};
        if (d + 3u < (u32)dhi && d + 3u < pre[s + 1 < nw ? s + 1 : nw]) {
            // the lane's four rows lie in ONE source wave's piece (the rule on big tables, where a wave range holds thousands
            // of pending rows): three 16-byte reads at a 4-byte-aligned address instead of twelve 4-byte ones
            const u64 at = wave_row_lo(p, s) + (d - pre[s]);
            const u32x4u vi = *reinterpret_cast<const u32x4u*>(s_idx + at), vl = *reinterpret_cast<const u32x4u*>(s_load + at),
                         va = *reinterpret_cast<const u32x4u*>(s_aff + at);
            ix[0] = vi.x; ix[1] = vi.y; ix[2] = vi.z; ix[3] = vi.w;
            ld[0] = vl.x; ld[1] = vl.y; ld[2] = vl.z; ld[3] = vl.w;
            af[0] = va.x; af[1] = va.y; af[2] = va.z; af[3] = va.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const u32 de = d + (u32)e;
                if (de < (u32)dhi) {
                    while (s + 1 < nw && pre[s + 1] <= de) ++s;  // (source waves without rows are stepped over)
                    const u64 at = wave_row_lo(p, s) + (de - pre[s]);
                    ix[e] = s_idx[at]; ld[e] = s_load[at]; af[e] = s_aff[at];
                }
            }
        }
        uint4 nx = make_uint4(kNone, kNone, kNone, kNone);
#define RIOGP_ROW(E, O)                                                               \
        if (d + E < (u32)dhi) {                                                       \
            const u32 A = af[E];                                                      \
            const bool ain = A < m;                                                   \
            const bool cl = ain && (p.sa || bit_of(alv, ain ? A : 0u));               \
            O = cl ? A : kSpillMark;                                                  \
            if (cl) atomicAdd(&hist[m + A], (u64)ld[E]);                              \
            sp_cnt += (u32)!cl;                                                       \
            sp_sum += cl ? 0ull : (u64)ld[E];                                         \
        }
        RIOGP_ROW(0, nx.x)
        RIOGP_ROW(1, nx.y)
        RIOGP_ROW(2, nx.z)
        RIOGP_ROW(3, nx.w)
#undef RIOGP_ROW
        const u64 o = obase + (d0 - dlo) + (u64)lane * 4;  // (whole vectors: the columns are padded, rows past the count are never read as rows)
        *reinterpret_cast<uint4*>(d_idx + o) = make_uint4(ix[0], ix[1], ix[2], ix[3]);
        *reinterpret_cast<uint4*>(d_load + o) = make_uint4(ld[0], ld[1], ld[2], ld[3]);
        *reinterpret_cast<uint4*>(d_aff + o) = make_uint4(af[0], af[1], af[2], af[3]);
        *reinterpret_cast<uint4*>(d_next + o) = nx;
        // the next tile starts where this tile's last row came from
        const u32 s_last = (u32)__builtin_amdgcn_readlane((int)s, 63);
        s_lo = s_last;
    }
    RIOGP_KT(p, 5, 2);
    sp_sum = wave_sum(sp_sum);
    sp_cnt = wave_sum32(sp_cnt);
    if (lane == 0) {
        wsp_sum[gw] = sp_sum;
        wsp_cnt[gw] = sp_cnt;
        d_wcnt[gw] = (u32)(dhi - dlo);
        atomicAdd(&bst[3], sp_cnt);
        if (sp_sum) atomicAdd(&bsum, sp_sum);
    }
    __syncthreads();
    if (tid == 0) { bsp_sum[blockIdx.x] = bsum; bsp_cnt[blockIdx.x] = bst[3]; }
    const u32 ng = (m + 7) >> 3;
    for (u32 k = tid; k < ng * 16; k += kBlock) {
        const u32 g = k >> 4, c = k & 15, j = g * 8 + (c & 7);
        H[h_line(g, blockIdx.x, pv.G) + c] = (c >= 8 && j < m) ? hist[m + j] : 0ull;
    }
    RIOGP_KT(p, 5, 7);
}

// ------------------------------------------------------------------------------------------------
// K2  k_resolve — per node: used = sum over blocks of the kept histogram, claim total, free, and
//     the verdict "claims fit" (fast path) or "cut" (fix-up needed).  Workgroup g owns node group g
//     (8 nodes) and reads exactly its G contiguous 128-byte lines of H ({kept x8 | claim x8} per
//     block, written that way by k_scan): algorithmic = fetched bytes.  Thread = (row group, column pair): one dwordx4
//     per line and thread, every load issued before the first wait.  Per-workgroup partial counters go straight into the
//     caller's pinned host slot (plain stores, no atomics, no fences, no copy kernel); the host adds the rows up.
//     For a node whose claims exceed its free capacity the block in which the ordered claim prefix crosses it is found
//     from the columns still in registers, and the claim load the cut rejects BEHIND that block goes into R[] (k_fill's
//     ordered spill prefix needs it per block).
//     SEARCH (packed fix-up: the pending rows of every wave sit at the front of its range, k_scan<COMPACT>): the exact cut
//     row is found here as well — the cut block's pending rows are a few thousand, so a pair of waves per node reads them
//     with every load in flight at once (one round trip) instead of a launch of its own (k_cut_find: 14 us per churn tick).
// ------------------------------------------------------------------------------------------------
constexpr int kResNodes = 8;
constexpr int kResRowGroups = 32;                          // 256 threads = 32 row groups x 8 column pairs
constexpr int kResRows = kMaxBlocks / kResRowGroups;       // 8 lines (16 B of each) per thread
constexpr int kSrchSlots = 10;                             // tiles (256 rows) the in-resolve search holds in registers per batch

// sums of this workgroup's 16 columns over the G blocks -> tot[16] (kept x8 | claim x8); v[r][0..1] keep the thread's
// own addends (columns 2*cp, 2*cp+1 of rows rg + 32 r) for the cut-block search.  Threads >= 256 only take the barriers.
__device__ __forceinline__ void resolve_column_sums(const u64* __restrict__ H, u32 g, u32 G, u64 (&v)[kResRows][2],
                                                    u64 (*part)[16], u64* tot) {
    const int tid = threadIdx.x, cp = tid & 7, rg = (tid >> 3) & (kResRowGroups - 1);
    const bool act = tid < 256;
#pragma unroll
    for (int r = 0; r < kResRows; ++r) {
        const u32 row = rg + r * kResRowGroups;
        if (act && row < G) {
            const uint4 x = *reinterpret_cast<const uint4*>(H + h_line(g, row, G) + 2 * cp);
            v[r][0] = ((u64)x.y << 32) | x.x;
            v[r][1] = ((u64)x.w << 32) | x.z;
        } else {
            v[r][0] = 0;
            v[r][1] = 0;
        }
    }
    u64 s0 = 0, s1 = 0;
#pragma unroll
    for (int r = 0; r < kResRows; ++r) { s0 += v[r][0]; s1 += v[r][1]; }
    if (act) {
        part[rg][2 * cp] = s0;
        part[rg][2 * cp + 1] = s1;
    }
    __syncthreads();
    if (tid < 16) {
        u64 t = 0;
#pragma unroll
        for (int q = 0; q < kResRowGroups; ++q) t += part[q][tid];
        tot[tid] = t;
    }
    __syncthreads();
}

struct ResolveArgs {
    const u64* H; const u64* blkstat; Plan p;
    const u64* cap; const u32* alive_bits; const u64* used_base;
    u64* used_kept; u64* used_cur; u64* claim_tot; u32* cutblk; u32* cutidx;
    u64* partial; u64* host_partial; u64* budget; u64* admpre; DevStats* stats;
    u64* RP;           // [node groups][G] claim load the cuts of a node group reject in the blocks BEFORE block b (nullptr: not
                       // wanted — row-sharded local sums): k_fill's round 0 sums column b.  Plain stores, one writer per row.
    u64* D;            // [kFillRounds][m] per-round admitted loads of k_fill: zeroed here (nullptr: none)
    u64* fold_into;    // committed `used` still waiting for the previous committed solve's D rows: folded in before they are zeroed
    u32 fold_rounds;
    // SEARCH: the packed pending rows (affinity, load) of every wave range
    const u32* pk_aff; const u32* pk_load;
    u64* Tg;           // [m][16] claim load of a cut node's undecided rows per wave of its cut block (k_cut_apply adds, k_cut_settle
                       // reads): the rows of the nodes that have a cut are zeroed here (nullptr: none)
    // the scan built no kept histogram (k_inc_scan): the kept load of node j is this committed vector's entry (after the fold
    // above, when it is the vector folded into) where j is alive, 0 where it is not; nullptr: the column sums of H
    const u64* kept_from;
};

template <bool SEARCH>
__global__ __launch_bounds__(SEARCH ? kBlock : 256) void k_resolve(const ResolveArgs a) {
    __shared__ u64 part[kResRowGroups][16];
    __shared__ u64 tot[16];
    __shared__ u64 red[8];
    __shared__ u64 cutfre[kResNodes];               // free capacity of a node of this workgroup that has a cut
    __shared__ u32 cutmask;                         // which of the eight nodes have one
    __shared__ u64 colbuf[kResNodes][kMaxBlocks];   // their claim columns, row order (only filled when cutmask != 0)
    __shared__ u32 s_cb[kResNodes];                 // cut block
    __shared__ u64 s_bud[kResNodes], s_adm[kResNodes], s_used[kResNodes];  // budget at its start | admitted before it | kept load
    __shared__ u64 s_half[kResNodes];               // SEARCH: claim load of the node in the first half of its cut block
    __shared__ u32 s_row[kResNodes];                // SEARCH: cut row found by the first half (kNoCut: not there)
    __shared__ u64 s_in[kResNodes];                 // SEARCH: load admitted inside the cut block
    const Plan& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, cp = tid & 7, rg = (tid >> 3) & (kResRowGroups - 1);
    const u32 m = p.m, G = p.G, nb = gridDim.x, g = blockIdx.x;
    const u32 j = g * kResNodes + (u32)tid;          // node of thread tid < 8
    const bool valid = tid < kResNodes && j < m;
    // node-table operands of the verdict are requested up front so their latency overlaps the H loads
    u64 cj = 0, ub = 0, kf = 0;
    bool alive_j = false;
    if (valid) {
        cj = a.cap[j];
        alive_j = bit_of(a.alive_bits, j);
        if (a.used_base) ub = a.used_base[j];
        if (a.kept_from) kf = a.kept_from[j];
        // the rounds' admitted-load vectors: what the previous committed solve left goes into the committed `used` first
        if (a.D) {
            if (a.fold_into) {
                u64 f = a.fold_into[j];
                for (u32 r = 0; r < a.fold_rounds; ++r) f += a.D[(size_t)r * m + j];
                a.fold_into[j] = f;
                if (a.kept_from == a.fold_into) kf = f;
            }
            for (u32 r = 0; r < kFillRounds; ++r) a.D[(size_t)r * m + j] = 0;
        }
    }
    if (tid < 8) { red[tid] = 0; s_row[tid] = kNoCut; s_half[tid] = 0; s_in[tid] = 0; }
    if (tid == 0) cutmask = 0;
    // this wave-1 lane's first word of the k_scan row counters (rows g, g + nb, ... of blkstat) is requested here, ahead of
    // the H loads, instead of after the two barriers below: one dependent round trip less in a 4 us kernel
    u64 bs_acc = 0;
    u32 bs_r = G;
    if (tid >= 64 && tid < 128) {
        bs_r = g + nb * (lane >> 2);
        if (bs_r < G) bs_acc = a.blkstat[(size_t)bs_r * 4 + (lane & 3)];
    }
    if (SEARCH) RIOGP_KT(p, 0, 0);
    u64 v[kResRows][2];
    resolve_column_sums(a.H, g, G, v, part, tot);      // two barriers inside: red / cutmask are published
    if (SEARCH) RIOGP_KT(p, 0, 1);
    if (valid) {
        const u64 kept_load = a.kept_from ? (alive_j ? kf : 0ull) : tot[tid], ctot = tot[tid + kResNodes];
        const u64 used = kept_load + ub;
        const u64 fre = ((alive_j || p.sa) && cj > used) ? cj - used : 0;  // (what the claimants may take)
        a.used_kept[j] = used;
        a.claim_tot[j] = ctot;
        a.cutblk[j] = kNoCut;
        a.cutidx[j] = kNoCut;
        a.used_cur[j] = used + ctot;  // final unless the node has a cut (the cut search rewrites it)
        s_used[tid] = used;
        atomicAdd(&red[0], kept_load);
        atomicAdd(&red[1], ctot);
        if (ctot > fre) {
            atomicAdd(&red[2], 1ull);
            cutfre[tid] = fre;
            atomicOr(&cutmask, 1u << tid);
        }
    }
    if (tid >= 64 && tid < 128) {  // slice of the k_scan row counters: rows g, g+nb, ... of blkstat
        u64 acc = bs_acc;
        if (bs_r < G)  // (more than one row per lane only for m < 128)
            for (u32 r = bs_r + nb * 16; r < G; r += nb * 16) acc += a.blkstat[(size_t)r * 4 + (lane & 3)];
        acc += shfl_xor64(acc, 4); acc += shfl_xor64(acc, 8); acc += shfl_xor64(acc, 16); acc += shfl_xor64(acc, 32);
        if (lane < 4) red[3 + lane] = acc;
    }
    __syncthreads();
    if (tid < 8) {
        const u64 x = (tid < 7) ? red[tid] : p.mark;  // column 7 = "row present" marker (1, or the host's sequence number)
        a.partial[(size_t)g * 8 + tid] = x;
        if (a.host_partial) a.host_partial[(size_t)g * 8 + tid] = x;
    }
    // A node whose claims exceed its free capacity: in which block (line of H) does the ordered prefix cross it?  The
    // column is still in this workgroup's registers, so the answer costs no extra launch (k_cutblk's job on the row-sharded
    // path): claim columns to LDS in row order, one wave per pair of nodes, four rows per lane, ordered scan.
    if (!a.budget || cutmask == 0) return;  // block-uniform
    if (tid < 256 && cp >= 4) {
#pragma unroll
        for (int r = 0; r < kResRows; ++r) {
            colbuf[2 * cp - kResNodes][rg + r * kResRowGroups] = v[r][0];
            colbuf[2 * cp - kResNodes + 1][rg + r * kResRowGroups] = v[r][1];
        }
    }
    __syncthreads();
    if (tid < 256) {
        for (int q = tid >> 6; q < kResNodes; q += 4) {  // wave -> nodes q, q + 4
            if (!((cutmask >> q) & 1u)) continue;
            const u64 fre = cutfre[q];
            const u64 x0 = colbuf[q][lane * 4 + 0], x1 = colbuf[q][lane * 4 + 1], x2 = colbuf[q][lane * 4 + 2],
                      x3 = colbuf[q][lane * 4 + 3];
            const u64 s1 = x0 + x1, s2 = s1 + x2, s3 = s2 + x3;
            const u64 inc = wave_incl_scan(s3, lane);
            const u64 ex = inc - s3;
            int e = 4;  // first row of this lane whose inclusive prefix exceeds the free capacity
            if (ex + s3 > fre) e = 3;
            if (ex + s2 > fre) e = 2;
            if (ex + s1 > fre) e = 1;
            if (ex + x0 > fre) e = 0;
            const u64 mask = __ballot(e < 4);
            if (mask) {
                const int fl = __ffsll((long long)mask) - 1;
                if (lane == fl) {
                    const u64 cum = ex + (e == 0 ? 0ull : e == 1 ? x0 : e == 2 ? s1 : s2);
                    const u32 jq = g * kResNodes + q;
                    a.cutblk[jq] = (u32)(lane * 4 + e);
                    a.budget[jq] = fre - cum;
                    a.admpre[jq] = cum;
                    if (a.Tg) {
                        uint4* z = reinterpret_cast<uint4*>(a.Tg + (size_t)jq * kWaves);
#pragma unroll
                        for (int w = 0; w < kWaves / 2; ++w) z[w] = make_uint4(0, 0, 0, 0);
                    }
                    s_cb[q] = (u32)(lane * 4 + e);
                    s_bud[q] = fre - cum;
                    s_adm[q] = cum;
                }
            } else if (lane == 0) {
                s_cb[q] = kNoCut;  // (cannot happen: the column total exceeds the free capacity)
            }
        }
    }
    if (tid == 0) atomicAdd(&a.stats->n_cut, 1ull);  // device-side "some node has a cut" flag (the cut kernels' guard)
    __syncthreads();
    // What this group's cuts reject, per block: behind a node's cut block its whole claim load, in the cut block what the exact
    // search does not admit.  k_fill's round 0 needs the sum over the blocks BEFORE its own: the exclusive prefix over the
    // blocks goes to RP[b][g] (plain stores; 32 k atomics on sixteen lines were 30 us of this kernel).  The in-block
    // correction is known here only when this kernel searches (SEARCH); k_cut_find puts it into R[cut block] otherwise.
    auto store_rejected = [&]() {
        __shared__ u64 wtot[4];
        u64 r = 0;
        if (tid < 256) {
#pragma unroll
            for (int q = 0; q < kResNodes; ++q)
                if (((cutmask >> q) & 1u) && s_cb[q] != kNoCut) {
                    if ((u32)tid >= s_cb[q]) r += colbuf[q][tid];
                    if (SEARCH && (u32)tid == s_cb[q]) r -= s_in[q];
                }
        }
        const u64 inc = wave_incl_scan(r, lane);
        if (tid < 256 && lane == 63) wtot[tid >> 6] = inc;
        __syncthreads();
        if (tid < (int)G) {
            u64 ex = inc - r;
            for (int w = 0; w < (tid >> 6); ++w) ex += wtot[w];
            a.RP[(size_t)g * G + tid] = ex;  // row g: this workgroup's 2 KB, whole lines (a [block][group] layout made every
                                             // word a partial-line store from another XCD: 32 k of them cost this kernel 20 us)
        }
    };
    if (!SEARCH) {
        if (a.RP) store_rejected();
        return;
    }
    RIOGP_KT(p, 0, 2);
    // ---- exact cut rows of this group's nodes: waves 2q, 2q+1 take node q; each reads eight wave ranges of the cut block.
    //      The rows of the eight ranges are taken in "slots" of 64 (range-major, so slot order = index order); a batch of
    //      kSrchSlots slots is requested at once — one round trip for ~2 500 rows, the usual half block of a churn tick —
    //      and the second half's first batch is in flight while the first half searches.
    {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = wave >> 1, half = wave & 1;
        const bool mine = ((cutmask >> q) & 1u) && s_cb[q] != kNoCut;
        const u32 jq = g * kResNodes + q;
        const u32 cb = mine ? s_cb[q] : 0u;
        const u64 gw0 = (u64)cb * kWaves + (u64)half * 8;
        // the eight ranges: first position and number of packed rows (lane s < 8 fetches range s)
        u64 lo_l = 0;
        u32 cn_l = 0;
        if (mine && lane < 8) {
            const u64 gw = gw0 + lane;
            u64 ws = wave_row_lo(p, gw), we = wave_row_lo(p, gw + 1);
            if (we > p.n) we = p.n;
            if (ws > we) ws = we;
            const u64 c = p.wcnt[gw];
            lo_l = ws;
            cn_l = (u32)(c < we - ws ? c : we - ws);
        }
        u32 cnt_s[8], pre_s[9];  // rows of range s | slots before range s
        u64 lo_s[8];
        pre_s[0] = 0;
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
            cnt_s[s8] = (u32)__builtin_amdgcn_readlane((int)cn_l, s8);
            lo_s[s8] = shfl64(lo_l, s8);
            pre_s[s8 + 1] = pre_s[s8] + ((cnt_s[s8] + (u32)kTile - 1u) / (u32)kTile);
        }
        const u32 nslots = mine ? pre_s[8] : 0u;
        // a slot = one tile (256 rows, dwordx4 per lane and column: the ranges start on tile boundaries); 64-row slots of
        // single dwords were 4x the load instructions and cost this kernel 20 us
        uint4 ra[kSrchSlots], rl[kSrchSlots];  // affinity | load of the batch's rows (rows = slot base + 4 lane ..)
        u32 d_lo = 0, d_hi = 0, d_rem = 0;     // lane k: first position (two halves) and rows of slot t0 + k
        // descriptors of the batch starting at slot t0 (lane k -> slot t0 + k), then every row of the batch is requested
        auto fetch = [&](u32 t0) {
            const u32 t = t0 + (u32)lane;
            u32 sg = 0;
#pragma unroll
            for (int s8 = 1; s8 < 8; ++s8) sg += t >= pre_s[s8];
            u64 lo = lo_s[0];
            u32 cn = cnt_s[0], pr = pre_s[0];
#pragma unroll
            for (int s8 = 1; s8 < 8; ++s8) {
                const bool pick = sg == (u32)s8;
                lo = pick ? lo_s[s8] : lo;
                cn = pick ? cnt_s[s8] : cn;
                pr = pick ? pre_s[s8] : pr;
            }
            const u32 off = (t - pr) * (u32)kTile;
            const bool ok = t < nslots && off < cn;
            const u64 st = ok ? lo + off : lo_s[0];
            const u32 rem = ok ? (cn - off < (u32)kTile ? cn - off : (u32)kTile) : 0u;
            d_lo = (u32)st; d_hi = (u32)(st >> 32); d_rem = rem;
#pragma unroll
            for (int k = 0; k < kSrchSlots; ++k) {
                const u64 base = ((u64)(u32)__builtin_amdgcn_readlane((int)d_hi, k) << 32) | (u32)__builtin_amdgcn_readlane((int)d_lo, k);
                const u64 at = base + (u64)lane * 4;  // (a whole tile is always addressable: the columns are padded)
                ra[k] = *reinterpret_cast<const uint4*>(a.pk_aff + at);
                rl[k] = *reinterpret_cast<const uint4*>(a.pk_load + at);
            }
        };
        // ordered walk over the batch in registers: claim load of node jq, and the first row whose inclusive prefix exceeds
        // the budget.  Rows of a packed range are all pending, so "claimant of jq" = affinity == jq (and inside the slot).
        u64 acc = 0, adm_in = 0;
        u32 row = kNoCut;
        bool found = false;
        auto consume = [&](u64 bud) {
#pragma unroll
            for (int k = 0; k < kSrchSlots; ++k) {
                const u32 rem_k = (u32)__builtin_amdgcn_readlane((int)d_rem, k);
                const u32 e0 = (u32)lane * 4u;
                const bool k0 = e0 + 0 < rem_k && ra[k].x == jq, k1 = e0 + 1 < rem_k && ra[k].y == jq;
                const bool k2 = e0 + 2 < rem_k && ra[k].z == jq, k3 = e0 + 3 < rem_k && ra[k].w == jq;
                if (!__ballot(k0 | k1 | k2 | k3)) continue;  // (wave-uniform: a node has a handful of claimants per block)
                const u64 l0 = k0 ? (u64)rl[k].x : 0ull, l1 = k1 ? (u64)rl[k].y : 0ull;
                const u64 l2 = k2 ? (u64)rl[k].z : 0ull, l3 = k3 ? (u64)rl[k].w : 0ull;
                const u64 s1 = l0 + l1, s2 = s1 + l2, s3 = s2 + l3;
                const u64 inc = wave_incl_scan(s3, lane);
                if (!found) {
                    const u64 ex = acc + inc - s3;  // claim load of jq before this lane's rows
                    int e = 4;
                    if (k3 && ex + s3 > bud) e = 3;
                    if (k2 && ex + s2 > bud) e = 2;
                    if (k1 && ex + s1 > bud) e = 1;
                    if (k0 && ex + l0 > bud) e = 0;
                    const u64 over = __ballot(e < 4);
                    if (over) {
                        const int fl = __ffsll((long long)over) - 1;
                        const int ef = __builtin_amdgcn_readlane(e, fl);
                        const u64 before = ef == 0 ? 0ull : (ef == 1 ? l0 : (ef == 2 ? s1 : s2));
                        const u64 base = ((u64)(u32)__builtin_amdgcn_readlane((int)d_hi, k) << 32) | (u32)__builtin_amdgcn_readlane((int)d_lo, k);
                        row = (u32)(base + (u64)fl * 4 + (u64)ef);
                        adm_in = shfl64(ex + before, fl);
                        found = true;
                    }
                }
                acc += shfl64(inc, 63);
            }
        };
        RIOGP_KT(p, 0, 3);
        if (mine && nslots) fetch(0);
        RIOGP_KT(p, 0, 4);
        if (mine && half == 0) {
            for (u32 t0 = 0; t0 < nslots; t0 += kSrchSlots) {
                if (t0) fetch(t0);
                consume(s_bud[q]);
            }
            if (lane == 0) { s_half[q] = acc; s_row[q] = row; s_in[q] = adm_in; }
        }
        RIOGP_KT(p, 0, 5);
        __syncthreads();
        if (mine && half == 1 && s_row[q] == kNoCut) {
            acc = s_half[q];  // the first half's claimants come first, and all of them were admitted
            for (u32 t0 = 0; t0 < nslots; t0 += kSrchSlots) {
                if (t0) fetch(t0);
                consume(s_bud[q]);
            }
            if (lane == 0) { s_row[q] = row; s_in[q] = adm_in; }
        }
        __syncthreads();
        RIOGP_KT(p, 0, 6);
        if (mine && half == 0 && lane == 0) {
            const u64 in_blk = s_row[q] == kNoCut ? 0ull : s_in[q];  // (kNoCut cannot happen: the block prefix crosses the budget here)
            a.cutidx[jq] = s_row[q];
            a.used_cur[jq] = s_used[q] + s_adm[q] + in_blk;
            s_in[q] = in_blk;
        }
        __syncthreads();
        if (a.RP) store_rejected();
        RIOGP_KT(p, 0, 7);
    }
}

// ------------------------------------------------------------------------------------------------
// K2c k_cutblk — fix-up path only: for nodes whose claims exceed their free capacity, the block in
//     which the index-ordered claim prefix first exceeds free ("cut block"), the budget left at its
//     start and the load admitted before it.  16 nodes x 16 row-groups per workgroup.
// ------------------------------------------------------------------------------------------------
constexpr int kCbNodes = 16, kCbGroups = 16, kCbRows = kMaxBlocks / kCbGroups;  // 16 rows/thread

__global__ __launch_bounds__(256) void k_cutblk(const u64* __restrict__ H, Plan p, const u64* __restrict__ cap,
                                                const u32* __restrict__ alive_bits,
                                                const u64* __restrict__ used_kept,
                                                const u64* __restrict__ claim_tot, u32* __restrict__ cutblk,
                                                u64* __restrict__ budget, u64* __restrict__ admpre,
                                                DevStats* __restrict__ stats) {
    __shared__ u64 sc[kCbGroups][kCbNodes];
    const int tid = threadIdx.x, nd = tid & (kCbNodes - 1), grp = tid >> 4;
    const u32 m = p.m, G = p.G;
    const u32 j = blockIdx.x * kCbNodes + nd;
    const bool valid = j < m;
    const u32 rows = (G + kCbGroups - 1) / kCbGroups;  // <= kCbRows
    u64 fre = 0, ctot = 0;
    if (valid) {
        const u64 c = cap[j], used = used_kept[j];
        fre = ((p.sa || bit_of(alive_bits, j)) && c > used) ? c - used : 0;
        ctot = claim_tot[j];
    }
    const bool has_cut = valid && ctot > fre;
    u64 cc[kCbRows];
    u64 tc = 0;
#pragma unroll
    for (int r = 0; r < kCbRows; ++r) {
        const u32 row = grp * rows + r;
        cc[r] = (has_cut && (u32)r < rows && row < G) ? H[h_line(j >> 3, row, G) + 8 + (j & 7)] : 0;
        tc += cc[r];
    }
    sc[grp][nd] = tc;
    // stats->n_cut (zeroed by k_scan) > 0 iff some node has a cut: k_cut_fused returns at once otherwise, which is what
    // makes it cheap to enqueue the fix-up speculatively behind a solve whose verdict the host has not read yet
    const int any_cut = __syncthreads_or(has_cut && grp == 0);
    if (tid == 0 && any_cut) atomicAdd(&stats->n_cut, 1ull);
    if (!has_cut) return;
    u64 pre = 0;
    for (int g = 0; g < grp; ++g) pre += sc[g][nd];
    if (pre <= fre && pre + tc > fre) {  // the cut is inside this thread's row group
        u64 cum = pre;
#pragma unroll
        for (int r = 0; r < kCbRows; ++r) {
            if (cum <= fre && cum + cc[r] > fre) {
                cutblk[j] = grp * rows + r;
                budget[j] = fre - cum;
                admpre[j] = cum;
            }
            cum += cc[r];
        }
    }
}

// rejected claimants of one workgroup: into its row of the fix-up counters, or atomically into DevStats
__device__ __forceinline__ void fx_add_rejected(const FxRows& fx, DevStats* stats, u64 cnt, u64 load) {
    if (fx.dev) {
        u64* r = fx.dev + (size_t)blockIdx.x * 8;
        r[0] += cnt;
        r[1] += load;
    } else if (cnt) {
        atomicAdd(&stats->rejected, cnt);
        atomicAdd(&stats->load_rejected, load);
    }
}

// ------------------------------------------------------------------------------------------------
// K3  the exact cut search for ONE block b of rows (cut_search_block) and the kernel that spreads the searches over the chip
//     (k_cut_find).  The exact cut row of node j is a fact about ONE block's rows (block cutblk[j], located by k_resolve):
//       P0  which nodes have their cut in b (a slice of them when the block's cuts are spread over several work items);
//       P1  per group of K such nodes, level by level: T[slot][piece of the node's current range] claim load of the
//           block's rows (LDS atomics), then an ordered walk over each node's T row shrinks its range to the piece
//           that holds the cut; another level only while row-by-row searches would cost more than one more pass;
//       P2  one wave per node: the exact row inside the remaining range (whole tiles, dwordx4);
//           cutidx[j] = that row, used_cur[j] = kept + admitted, R[b] -= what the block admits of the node.
// ------------------------------------------------------------------------------------------------
constexpr u32 kSlotNone = 0xFFFFu;
constexpr int kCutMinSubs = 16;  // smallest fan-out of a refinement level

// fixed LDS of the block search in front of the T region (keep in step with cut_find_lds)
__host__ __device__ __forceinline__ size_t cut_fused_fixed(u32 m, u32 mwords) {
    const u32 mr = (m + 7) & ~7u;
    return kSmall + (size_t)mr * 8 + (size_t)((mwords + 3) & ~3u) * sizeof(u32) + 2 * kWaves * sizeof(u64);
}

// The search half of the cut fix-up for ONE block b of rows: which nodes have their cut in b (all of them, or the slice
// j % sel_mod == sel_rem of them when the block's cuts are spread over several workgroups), and for each the exact row —
// P0..P2 of the description above.  Every thread of the workgroup calls it; on return thr[] (LDS) holds the reject
// threshold of every node as seen from block b, and cutidx[] / used_cur[] (global) are final for the nodes searched.
template <bool VIRT>
__device__ __forceinline__ void cut_search_block(unsigned char* smem, const u32 b, const u32 sel_mod, const u32 sel_rem,
                                                 const bool write_forced, const u32* __restrict__ cur,
                                                 const u32* __restrict__ load, const u32* __restrict__ aff,
                                                 const u32* __restrict__ alive_bits, const Plan& p,
                                                 const u32* __restrict__ cutblk, const u64* __restrict__ budget,
                                                 const u64* __restrict__ admpre, const u64* __restrict__ used_kept,
                                                 const u32* __restrict__ forced_bits, u32* __restrict__ cutidx,
                                                 u64* __restrict__ used_cur, const u32 tcap, u64* __restrict__ R) {
    const u32 m = p.m, mr = (m + 7) & ~7u;
    const bool sa_ = p.sa != 0;  // claims do not need a live node (Plan::sa)
    u32& nlocal = *reinterpret_cast<u32*>(smem);
    u64* red = reinterpret_cast<u64*>(smem + 16);                            // [2]
    u32* thr = reinterpret_cast<u32*>(smem + kSmall);                        // [mr] reject threshold by node
    unsigned short* slot = reinterpret_cast<unsigned short*>(thr + mr);      // [mr] local slot of a node or kSlotNone
    unsigned short* node_of = slot + mr;                                     // [mr] node of a local slot
    u32* alv = reinterpret_cast<u32*>(node_of + mr);                         // [(mwords+3)&~3]
    u64* rlo = reinterpret_cast<u64*>(alv + ((p.mwords + 3) & ~3u));         // [16] first row position of wave range w
    u64* whi = rlo + kWaves;                                                 // [16] end of its LIVE rows (n / packed count)
    u64* T = whi + kWaves;                                                   // [tcap]: T[K][S] from the front,
                                                                             //         budget of slot s at T[tcap-1-s]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: row ranges in SGPRs
    if (tid == 0) { nlocal = 0; red[0] = 0; red[1] = 0; red[2] = 0; red[3] = 0; }
    // one round trip for the liveness words, the packed row count of this wave and the thread's first cutblk[] word
    const u64 gw = (u64)b * kWaves + wave;
    const u32 cb0 = cutblk[(u32)tid < m ? (u32)tid : m - 1];
    const u32 ak = (u32)tid < p.mwords ? (u32)tid : p.mwords - 1;  // mwords <= 256 < kBlock
    const u32 aw = alive_bits[ak];
    const u32 wc = *(p.wcnt ? p.wcnt + gw : cutblk);
    const u64 bstart = block_row_lo(p, b);
    u64 wstart = wave_row_lo(p, gw), wend = wave_row_lo(p, gw + 1);
    if (wend > p.n) wend = p.n;
    if (wstart > wend) wstart = wend;
    if (p.wcnt && wstart + wc < wend) wend = wstart + wc;
    alv[ak] = aw;  // threads past mwords rewrite the last word with its own value
    if (lane == 0) { rlo[wave] = wave_row_lo(p, gw); whi[wave] = wend; }
    __syncthreads();
    for (u32 j = tid; j < m; j += kBlock) {
        const u32 cbq = j == (u32)tid ? cb0 : cutblk[j];  // (m > 1 024: the later words are fetched here)
        u32 t = kNoCut;
        u32 s = kSlotNone;
        if (forced_bits && bit_of(forced_bits, j)) {  // row-sharded solve: the prefix overflowed on a lower rank
            t = 0;
            if (write_forced && b == 0) { cutidx[j] = 0; used_cur[j] = used_kept[j]; }
        } else if (cbq < b) {
            t = 0;
        } else if (cbq == b && (sel_mod <= 1 || j % sel_mod == sel_rem)) {  // this work item's slice of the block's cuts
            s = atomicAdd(&nlocal, 1u);
            node_of[s] = (unsigned short)j;
        }
        thr[j] = t;
        slot[j] = (unsigned short)s;
    }
    __syncthreads();
    const u32 nloc = nlocal;
    // Where is each local node's cut row?  Refinement by S-way histograms over the workgroup's rows, level by level:
    // level 0 splits the block's tiles into S pieces, a pass over the rows adds every claimant's load to T[node][piece],
    // an ordered walk over the node's T row finds the piece that holds the cut and shrinks the node's range to it;
    // the next level splits THAT range, and so on.  A level costs one pass over the block's rows; it is taken while
    // letting every node scan its remaining range row by row would cost more (cuts are not spread evenly over blocks:
    // a nearly full cluster cuts every node within its first claimants, i.e. all in block 0 — and on a 1.5e9-row table
    // one workgroup owning 964 cuts with 16 pieces of 1 400 tiles each spent 38 ms re-reading its rows per node).
    // S is chosen so that all local nodes share the passes when they fit (K nodes per group otherwise).
    // LDS: T rows [K][Sp] from the front of the T region, node state (budget left, load admitted before the range,
    // first tile of the range) in 3 x [K] words at its end.
    const u32 F = p.sub / kTile;
    const u32 BT = p.subs * F;                       // tiles per block (plan bound, the same for every block)
    u32 S = BT < 255u ? BT : 255u;                   // fan-out per level
    if (nloc) {
        const u32 per = tcap / nloc;                 // words per node if every local node is in ONE group
        const u32 sfit = per > 4 ? per - 4 : 0;      // T words of them (3 state words, 1 for the odd stride)
        if (sfit < S) S = sfit < (u32)kCutMinSubs ? (u32)kCutMinSubs : sfit;
        if (S > BT) S = BT;
    }
    if (S < 1) S = 1;
    const u32 Sp = (S | 1u) < 3u ? 3u : (S | 1u);    // T row stride: odd (lane-per-node walks are bank-conflict free),
                                                     // >= 3 (the T region is reused for two words per node below)
    u32 K = tcap / (Sp + 3);
    if (K < 1) K = 1;
    u64 bend = block_row_lo(p, b + 1);
    if (bend > p.n) bend = p.n;
    // tiles of this block that hold live rows (packed fix-up: a fraction of BT), for the "is another level worth a pass" test
    u32 live_tiles;
    {
        const u32 mine = lane < kWaves ? (u32)((whi[lane & (kWaves - 1)] > rlo[lane & (kWaves - 1)]
                                                     ? whi[lane & (kWaves - 1)] - rlo[lane & (kWaves - 1)] + kTile - 1 : 0) / kTile) : 0u;
        live_tiles = wave_sum32(mine);
        if (live_tiles < 1) live_tiles = 1;
    }

    for (u32 g0 = 0; g0 < nloc; g0 += K) {
        const u32 kn = nloc - g0 < K ? nloc - g0 : K;
        u64* st_bud = T + tcap - 3 * (size_t)K;      // [K] budget left inside the node's current range
        u64* st_pre = st_bud + K;                    // [K] claim load admitted before the range (inside this block)
        u64* st_rs = st_pre + K;                     // [K] first tile of the range (tile offset inside the block)
        for (u32 ls = tid; ls < kn; ls += kBlock) {
            st_bud[ls] = budget[node_of[g0 + ls]];
            st_pre[ls] = 0;
            st_rs[ls] = 0;
        }
        u32 rlen = BT;                               // tiles in every node's current range (uniform per level)
        for (u32 level = 0;; ++level) {
            const u32 ptiles = (rlen + S - 1) / S;   // tiles per piece at this level
            const u32 np = (rlen + ptiles - 1) / ptiles;  // pieces actually used (<= S)
            for (u32 k = tid; k < kn * Sp; k += kBlock) T[k] = 0;
            __syncthreads();
            if (kn == 1) {
                // ONE local node (a few hot nodes cut, each in a block of its own — skewed affinity): its claimants of a tile all
                // add to the same piece, so the tile is summed in registers (one DPP reduction) and added once; no slot table,
                // no per-row atomics, and tiles outside the node's current range are not even read.  The general pass below
                // spends ~240 instructions per tile and wave on this, 12 us per pass on the one CU that owns the item.
                const u32 nd0 = (u32)node_of[g0];
                const bool live0 = p.sa || bit_of(alv, nd0);
                const u32 rs0 = (u32)st_rs[0];
                for (u64 it = wstart; it < wend; it += kTile) {
                    const u32 tb = (u32)((it - bstart) / kTile);
                    u32 t = tb / ptiles;
                    if (level != 0) {
                        const u32 off = tb - rs0;  // below the range wraps to a huge value
                        if (off >= rlen) continue;
                        t = off / ptiles;
                    }
                    const u64 i0 = it + (u64)lane * 4;
                    const uint4 cv = *reinterpret_cast<const uint4*>(cur + i0);
                    const uint4 av = *reinterpret_cast<const uint4*>(aff + i0);
                    const uint4 lv = *reinterpret_cast<const uint4*>(load + i0);
                    u64 sum = 0;
#define RIOGP_ROW(C, A, L, E)                                                                              \
                    {                                                                                      \
                        const bool cin = C < m;                                                            \
                        const bool kept = VIRT ? cin : (cin & bit_of(alv, cin ? C : 0u));                  \
                        const bool hit = (i0 + E < wend) & !kept & !(VIRT && C == kSkipMark) & (A == nd0) & live0; \
                        sum += hit ? (u64)L : 0ull;                                                        \
                    }
                    RIOGP_ROW(cv.x, av.x, lv.x, 0)
                    RIOGP_ROW(cv.y, av.y, lv.y, 1)
                    RIOGP_ROW(cv.z, av.z, lv.z, 2)
                    RIOGP_ROW(cv.w, av.w, lv.w, 3)
#undef RIOGP_ROW
                    sum = wave_sum(sum);
                    if (lane == 0 && sum) atomicAdd(&T[t], sum);
                }
            } else {
                // pass: claim load per (local node, piece of its range); a tile (256 rows) lies inside ONE piece.  The next tile is
                // in flight while this one is processed: ONE workgroup streams the block (470 KB of a 10 M-row table), and with a
                // single tile per wave outstanding it ran at ~25 GB/s — 12 us per pass while most of the chip idles.
                uint4 cvn = make_uint4(0, 0, 0, 0), avn = cvn, lvn = cvn;
                if (wstart < wend) {
                    const u64 f0 = wstart + (u64)lane * 4;
                    cvn = *reinterpret_cast<const uint4*>(cur + f0);
                    avn = *reinterpret_cast<const uint4*>(aff + f0);
                    lvn = *reinterpret_cast<const uint4*>(load + f0);
                }
                for (u64 it = wstart; it < wend; it += kTile) {
                    const u64 i0 = it + (u64)lane * 4;
                    const uint4 cv = cvn, av = avn, lv = lvn;
                    {
                        const u64 pn = (it + kTile < wend ? it + kTile : it) + (u64)lane * 4;  // (the last tile re-reads itself)
                        cvn = *reinterpret_cast<const uint4*>(cur + pn);
                        avn = *reinterpret_cast<const uint4*>(aff + pn);
                        lvn = *reinterpret_cast<const uint4*>(load + pn);
                    }
                    const u32 tb = (u32)((it - bstart) / kTile);
                    const u32 t_lvl0 = tb / ptiles;
    #define RIOGP_ROW(C, A, L, E)                                                                              \
                    {                                                                                          \
                        /* branch-free class test (as P3): claimant = in range, not kept, not a duplicate, affinity live */ \
                        const bool cin = C < m, ain = A < m;                                                   \
                        const u32 cx = cin ? C : 0u, ax = ain ? A : 0u;                                        \
                        const bool kept = VIRT ? cin : (cin & bit_of(alv, cx));                                \
                        const u32 ls = (u32)slot[ax] - g0;  /* kSlotNone - g0 >= kn always */                  \
                        bool hit = (i0 + E < wend) & !kept & !(VIRT && C == kSkipMark) & ain & (sa_ | bit_of(alv, ax)) & (ls < kn); \
                        u32 t = 0;                                                                             \
                        if (level == 0) {                                                                      \
                            t = t_lvl0;  /* every range is the whole block: the piece is a property of the tile */ \
                        } else if (hit) {                                                                      \
                            const u32 off = tb - (u32)st_rs[ls];  /* below the range wraps to a huge value */  \
                            hit = off < rlen;                                                                  \
                            t = off / ptiles;                                                                  \
                        }                                                                                      \
                        const u64 todo = __ballot(hit);                                                        \
                        if (__popcll(todo) >= 16) { /* a HOT node (>= 16 rows of this wave-element): one LDS atomic, not 16+ */ \
                            const int ld = __ffsll((long long)todo) - 1;                                       \
                            const u32 s0 = (u32)__shfl((int)(ls * Sp + t), ld, 64);                            \
                            const bool same = hit && ls * Sp + t == s0;                                        \
                            if (__popcll(__ballot(same)) >= 16) {                                              \
                                const u64 sum = wave_sum(same ? (u64)L : 0ull);                                \
                                if (lane == ld) atomicAdd(&T[s0], sum);                                        \
                                hit = hit && !same;                                                            \
                            }                                                                                  \
                        }                                                                                      \
                        if (hit) atomicAdd(&T[ls * Sp + t], (u64)L);                                           \
                    }
                    RIOGP_ROW(cv.x, av.x, lv.x, 0)
                    RIOGP_ROW(cv.y, av.y, lv.y, 1)
                    RIOGP_ROW(cv.z, av.z, lv.z, 2)
                    RIOGP_ROW(cv.w, av.w, lv.w, 3)
    #undef RIOGP_ROW
                }
            }
            __syncthreads();
            // ordered walk over each node's T row: the piece that holds the cut becomes the node's range.  Two forms,
            // picked by estimated instruction count: one LANE per node (many nodes, short rows: row stride odd, so no
            // bank conflicts) or one WAVE per node with a DPP scan (few nodes, long rows).  LDS only.
            const u32 cost_lane = ((np + 7) / 8) * 100u;
            const u32 cost_wave = ((kn + kWaves - 1) / kWaves) * (((np + 63) / 64) * 80u + 120u);
            if (cost_lane <= cost_wave) {
                for (u32 ls = tid; ls < kn; ls += kBlock) {
                    const u64* Tj = T + (size_t)ls * Sp;
                    const u64 bud = st_bud[ls];
                    u64 acc = 0;
                    u32 tstar = np;
                    for (u32 t0 = 0; t0 < np && tstar == np; t0 += 8) {  // 8 independent LDS reads, then a branch-free walk
                        u64 v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = t0 + q < np ? Tj[t0 + q] : 0ull;
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const u64 nv = acc + v[q];
                            const bool open = tstar == np;
                            const bool over = nv > bud;  // padding words are 0: never over
                            tstar = (open && over) ? t0 + q : tstar;
                            acc = (open && !over) ? nv : acc;
                        }
                    }
                    if (tstar == np) { tstar = 0; acc = 0; }  // no overflow anywhere (k_cutblk says there is one): range start
                    st_bud[ls] = bud - acc;
                    st_pre[ls] += acc;
                    st_rs[ls] += (u64)tstar * ptiles;
                }
            } else {
                for (u32 ls = wave; ls < kn; ls += kWaves) {
                    const u64* Tj = T + (size_t)ls * Sp;
                    const u64 bud = st_bud[ls];
                    u64 acc = 0, pre = 0;
                    u32 tstar = 0;
                    bool found = false;
                    for (u32 g = 0; g < np && !found; g += 64) {
                        const u32 t = g + lane;
                        const u64 v = t < np ? Tj[t] : 0;
                        const u64 inc = wave_incl_scan(v, lane);
                        const u64 mask = __ballot(acc + inc > bud);
                        if (mask) {
                            const int fl = __ffsll((long long)mask) - 1;
                            tstar = g + fl;
                            pre = acc + shfl64(inc - v, fl);
                            found = true;
                        } else {
                            acc += shfl64(inc, 63);
                        }
                    }
                    if (lane == 0) {
                        st_bud[ls] = bud - pre;
                        st_pre[ls] += pre;
                        st_rs[ls] += (u64)tstar * ptiles;
                    }
                }
            }
            __syncthreads();
            rlen = ptiles;
            // another level?  In tile-steps shared by 16 waves: row-by-row searches of the remaining ranges cost about
            // kn x rlen x (live share) ordered scans; one more level costs a pass over the live tiles (a third of a scan
            // each) plus a fixed part (zeroing, barriers, walks ~ 100 scans).  Taken only when it clearly pays (2x).
            if (rlen <= 1 || (u64)kn * rlen * live_tiles <= (u64)BT * (2ull * (live_tiles / 3 + 100))) break;
        }
        // the rows each node's search covers: from the first tile of its range that holds live rows (packed fix-up: most
        // positions of a wave range are dead) to the end of the range — one lane per node, LDS only, into the T region
        // (free now): T[ls] = first row, T[kn + ls] = end.  Done here so that the searches below start with their loads.
        for (u32 ls = tid; ls < kn; ls += kBlock) {
            u64 rs = bstart + st_rs[ls] * kTile;
            u64 re = rs + (u64)rlen * kTile;
            if (re > bend) re = bend;
            if (rs > re) rs = re;
            u64 st = rs;
            while (st + kTile < re) {
                int w = 0;
                for (int q = 1; q < kWaves; ++q) w += rlo[q] <= st;
                if (whi[w] > st) break;
                st += kTile;
            }
            T[ls] = st;
            T[kn + ls] = re;
        }
        __syncthreads();
        // P2b: one wave per node, three nodes per wave in flight — the exact row inside the range, tile by tile
        //      (dwordx4 columns, in-tile order = lane, element).  The tiles and node words of three nodes are requested
        //      before the first is searched: a workgroup that owns hundreds of cuts is bound by the round trips of its
        //      16 waves, so each trip has to carry several nodes.
        struct Job { u32 ls; uint4 c, a, l; u64 uk, ad; };  // only what is in flight; the rest is re-read from LDS
        // first live tile at or after t, below lim (LDS tables only; wave-uniform)
        auto next_live = [&](u64 t, u64 lim) -> u64 {
            while (t < lim) {
                const int w = __popcll(__ballot(lane < kWaves && rlo[lane & (kWaves - 1)] <= t)) - 1;
                if (whi[w < 0 ? 0 : w] > t) break;
                t += kTile;
            }
            return t;
        };
        auto fetch = [&](Job& q, u32 ls) {
            const u64 start = T[ls];
            const u32 j = node_of[g0 + ls];
            q.ls = ls;
            q.c = *reinterpret_cast<const uint4*>(cur + start + (u64)lane * 4);
            q.a = *reinterpret_cast<const uint4*>(aff + start + (u64)lane * 4);
            q.l = *reinterpret_cast<const uint4*>(load + start + (u64)lane * 4);
            q.uk = used_kept[j];
            q.ad = admpre[j];
        };
        auto run = [&](const Job& q) {
            const u64 q_start = T[q.ls], q_end = T[kn + q.ls];
            const u64 bud2 = st_bud[q.ls], q_pre_sub = st_pre[q.ls];
            const u32 j = node_of[g0 + q.ls];
            u64 acc2 = 0, cut_row = kNoCut, adm_in = 0;
            // one tile of the search: rows of node j's claimants in (lane, element) order against bud2
            auto scan1 = [&](const uint4& xc, const uint4& xa, const uint4& xl, u64 t0) -> bool {
                // the wave range this tile lies in (ranges are whole tiles) and where its live rows end
                const int w = __popcll(__ballot(lane < kWaves && rlo[lane & (kWaves - 1)] <= t0)) - 1;
                const u64 lim = whi[w < 0 ? 0 : w];
                const u64 i0 = t0 + (u64)lane * 4;
                u64 l0, l1, l2, l3;
                bool k0, k1, k2, k3;
                // branch-free: j is a live node, so "claimant of j" = aff == j and the row is pending (not kept, not a
                // duplicate request); short-circuit && here costs an exec-mask branch per term and element
#define RIOGP_EL(C, A, L, E, KO, LO)                                                        \
                {                                                                            \
                    const u32 cx = C < m ? C : 0u;                                           \
                    const bool keptx = VIRT ? (C < m) : ((C < m) & bit_of(alv, cx));         \
                    KO = (i0 + E < lim) & (A == j) & !keptx & !(VIRT && C == kSkipMark);     \
                    LO = KO ? (u64)L : 0ull;                                                 \
                }
                RIOGP_EL(xc.x, xa.x, xl.x, 0, k0, l0)
                RIOGP_EL(xc.y, xa.y, xl.y, 1, k1, l1)
                RIOGP_EL(xc.z, xa.z, xl.z, 2, k2, l2)
                RIOGP_EL(xc.w, xa.w, xl.w, 3, k3, l3)
#undef RIOGP_EL
                const u64 s1 = l0 + l1, s2 = s1 + l2, s3 = s2 + l3;
                const u64 inc = wave_incl_scan(s3, lane);
                const u64 ex = acc2 + inc - s3;  // load of this node's claimants before this lane's rows
                int e = 4;
                if (k3 && ex + s3 > bud2) e = 3;
                if (k2 && ex + s2 > bud2) e = 2;
                if (k1 && ex + s1 > bud2) e = 1;
                if (k0 && ex + l0 > bud2) e = 0;
                const u64 mask = __ballot(e < 4);
                if (mask) {
                    const int fl = __ffsll((long long)mask) - 1;
                    const int ef = __builtin_amdgcn_readlane(e, fl);
                    const u64 before = ef == 0 ? 0ull : (ef == 1 ? l0 : (ef == 2 ? s1 : s2));
                    cut_row = t0 + (u64)fl * 4 + (u64)ef;
                    adm_in = shfl64(ex + before, fl);
                    return true;
                }
                acc2 += shfl64(inc, 63);
                return false;
            };
            // the first tile is the one this job prefetched; deeper tiles of the range are rare and short (the refinement
            // levels above keep kn x range small), so they are read one at a time, tiles without live rows skipped
            bool found = scan1(q.c, q.a, q.l, q_start);
            for (u64 t0 = next_live(q_start + kTile, q_end); t0 < q_end && !found; t0 = next_live(t0 + kTile, q_end)) {
                const uint4 xc = *reinterpret_cast<const uint4*>(cur + t0 + (u64)lane * 4);
                const uint4 xa = *reinterpret_cast<const uint4*>(aff + t0 + (u64)lane * 4);
                const uint4 xl = *reinterpret_cast<const uint4*>(load + t0 + (u64)lane * 4);
                found = scan1(xc, xa, xl, t0);
            }
            if (lane == 0) {
                thr[j] = (u32)cut_row;
                cutidx[j] = (u32)cut_row;
                used_cur[j] = q.uk + q.ad + q_pre_sub + adm_in;
                // k_resolve counted the node's whole claim load of this block as rejected (RP, k_fill's ordered spill prefix):
                // what the block admits of it comes off again — summed over the item in LDS, one global atomic per item
                if (R && q_pre_sub + adm_in) atomicAdd(&red[1], q_pre_sub + adm_in);
            }
        };
        constexpr int kInFlight = 3;
        for (u32 base = wave; base < kn; base += kInFlight * kWaves) {
            Job q[kInFlight];
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) {  // clamped past the last node: the requests stay unconditional
                const u32 ls = base + (u32)u * kWaves;
                fetch(q[u], ls < kn ? ls : base);
            }
#pragma unroll
            for (int u = 0; u < kInFlight; ++u)
                if (base + (u32)u * kWaves < kn) run(q[u]);
        }
        __syncthreads();
    }
    if (R && tid == 0 && red[1]) atomicAdd(&R[b], (u64)0 - red[1]);  // what this item's cuts admit inside block b
}

// ------------------------------------------------------------------------------------------------
// K3f k_cut_find — the exact cut search, spread over the chip (whole-table solves; the packed fix-up searches inside
//     k_resolve).  If the workgroup that owns block b of rows also owned every cut that falls into b, a nearly full cluster —
//     most nodes reject within their first claimants — would have one workgroup search 200-400 nodes (31 us measured)
//     while ~250 CUs idle.  The search for node j
//     needs only block cutblk[j]'s rows, so it is a work ITEM (block b, slice s of the nodes cut in b): every
//     workgroup derives the same item list from cutblk[] (a histogram over blocks, ceil(count / 16) slices per block,
//     at most 64; slice = j mod slices), takes the items blockIdx.x, +gridDim.x, ... and runs the block search on each.
//     The re-marking pass that needs EVERY node's result is the next launch (k_fill).
// ------------------------------------------------------------------------------------------------
constexpr u32 kCutPerItem = 16;    // cut nodes per work item (one per wave of the workgroup that searches them)
constexpr u32 kCutMaxSlices = 64;  // slices of one block's cuts
constexpr u32 kCutFindGrid = 256;  // one workgroup per CU (the search needs ~150 KiB of LDS)

template <bool VIRT>
__global__ __launch_bounds__(kBlock) void k_cut_find(const u32* __restrict__ cur, const u32* __restrict__ load,
                                                     const u32* __restrict__ aff, const u32* __restrict__ alive_bits,
                                                     Plan p, const u32* __restrict__ cutblk,
                                                     const u64* __restrict__ budget, const u64* __restrict__ admpre,
                                                     const u64* __restrict__ used_kept,
                                                     const u32* __restrict__ forced_bits, u32* __restrict__ cutidx,
                                                     u64* __restrict__ used_cur, DevStats* __restrict__ stats, u32 tcap,
                                                     u64* __restrict__ R) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ u32 cnt[kMaxBlocks];                 // nodes whose cut falls into block b
    __shared__ unsigned short istart[kMaxBlocks + 1];  // first item of block b; [kMaxBlocks] = number of items
    __shared__ unsigned char nsl[kMaxBlocks];       // slices (= items) of block b
    const int tid = threadIdx.x, lane = tid & 63;
    const u32 m = p.m, G = p.G;
    const u64 ncut = stats->n_cut;  // one round trip for the verdict and the thread's first cutblk[] word
    const u32 cb0 = cutblk[(u32)tid < m ? (u32)tid : m - 1];
    if (ncut == 0 && forced_bits == nullptr) return;  // speculative launch behind a solve that had no cut
    if (forced_bits && blockIdx.x == 0)  // row-sharded solve: the prefix overflowed on a lower rank — nothing is admitted here
        for (u32 j = tid; j < m; j += kBlock)
            if (bit_of(forced_bits, j)) { cutidx[j] = 0; used_cur[j] = used_kept[j]; }
    for (u32 k = tid; k < kMaxBlocks; k += kBlock) cnt[k] = 0;
    __syncthreads();
    for (u32 j = tid; j < m; j += kBlock) {
        const u32 cb = j == (u32)tid ? cb0 : cutblk[j];
        if (cb < G && !(forced_bits && bit_of(forced_bits, j))) atomicAdd(&cnt[cb], 1u);
    }
    __syncthreads();
    if (tid < 64) {  // slices per block and their exclusive prefix: one wave, four blocks per lane
        u32 c[4], tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32 bq = (u32)tid * 4 + q;
            const u32 nn = bq < G ? cnt[bq] : 0;
            u32 sl = (nn + kCutPerItem - 1) / kCutPerItem;
            if (sl > kCutMaxSlices) sl = kCutMaxSlices;
            c[q] = sl;
            tot += sl;
        }
        const u64 inc = wave_incl_scan((u64)tot, lane);
        u32 ex = (u32)inc - tot;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            istart[tid * 4 + q] = (unsigned short)ex;
            nsl[tid * 4 + q] = (unsigned char)c[q];
            ex += c[q];
        }
        if (tid == 63) istart[kMaxBlocks] = (unsigned short)inc;
    }
    __syncthreads();
    const u32 items = istart[kMaxBlocks];
    for (u32 item = blockIdx.x; item < items; item += gridDim.x) {
        u32 lo = 0, hi = kMaxBlocks;  // the last block whose first item is <= item (blocks without cuts share their successor's start)
        while (hi - lo > 1) {
            const u32 mid = lo + ((hi - lo) >> 1);
            if (istart[mid] <= item) lo = mid; else hi = mid;
        }
        __syncthreads();  // the previous item's LDS tables are dead from here on
        cut_search_block<VIRT>(smem, lo, nsl[lo], item - istart[lo], false, cur, load, aff, alive_bits, p, cutblk, budget,
                               admpre, used_kept, forced_bits, cutidx, used_cur, tcap, R);
    }
}

// block-wide exclusive prefix (1 024 threads), optionally saturating
__device__ u64 block_excl_scan_1024(u64 v, bool saturating, u64* lds_part /*[16]*/, u64* total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 inc = saturating ? wave_incl_scan_sat(v, lane) : wave_incl_scan(v, lane);
    if (lane == 63) lds_part[wave] = inc;
    __syncthreads();
    u64 base = 0, tot = 0;
    for (int w = 0; w < kWaves; ++w) {
        const u64 x = lds_part[w];
        if (w < wave) base = saturating ? sat_add(base, x) : base + x;
        tot = saturating ? sat_add(tot, x) : tot + x;
    }
    __syncthreads();
    if (total) *total = tot;
    // exclusive = base + (inc - v); with saturation inc-v is wrong once saturated, recompute:
    u64 excl;
    if (saturating) {
        u64 prev = shfl_up64(inc, 1);
        if (lane == 0) prev = 0;
        excl = sat_add(base, prev);
    } else {
        excl = base + (inc - v);
    }
    return excl;
}

// ------------------------------------------------------------------------------------------------
// Water-fill order inside the workgroup.  The water-fill takes the nodes by CAPACITY CLASS descending, node index ascending
// inside a class; the class of a free capacity f > 0 is f rounded down to three significant bits, as an ordinal
// (4 * floor(log2 f) + the two bits below the leading one: 256 classes, monotone in f — DESIGN.md section 2 step 3,
// oracle/placement_oracle.c wf_class).  That order is a COUNTING problem: rank = nodes in higher classes + nodes of the same
// class with a lower index.  Every workgroup of k_fill computes it itself, in LDS, while its first rows are on their way
// from HBM (~100 vector instructions per node; the first version of this solver ranked by exact free capacity, which needs
// a comparison sort: ~1 400 instructions per node on one CU = 10 us at m = 1 024, or a launch of its own per round).
//   group = the 64 nodes one wave holds in one pass (node j = tid + 1024 q: group q * 16 + wave, in node order);
//   tab[group][class] = nodes of the class in the group -> exclusive prefix over the groups (thread = class);
//   rank(j) = nodes in higher classes + tab[group(j)][class] + position of j among the group's nodes of its class.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 wf_class(u64 f) {  // f > 0
    const u32 e = 63u - (u32)__clzll((long long)f);
    const u32 mant = e >= 2 ? (u32)(f >> (e - 2)) & 3u : (u32)(f << (2 - e)) & 3u;
    return e * 4u + mant;
}
__host__ __device__ __forceinline__ size_t rank_tab_bytes(u32 m) {  // u16 [groups][256]
    const u32 per = (m + kBlock - 1) / kBlock;
    return ((size_t)(per ? per : 1) * kWaves + 2) * 256 * sizeof(unsigned short);  // + class totals / offsets (+ the total)
}
// fr[q] = free capacity of node tid + 1024 q (0: not ranked), q < per <= 8.  On return rk[q] = rank of that node (only
// where fr[q] != 0) and the return value = number of ranked nodes.  tab: rank_tab_bytes(m) of LDS, ZEROED by the caller
// before its last barrier.  Every thread of the workgroup calls it (three barriers inside).
// tab layout: [class][pass q][wave] counts (the 16 waves of a pass = one 16-lane DPP row), then 256 class totals / offsets.
__device__ __forceinline__ u32 rank_by_class(unsigned short* tab, const u32 per, const u64 (&fr)[8], u32 (&rk)[8], const Plan& p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned short* ctot = tab + (size_t)per * kWaves * 256;  // [256] nodes of the class, then: nodes in higher classes
    u32 cls[8], pos[8];
    const u64 below = (1ull << lane) - 1ull;
#pragma unroll
    for (u32 q = 0; q < 8; ++q) {
        cls[q] = 0; pos[q] = 0;
        if (q < per) {  // (uniform)
            const bool in = fr[q] != 0;
            const u32 c = in ? wf_class(fr[q]) : 0x100u;
            u64 same = ~0ull;  // lanes of this group with the same class (nine ballots: eight class bits + "ranked at all")
#pragma unroll
            for (int b = 0; b < 9; ++b) {
                const u64 bal = __ballot((c >> b) & 1u);
                same &= ((c >> b) & 1u) ? bal : ~bal;
            }
            cls[q] = c;
            pos[q] = (u32)__popcll(same & below);
            if (in && pos[q] == 0) tab[((size_t)c * per + q) * kWaves + wave] = (unsigned short)__popcll(same);
        }
    }
    RIOGP_KT(p, 4, 0);
    __syncthreads();
    RIOGP_KT(p, 4, 1);
    {   // exclusive prefix over the groups (pass-major, then wave = node order), one 16-lane row per class and pass
        const u32 row = (u32)lane >> 4, l16 = (u32)lane & 15u;
        for (u32 c = (u32)wave * 4u + row; c < 256u; c += kWaves * 4u) {
            u32 carry = 0;
            for (u32 q = 0; q < per; ++q) {
                unsigned short* cell = tab + ((size_t)c * per + q) * kWaves + l16;
                const u32 v = *cell;
                u32 inc = v, t;
                t = dpp32<0x111>(inc); if (l16 >= 1) inc += t;
                t = dpp32<0x112>(inc); if (l16 >= 2) inc += t;
                t = dpp32<0x114>(inc); if (l16 >= 4) inc += t;
                t = dpp32<0x118>(inc); if (l16 >= 8) inc += t;
                *cell = (unsigned short)(carry + inc - v);
                carry += (u32)__shfl((int)inc, (lane | 15), 64);
            }
            if (l16 == 0) ctot[c] = (unsigned short)carry;
        }
    }
    RIOGP_KT(p, 4, 2);
    __syncthreads();
    u32 all = 0;
    if (wave == 0) {  // nodes in higher classes: suffix sum over the 256 class totals, four classes per lane
        const u32 c0 = ctot[4 * lane], c1 = ctot[4 * lane + 1], c2 = ctot[4 * lane + 2], c3 = ctot[4 * lane + 3];
        const u32 s4 = c0 + c1 + c2 + c3;
        const u32 inc = (u32)wave_incl_scan((u64)s4, lane);
        all = (u32)__builtin_amdgcn_readlane((int)inc, 63);
        const u32 above = all - inc;  // nodes in the classes of higher lanes
        ctot[4 * lane + 3] = (unsigned short)above;
        ctot[4 * lane + 2] = (unsigned short)(above + c3);
        ctot[4 * lane + 1] = (unsigned short)(above + c3 + c2);
        ctot[4 * lane] = (unsigned short)(above + c3 + c2 + c1);
        if (lane == 0) ctot[256] = (unsigned short)all;
    }
    __syncthreads();
    RIOGP_KT(p, 4, 3);
    all = ctot[256];
#pragma unroll
    for (u32 q = 0; q < 8; ++q) {
        rk[q] = 0;
        if (q < per && cls[q] < 0x100u)
            rk[q] = (u32)ctot[cls[q]] + (u32)tab[((size_t)cls[q] * per + q) * kWaves + wave] + pos[q];
    }
    RIOGP_KT(p, 4, 4);
    return all;
}

// ------------------------------------------------------------------------------------------------
// K4  k_fill — everything behind the cut search in ONE launch per water-fill round:
//       pass A (APPLY)  re-mark the rejected claimants (row i of a claimant of node a is rejected iff i >= cutidx[a];
//                       every claimant of a forced node is), rebuild the per-wave spill totals, optionally (PACK) copy the
//                       rows that go on to the water-fill to the front of the wave's range in the pack columns;
//       order           the nodes by capacity class, in LDS (rank_by_class): no ranking launch, no comparison sort;
//       pass B (FILL)   the water-fill round: the spill set in index order has exclusive load prefix Q, the node whose
//                       cumulative-free interval [C[k], C[k+1]) contains Q takes the row iff it fits entirely.
//     Round 0 of a solve is APPLY + FILL (this replaced three launches: re-marking, ranking, water-fill); later rounds are
//     FILL only (two launches each before); the row-sharded solve, whose ranks exchange their admitted loads
//     between the cut and the rounds, uses APPLY only and FILL only.
//     What makes APPLY + FILL one launch: the ordered spill prefix at a workgroup's first row needs the spill load of every
//     EARLIER block after the cut — the candidates k_scan counted (bsp_sum) plus the claim load the cut rejected there,
//     R[b], which k_resolve (blocks behind a node's cut block: the whole claim load of the node) and the cut search (the cut
//     block itself) accumulate without touching a row.  Inside the block the workgroup has its own rows.
//     `used` is never read and written in the same round — a workgroup that starts late must order the nodes by the same free
//     capacities as one that started early: round r reads Uread + D[0] + ... + D[r-1] and adds what it admits into Uadd.
//     One GPU: Uread = the solve's used_cur, Uadd = D[r] (zeroed by k_resolve); the host folds used_cur + sum D into the
//     committed vector when somebody needs it (k_used_fold).  Row-sharded solve: Uread = the snapshot of the global vector the
//     exchange left (gprev), D = nullptr, Uadd = used_cur (what the next exchange exports).
// ------------------------------------------------------------------------------------------------

struct FillArgs {
    const u32* cur; const u32* load; const u32* aff; u32* next;   // the table (pass A; pass B too unless PACK)
    const u32* alive_bits;
    Plan p;
    const u32* cutidx; const u32* forced_bits;                     // APPLY
    const u64* cap; const u64* Uread; const u64* D; u64* Uadd; u32 round;   // FILL
    const u64* wsp_sum_in; const u32* wsp_cnt_in; u64* wsp_sum_out; u32* wsp_cnt_out;
    const u64* bsp_sum_in; const u32* bsp_cnt_in; u64* bsp_sum_out; u32* bsp_cnt_out;
    const u64* R; const u64* RP; u32 ng;                           // APPLY && FILL: the cuts' rejected claim load (k_scan's comment)
    int last;                                                      // 0 | 1 last round | 2 last round, NONE already in the real rows
    DevStats* stats;
    const u32* pk_idx; u32* real_next;                             // rows are packed: decisions also go to real_next[pk_idx[pos]]
    const u64* rank_base; const u64* pending_global;               // row-sharded solve
    const u64* run_if;                                             // APPLY only: nothing to do when *run_if == 0 (nullptr: run)
    FxRows fx;
    PackOut pko;                                                   // PACK
};

// LDS layout of k_fill (bytes): [0,512) small | C[m+1] u64 | ord[m] u16 | X = max(adm[m] u64, thr[mr] u32 + alv + rings, class table)
__host__ __device__ __forceinline__ size_t fill_lds_x(u32 m, u32 mwords, bool pack) {
    const size_t a = (size_t)m * sizeof(u64);
    const size_t b = (size_t)((m + 3) & ~3u) * sizeof(u32) + (size_t)((mwords + 3) & ~3u) * sizeof(u32) +
                     (pack ? (size_t)kWaves * 2 * kStageCap * sizeof(u32) : 0);
    const size_t c = rank_tab_bytes(m);
    const size_t x = a > b ? a : b;
    return ((x > c ? x : c) + 15) & ~(size_t)15;
}
__host__ __device__ __forceinline__ size_t fill_lds_off_ord(u32 m) { return 512 + (((size_t)(m + 1) * sizeof(u64) + 15) & ~(size_t)15); }
__host__ __device__ __forceinline__ size_t fill_lds_off_x(u32 m) { return fill_lds_off_ord(m) + (((size_t)m * sizeof(unsigned short) + 15) & ~(size_t)15); }

template <bool VIRT, bool APPLY, bool FILL, bool PACK>
__global__ __launch_bounds__(kBlock) void k_fill(const FillArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const Plan& p = a.p;
    const u32 m = p.m;
    const bool sa_ = p.sa != 0;  // claims do not need a live node (Plan::sa)
    // red[16]: 0 spill load of the earlier blocks | 1 pending rows anywhere | 2 placed rows | 3 placed load | 4 remaining load |
    //          5 remaining rows | 6 rejected rows | 7 rejected load | 8 nodes with room (u32)
    u64* red = reinterpret_cast<u64*>(smem);                      // [16]
    u64* part = reinterpret_cast<u64*>(smem + 128);               // [16] block-scan partials
    u64* wsum = reinterpret_cast<u64*>(smem + 256);               // [16] spill load per wave of this block (after pass A)
    u32* wcn = reinterpret_cast<u32*>(smem + 384);                // [16] pending rows per wave
    u64* C = reinterpret_cast<u64*>(smem + 512);                  // [m+1]
    unsigned short* ord = reinterpret_cast<unsigned short*>(smem + fill_lds_off_ord(m));  // [m] node of rank k
    unsigned char* X = smem + fill_lds_off_x(m);
    u64* adm = reinterpret_cast<u64*>(X);                         // [m] admitted load by node (pass B)
    u32* thr = reinterpret_cast<u32*>(X);                         // [mr] first rejected row of a node's claimants (pass A)
    u32* alv = thr + ((m + 3) & ~3u);                             // [mwords]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr u32 kKeepVal = 0xFFFFFFF0u;  // "no store for this row" among the values bound for the real column
    const u64 gw = (u64)blockIdx.x * kWaves + wave;
    const u32 G = p.G, w0 = blockIdx.x * kWaves;

    constexpr int kt = APPLY ? 1 : 2;  // trace table (lab build)
    RIOGP_KT(p, kt, 0);
    // ---- prologue: every global operand is requested before the first one is used, with clamped addresses instead of
    //      predicates (a load behind a branch turns every later wait into a wait for ALL loads)
    if (APPLY && !FILL && a.run_if && *a.run_if == 0) return;  // (wave-uniform, before anything is requested)
    const u32 wc = *(p.wcnt ? p.wcnt + gw : a.wsp_cnt_in + gw);
    const u32 pc = a.wsp_cnt_in[gw];
    const u64 ncut = a.stats->n_cut;
    const u32 per = (m + kBlock - 1) / kBlock;  // <= 8
    u32 tv[8];
    u32 aw = 0;
    const u32 ak = (u32)tid < p.mwords ? (u32)tid : p.mwords - 1;  // mwords <= 256 < kBlock
    if (APPLY) {
#pragma unroll
        for (u32 q = 0; q < 8; ++q) tv[q] = 0;
        tv[0] = a.cutidx[(u32)tid < m ? (u32)tid : m - 1];
        if (per > 1) {
#pragma unroll
            for (u32 q = 1; q < 8; ++q) {
                const u32 j = tid + (q < per ? q : per - 1) * kBlock;
                tv[q] = a.cutidx[j < m ? j : m - 1];
            }
        }
        aw = a.alive_bits[ak];
    }
    // free capacity of this thread's nodes (tid, tid + 1024, ...): what the water-fill orders the nodes by
    u64 fr[8];
    u32 cnt = 0;
    u64 pgv = 0, rbv = 0;
    if (FILL) {
#pragma unroll
        for (u32 q = 0; q < 8; ++q) fr[q] = 0;
        pgv = a.pending_global ? *a.pending_global : 0ull;
        rbv = a.rank_base ? *a.rank_base : 0ull;
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
            if (q < per) {  // (uniform)
                const u32 j = tid + q * kBlock, jj = j < m ? j : m - 1;
                u64 u = a.Uread[jj];
                if (a.D)
                    for (u32 r = 0; r < a.round; ++r) u += a.D[(size_t)r * m + jj];
                const u64 c = a.cap[jj];
                const bool al = bit_of(a.alive_bits, jj);
                fr[q] = (j < m && al && c > u) ? c - u : 0ull;
            }
        }
    }
    const u32 bt = (u32)tid < G ? (u32)tid : G - 1;
    const u64 bv = a.bsp_sum_in[bt] + ((APPLY && FILL) ? a.R[bt] : 0ull);
    // ... and, by node group, the claim load the cuts reject in the blocks before this one (column blockIdx.x of RP)
    const u64 rpv = (APPLY && FILL) ? a.RP[(size_t)((u32)tid < a.ng ? (u32)tid : a.ng - 1) * G + blockIdx.x] : 0ull;
    const u32 bc = a.bsp_cnt_in[bt];
    const u64 pw = a.wsp_sum_in[w0 + (tid & (kWaves - 1))];
    u64 wstart = wave_row_lo(p, gw), wend = wave_row_lo(p, gw + 1);
    if (wend > p.n) wend = p.n;
    if (wstart > wend) wstart = wend;
    if (p.wcnt && wstart + wc < wend) wend = wstart + wc;  // packed rows: only the first wcnt[gw] positions hold rows
    u64 bstart = wstart, bend = wend;                      // pass B's rows
    if (FILL && !APPLY && pc == 0) bend = bstart;          // no pending row in this wave's range (later rounds: most waves)
    u32 own_cnt = pc;                                      // pending rows of this wave's range (APPLY: counted by pass A)
    // first tile of the wave's rows: pass A's columns (APPLY) or pass B's (FILL only)
    const u64 rs = (wstart < wend ? wstart : 0ull) + (u64)lane * 4;
    uint4 cvn = make_uint4(0, 0, 0, 0), avn = cvn, lvn = cvn, nvn = cvn, ivn = cvn;
    const bool scat = a.pk_idx != nullptr;
    if (APPLY) {
        cvn = *reinterpret_cast<const uint4*>(a.cur + rs);
        avn = *reinterpret_cast<const uint4*>(a.aff + rs);
        lvn = *reinterpret_cast<const uint4*>(a.load + rs);
        // pass B's first tile of real-row indices (packed rows): a cold read, requested here instead of behind pass A
        if (FILL && !PACK) ivn = *reinterpret_cast<const uint4*>((scat ? a.pk_idx : a.load) + rs);
    } else {
        const u64 rb = (bstart < bend ? bstart : 0ull) + (u64)lane * 4;
        nvn = *reinterpret_cast<const uint4*>(a.next + rb);
        lvn = *reinterpret_cast<const uint4*>(a.load + rb);
        ivn = *reinterpret_cast<const uint4*>((scat ? a.pk_idx : a.load) + rb);
    }

    // ---- is there anything to do?  (every fix-up launch may be speculative: enqueued behind a solve whose verdict
    //      nobody has read yet)
    if (APPLY && !FILL) {
        if (!PACK && ncut == 0 && a.forced_bits == nullptr) return;  // no cut anywhere: k_scan's spill totals stand
        if (PACK && ncut == 0 && a.forced_bits == nullptr && a.bsp_cnt_in[blockIdx.x] == 0) {
            if (lane == 0) a.pko.wcnt[gw] = 0;
            return;
        }
    }
    if (tid < 16) red[tid] = 0;
    if (FILL)  // the class table of the node order (rank_by_class), zeroed ahead of a barrier that is needed anyway
        for (u32 k = tid; k < (u32)(rank_tab_bytes(m) / 4); k += kBlock) reinterpret_cast<u32*>(X)[k] = 0;
    __syncthreads();
    u64 blk_base = 0;  // spill load of every earlier block
    if (FILL) {
        u64 sb = ((u32)tid < G && (u32)tid < blockIdx.x) ? bv : 0ull;
        if (APPLY && (u32)tid < a.ng) sb += rpv;
        u32 c = (u32)tid < G ? bc : 0u;
        sb = wave_sum(sb);
        c = wave_sum32(c);
        if (lane == 0 && (sb | c)) { atomicAdd(&red[0], sb); atomicAdd(&red[1], (u64)c); }
        __syncthreads();
        const bool pending = a.pending_global ? (pgv != 0) : (red[1] != 0 || (APPLY && (ncut != 0 || a.forced_bits != nullptr)));
        if (!pending) {  // nothing pending anywhere: the round is a no-op
            if (lane == 0) {
                a.wsp_sum_out[gw] = 0;
                a.wsp_cnt_out[gw] = 0;
                if (PACK) a.pko.wcnt[gw] = 0;
            }
            if (tid == 0) { a.bsp_sum_out[blockIdx.x] = 0; a.bsp_cnt_out[blockIdx.x] = 0; }
            if (a.last && a.fx.dev && a.fx.seq && tid < 8)  // the host may be spinning on this row
                a.fx.host[(size_t)blockIdx.x * 8 + tid] = tid == 7 ? a.fx.seq : a.fx.dev[(size_t)blockIdx.x * 8 + tid];
            return;
        }
        if (!a.fx.dev && blockIdx.x == 0 && tid == 0) atomicAdd(&a.stats->rounds_run, 1ull);  // (fx rows: counted in the epilogue)
        blk_base = red[0] + rbv;  // row-sharded solve: the spill load of every lower rank comes first
        RIOGP_KT(p, kt, 1);
    }

    // ---- order of the nodes: C[0] = 0, C[k+1] = sat(C[k] + free of rank k), ord[k] = node of rank k
    if (FILL) {
        u32 rk[8];
        cnt = rank_by_class(reinterpret_cast<unsigned short*>(X), per, fr, rk, p);
        // free capacities to their ranks (C[k+1] holds the free capacity of rank k until the scan below), nodes to ord[]
#pragma unroll
        for (u32 q = 0; q < 8; ++q)
            if (q < per && fr[q] != 0) { C[rk[q] + 1] = fr[q]; ord[rk[q]] = (unsigned short)(tid + q * kBlock); }
        __syncthreads();
        RIOGP_KT(p, 4, 5);
        u64 fk[8];
        u64 loc = 0;
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
            const u32 k = tid * per + q;
            fk[q] = (q < per && k < cnt) ? C[k + 1] : 0ull;
            loc = sat_add(loc, fk[q]);
        }
        u64 excl = block_excl_scan_1024(loc, true, part, nullptr);  // (barriers inside: every C[k+1] is read before it is rewritten)
        RIOGP_KT(p, 4, 6);
        if (tid == 0) C[0] = 0;
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
            const u32 k = tid * per + q;
            excl = sat_add(excl, fk[q]);
            if (q < per && k < m) C[k + 1] = k < cnt ? excl : ~0ull;
        }
        __syncthreads();  // C / ord complete; the class table in X is dead
        RIOGP_KT(p, kt, 2);
        RIOGP_KT(p, 4, 7);
    }

    // ---- pass A: re-mark, spill totals, packing
    if (APPLY) {
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
            const u32 j = tid + q * kBlock;
            if (q < per && j < m) thr[j] = (a.forced_bits && bit_of(a.forced_bits, j)) ? 0u : tv[q];
        }
        alv[ak] = aw;  // threads past mwords rewrite the last word with its own value
        __syncthreads();
        u64 sp_sum = 0, rej_sum = 0;
        u32 sp_cnt = 0, rej_cnt = 0;  // wave-uniform
        u64 pk_pos = wstart;          // PACK: this wave's packed write cursor (wave-uniform)
        u32* stage = PACK ? alv + ((p.mwords + 3) & ~3u) + (size_t)wave * 2 * kStageCap : nullptr;
        u32 st_head = 0, st_fill = 0;
        for (u64 it = wstart; it < wend; it += kTile) {
            const u64 i0 = it + (u64)lane * 4;
            const uint4 cv = cvn, av = avn, lv = lvn;
            u32 pm = 0;  // PACK: which of the lane's four rows go on to the water-fill
            const u64 pit = (it + kTile < wend ? it + kTile : it) + (u64)lane * 4;  // next tile in flight (the last re-reads its own)
            cvn = *reinterpret_cast<const uint4*>(a.cur + pit);
            avn = *reinterpret_cast<const uint4*>(a.aff + pit);
            lvn = *reinterpret_cast<const uint4*>(a.load + pit);
            // The rows' `next` values are rebuilt from the columns (what k_scan wrote: kept -> cur, claimant -> affinity,
            // duplicate request -> skip mark, not an object -> NONE, the rest -> spill mark) with the rejected claimants turned
            // into spill marks (PACK: every row that goes on to the water-fill into NONE, final unless a round places it
            // through pk_idx), and a wave that changes anything writes its whole kilobyte back, every lane its 16 bytes.
            uint4 ov;
            bool chg = false;
#define RIOGP_ROW(CC, A, L, O, E)                                                                    \
            {                                                                                        \
                const bool inr = i0 + E < wend;                                                      \
                const bool cin = CC < m, ain = A < m;                                                \
                const u32 cx = cin ? CC : 0u, ax = ain ? A : 0u;                                     \
                const bool kept = VIRT ? cin : (cin & bit_of(alv, cx));                              \
                const bool skip = VIRT && CC == kSkipMark;                                           \
                const bool dead = !VIRT && A == kAffInactive;                                        \
                const bool cl = inr & !kept & !skip & ain & (sa_ | bit_of(alv, ax));                 \
                const bool sp = inr & !kept & !skip & !cl & !dead;                                   \
                const bool rej = cl & ((u32)(i0 + E) >= thr[ax]);                                    \
                const u32 mark = PACK ? kNone : kSpillMark;                                          \
                O = kept ? CC : (cl ? (rej ? mark : A) : (skip ? kSkipMark : (dead ? kNone : mark))); \
                if (PACK) pm |= (u32)(sp | rej) << E;                                                \
                chg |= PACK ? (sp | rej) : rej;                                                      \
                sp_sum += (sp | rej) ? (u64)L : 0ull;                                                \
                rej_sum += rej ? (u64)L : 0ull;                                                      \
                sp_cnt += (u32)__popcll(__ballot(sp | rej));                                         \
                rej_cnt += (u32)__popcll(__ballot(rej));                                             \
            }
            RIOGP_ROW(cv.x, av.x, lv.x, ov.x, 0)
            RIOGP_ROW(cv.y, av.y, lv.y, ov.y, 1)
            RIOGP_ROW(cv.z, av.z, lv.z, ov.z, 2)
            RIOGP_ROW(cv.w, av.w, lv.w, ov.w, 3)
#undef RIOGP_ROW
            if (__ballot(chg)) *reinterpret_cast<uint4*>(a.next + i0) = ov;
            if (PACK) {  // index order = lane-major, then element; through this wave's LDS ring, 64 records per flush
                const u64 b0 = __ballot(pm & 1u), b1 = __ballot(pm & 2u), b2 = __ballot(pm & 4u), b3 = __ballot(pm & 8u);
                if (b0 | b1 | b2 | b3) {
                    const u64 lt = (1ull << lane) - 1ull;
                    u32 e = st_head + st_fill + (u32)(__popcll(b0 & lt) + __popcll(b1 & lt) + __popcll(b2 & lt) + __popcll(b3 & lt));
#define RIOGP_PK(E, L)                                                                                     \
                    if (pm & (1u << E)) {                                                                  \
                        const u32 x = e >= kStageCap ? (e >= 2 * kStageCap ? e - 2 * kStageCap : e - kStageCap) : e;  \
                        stage[x] = (u32)(i0 + E); stage[kStageCap + x] = L; ++e;                           \
                    }
                    RIOGP_PK(0, lv.x)
                    RIOGP_PK(1, lv.y)
                    RIOGP_PK(2, lv.z)
                    RIOGP_PK(3, lv.w)
#undef RIOGP_PK
                    st_fill += (u32)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));
                    __builtin_amdgcn_wave_barrier();
                    while (st_fill >= 64u) {  // wave-uniform
                        u32 x = st_head + (u32)lane;
                        x = x >= kStageCap ? x - kStageCap : x;
                        const u64 o = pk_pos + (u32)lane;
                        a.pko.idx[o] = stage[x]; a.pko.load[o] = stage[kStageCap + x]; a.pko.next[o] = kSpillMark;
                        st_head = st_head + 64u >= kStageCap ? st_head + 64u - kStageCap : st_head + 64u;
                        st_fill -= 64u;
                        pk_pos += 64u;
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
        }
        if (PACK && st_fill) {  // what is left in the ring (< 64 records)
            u32 x = st_head + (u32)lane;
            x = x >= kStageCap ? x - kStageCap : x;
            if ((u32)lane < st_fill) {
                const u64 o = pk_pos + (u32)lane;
                a.pko.idx[o] = stage[x]; a.pko.load[o] = stage[kStageCap + x]; a.pko.next[o] = kSpillMark;
            }
            pk_pos += st_fill;
        }
        sp_sum = wave_sum(sp_sum);
        rej_sum = wave_sum(rej_sum);
        if (lane == 0) {
            if (PACK) a.pko.wcnt[gw] = (u32)(pk_pos - wstart);
            wsum[wave] = sp_sum;
            wcn[wave] = sp_cnt;
            if (rej_cnt) { atomicAdd(&red[6], (u64)rej_cnt); atomicAdd(&red[7], rej_sum); }
        }
        if (!FILL) {
            if (lane == 0) {
                a.wsp_sum_out[gw] = sp_sum;
                a.wsp_cnt_out[gw] = sp_cnt;
                if (sp_cnt) { atomicAdd(&red[4], sp_sum); atomicAdd(&red[5], (u64)sp_cnt); }
            }
            __syncthreads();
            if (tid == 0) {
                fx_add_rejected(a.fx, a.stats, red[6], red[7]);
                a.bsp_sum_out[blockIdx.x] = red[4];
                a.bsp_cnt_out[blockIdx.x] = (u32)red[5];
            }
            return;
        }
        // pass B runs over what pass A left: the packed rows (PACK) or the same rows with their new marks
        if (PACK) { bstart = wstart; bend = pk_pos; }
        if (sp_cnt == 0) bend = bstart;
        own_cnt = sp_cnt;
        __syncthreads();  // wsum complete; thr / alv / rings are dead: the region becomes adm.  (Also orders this wave's
                          // mark / pack stores before its own loads of the same rows below.)
        const u64 rb = (bstart < bend ? bstart : 0ull) + (u64)lane * 4;
        const u32* nsrc = PACK ? a.pko.next : a.next;
        const u32* lsrc = PACK ? a.pko.load : a.load;
        const u32* isrc = PACK ? a.pko.idx : (scat ? a.pk_idx : a.load);
        nvn = *reinterpret_cast<const uint4*>(nsrc + rb);
        lvn = *reinterpret_cast<const uint4*>(lsrc + rb);
        if (PACK) ivn = *reinterpret_cast<const uint4*>(isrc + rb);  // (else: requested in the prologue, rb == rs)
        RIOGP_KT(p, kt, 3);
    } else if (FILL) {
        if (tid < kWaves) wsum[tid] = pw;  // staged for the in-block prefix below
    }
    if (!FILL) return;

    // ---- pass B: the water-fill round
    const bool bscat = PACK || scat;
    u32* const bnext = PACK ? a.pko.next : a.next;
    const u32* const bload = PACK ? a.pko.load : a.load;
    const u32* const bidx = PACK ? a.pko.idx : a.pk_idx;
    for (u32 k = tid; k < m; k += kBlock) adm[k] = 0;
    __syncthreads();
    RIOGP_KT(p, kt, 4);
    u64 run = blk_base;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    const u64 F = C[cnt];
    u64 rem_sum = 0, pl_sum = 0;
    u32 rem_cnt = 0, pl_cnt = 0;
    // Q only grows along a wave's rows, so the position in C[] is carried from tile to tile:
    // lo_run = largest k < cnt with C[k] <= run (wave-uniform; one binary search per wave, then a short gallop per tile)
    u32 lo_run = 0;
    if (cnt) {
        u32 lo = 0, hi = cnt;
        while (hi - lo > 1) {
            const u32 mid = lo + ((hi - lo) >> 1);
            if (C[mid] <= run) lo = mid; else hi = mid;
        }
        lo_run = lo;
    }
    const int last = a.last;
    // Rows that cannot be placed AND need no store are not read: once the ordered prefix has passed this round's total free
    // capacity (a cluster that is full: every wave but the first few, from their first row) nothing behind it in the wave's
    // range can be placed, and no row has to be written when the rows keep their marks for the next round, or (last round over
    // packed rows whose real rows hold NONE already) nobody reads them again.  What is left is counted from the totals the
    // previous launch left for this wave.
    const bool quiet_tail = !last || (bscat && last == 2);
    const u64 run0 = run;
    for (u64 it = bstart; it < bend; it += kTile) {
        if (quiet_tail && (cnt == 0 || run >= F)) {  // (wave-uniform)
            const u32 seen = wave_sum32(pl_cnt + rem_cnt);
            if (lane == 0) { rem_sum += wsum[wave] - (run - run0); rem_cnt += own_cnt - seen; }
            break;
        }
        const u64 i0 = it + (u64)lane * 4;
        const uint4 nv = nvn, lv = lvn, iv = ivn;
        const u64 pit = it + kTile < bend ? it + kTile : it;  // next tile in flight (the last iteration re-reads its own)
        nvn = *reinterpret_cast<const uint4*>(bnext + pit + (u64)lane * 4);
        lvn = *reinterpret_cast<const uint4*>(bload + pit + (u64)lane * 4);
        if (bscat) ivn = *reinterpret_cast<const uint4*>(bidx + pit + (u64)lane * 4);
        const bool mk0 = i0 + 0 < bend && nv.x == kSpillMark, mk1 = i0 + 1 < bend && nv.y == kSpillMark;
        const bool mk2 = i0 + 2 < bend && nv.z == kSpillMark, mk3 = i0 + 3 < bend && nv.w == kSpillMark;
        if (!__ballot(mk0 | mk1 | mk2 | mk3)) continue;  // wave-uniform: nothing to spill in this tile
        const u64 l0 = mk0 ? lv.x : 0, l1 = mk1 ? lv.y : 0;
        const u64 l2 = mk2 ? lv.z : 0, l3 = mk3 ? lv.w : 0;
        const u64 lsum = l0 + l1 + l2 + l3;
        if (cnt == 0 || run >= F) {
            // Nothing from here on can be placed: the prefix Q of every remaining row of this wave is >= run >= F, the total
            // free capacity of this round (wave-uniform, and it stays true for the wave's later tiles): the rows are only counted.
            rem_sum += lsum;
            rem_cnt += (u32)mk0 + (u32)mk1 + (u32)mk2 + (u32)mk3;
            if (last) {
                uint4 ov = nv;
                ov.x = mk0 ? kNone : ov.x; ov.y = mk1 ? kNone : ov.y; ov.z = mk2 ? kNone : ov.z; ov.w = mk3 ? kNone : ov.w;
                *reinterpret_cast<uint4*>(bnext + i0) = ov;  // (every lane: whole lines; lanes without marks rewrite their values)
                if (bscat && last == 1) {
                    if (mk0) a.real_next[iv.x] = kNone;
                    if (mk1) a.real_next[iv.y] = kNone;
                    if (mk2) a.real_next[iv.z] = kNone;
                    if (mk3) a.real_next[iv.w] = kNone;
                }
            }
            continue;
        }
        const u64 inc = wave_incl_scan(lsum, lane);
        u64 Q = run + (inc - lsum);
        const u64 run_end = run + shfl64(inc, 63);
        // wave-uniform bracket [lo_run, hi_run] of this tile's Q range: gallop from the carried position
        u32 hi_run = lo_run;
        {
            u32 step = 1;
            while (hi_run + step < cnt && C[hi_run + step] <= run_end) { hi_run += step; step <<= 1; }
            u32 top = hi_run + step < cnt ? hi_run + step : cnt;  // C[top] > run_end or top == cnt
            while (top - hi_run > 1) {
                const u32 mid = hi_run + ((top - hi_run) >> 1);
                if (C[mid] <= run_end) hi_run = mid; else top = mid;
            }
        }
        uint4 ov = nv;  // the lane's four decisions leave as ONE 16-byte store (rows that stay pending keep their mark)
        uint4 wv = make_uint4(kKeepVal, kKeepVal, kKeepVal, kKeepVal);  // what goes into the real assignment column
        if (hi_run == lo_run && run_end <= C[lo_run + 1]) {
            // The whole tile lies inside ONE node's interval (a node's free capacity is thousands of rows wide: nearly every
            // tile): every marked row fits entirely, no per-row search, one atomic for the tile's admitted load.
            const u32 nd0 = (u32)ord[lo_run];
            ov.x = mk0 ? nd0 : ov.x; ov.y = mk1 ? nd0 : ov.y; ov.z = mk2 ? nd0 : ov.z; ov.w = mk3 ? nd0 : ov.w;
            pl_sum += lsum;
            pl_cnt += (u32)mk0 + (u32)mk1 + (u32)mk2 + (u32)mk3;
            if (lane == 0) atomicAdd(&adm[nd0], run_end - run);
            *reinterpret_cast<uint4*>(bnext + i0) = ov;
            if (bscat) {
                if (mk0) a.real_next[iv.x] = nd0;
                if (mk1) a.real_next[iv.y] = nd0;
                if (mk2) a.real_next[iv.z] = nd0;
                if (mk3) a.real_next[iv.w] = nd0;
            }
            run = run_end;
            continue;
        }
        u32 pn[4] = {kNone, kNone, kNone, kNone};  // node of the lane's placed rows and their loads: admitted after the row loop
        u64 pll[4] = {0, 0, 0, 0};
#define RIOGP_ROW(MK, L, E, OUT, WOUT)                                            \
        if (MK) {                                                                 \
            u32 nd = kNone;                                                       \
            if (Q < F) {                                                          \
                u32 lo = lo_run, hi = hi_run + 1;                                 \
                while (hi - lo > 1) {                                             \
                    const u32 mid = lo + ((hi - lo) >> 1);                        \
                    if (C[mid] <= Q) lo = mid; else hi = mid;                     \
                }                                                                 \
                if (Q + L <= C[lo + 1]) nd = (u32)ord[lo];                        \
            }                                                                     \
            if (nd != kNone) {                                                    \
                OUT = nd;                                                         \
                WOUT = nd;                                                        \
                pn[E] = nd; pll[E] = (u64)L;                                      \
                pl_sum += L; ++pl_cnt;                                            \
            } else {                                                              \
                if (last) {                                                       \
                    OUT = kNone;                                                  \
                    if (last == 1) WOUT = kNone;  /* 2: the real rows already hold NONE */ \
                }                                                                 \
                rem_sum += L; ++rem_cnt;                                          \
            }                                                                     \
            Q += L;                                                               \
        }
        RIOGP_ROW(mk0, l0, 0, ov.x, wv.x)
        RIOGP_ROW(mk1, l1, 1, ov.y, wv.y)
        RIOGP_ROW(mk2, l2, 2, ov.z, wv.z)
        RIOGP_ROW(mk3, l3, 3, ov.w, wv.w)
#undef RIOGP_ROW
        {   // Admitted load per node (adm[], LDS).  The rows of a tile mostly land on ONE node (a node's free capacity is
            // thousands of rows wide): when every placed row of the tile has the same node the wave adds them up in
            // registers (DPP) and issues ONE atomic instead of up to 256 on one address.
            const u32 n0 = pn[0] != kNone ? pn[0] : pn[1] != kNone ? pn[1] : pn[2] != kNone ? pn[2] : pn[3];
            const u64 pmask = __ballot(n0 != kNone);
            if (pmask) {
                const u32 nd0 = (u32)__builtin_amdgcn_readlane((int)n0, __ffsll((long long)pmask) - 1);
                const bool odd = (pn[0] != kNone && pn[0] != nd0) | (pn[1] != kNone && pn[1] != nd0) |
                                 (pn[2] != kNone && pn[2] != nd0) | (pn[3] != kNone && pn[3] != nd0);
                if (!__ballot(odd)) {
                    const u64 tot = wave_sum(pll[0] + pll[1] + pll[2] + pll[3]);
                    if (lane == 0) atomicAdd(&adm[nd0], tot);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (pn[e] != kNone) atomicAdd(&adm[pn[e]], pll[e]);
                }
            }
        }
        if ((ov.x != nv.x) | (ov.y != nv.y) | (ov.z != nv.z) | (ov.w != nv.w)) *reinterpret_cast<uint4*>(bnext + i0) = ov;
        if (bscat) {  // the decisions of the packed rows go to their REAL rows: scattered 4-byte stores
            if (wv.x != kKeepVal) a.real_next[iv.x] = wv.x;
            if (wv.y != kKeepVal) a.real_next[iv.y] = wv.y;
            if (wv.z != kKeepVal) a.real_next[iv.z] = wv.z;
            if (wv.w != kKeepVal) a.real_next[iv.w] = wv.w;
        }
        run = run_end;
        lo_run = hi_run;
    }
    RIOGP_KT(p, kt, 5);
    rem_sum = wave_sum(rem_sum);
    rem_cnt = wave_sum32(rem_cnt);
    pl_sum = wave_sum(pl_sum);
    pl_cnt = wave_sum32(pl_cnt);
    if (lane == 0) {
        a.wsp_sum_out[gw] = rem_sum;
        a.wsp_cnt_out[gw] = rem_cnt;
        if (pl_cnt) { atomicAdd(&red[2], (u64)pl_cnt); atomicAdd(&red[3], pl_sum); }
        if (rem_cnt) { atomicAdd(&red[4], rem_sum); atomicAdd(&red[5], (u64)rem_cnt); }
    }
    __syncthreads();
    RIOGP_KT(p, kt, 6);
    if (tid == 0) { a.bsp_sum_out[blockIdx.x] = red[4]; a.bsp_cnt_out[blockIdx.x] = (u32)red[5]; }
    for (u32 k = tid; k < m; k += kBlock)
        if (adm[k]) atomicAdd(&a.Uadd[k], adm[k]);  // integer sums: order-independent
    if (a.fx.dev) {  // this workgroup's row of the fix-up counters, and its copy in the host's pinned slot: one round trip
        if (tid < 8) {
            u64* r = a.fx.dev + (size_t)blockIdx.x * 8;
            const u64 add = tid == 0 ? red[6] : tid == 1 ? red[7] : tid == 2 ? red[2] : tid == 3 ? red[3]
                          : (tid == 4 && last) ? red[5] : (tid == 5 && last) ? red[4]
                          : (tid == 6 && blockIdx.x == 0) ? 1ull : 0ull;  // [6] of row 0 = rounds run
            const u64 v = (tid == 7 && last && a.fx.seq) ? a.fx.seq : r[tid] + add;  // [7] = the host's sequence number
            r[tid] = v;
            a.fx.host[(size_t)blockIdx.x * 8 + tid] = v;
        }
    } else if (tid == 0) {
        if (APPLY && red[6]) { atomicAdd(&a.stats->rejected, red[6]); atomicAdd(&a.stats->load_rejected, red[7]); }
        if (red[2]) { atomicAdd(&a.stats->spilled, red[2]); atomicAdd(&a.stats->load_spilled, red[3]); }
        if (last && red[5]) { atomicAdd(&a.stats->unplaced, red[5]); atomicAdd(&a.stats->load_unplaced, red[4]); }
    }
    RIOGP_KT(p, kt, 7);
}

// ------------------------------------------------------------------------------------------------
// K4c k_cut_apply + k_cut_settle — the exact cuts AND the re-marking (AND the packing) of a whole-table solve in ONE pass over
//     the rows that need it.  Before: k_cut_find (a pass over the blocks that own cuts, 25 us at 10 M rows) and then pass A of
//     k_fill (another 12 B/row) — two table-wide passes between k_resolve and the water-fill.
//     What makes one pass enough: after k_resolve a claimant of node a in block b is admitted when cutblk[a] > b, rejected when
//     cutblk[a] < b, and UNDECIDED only when cutblk[a] == b — a few dozen rows per (node, block).  And a block none of whose
//     nodes is cut at or before it and that holds no spill candidate has nothing to re-mark at all (a contended cluster: the
//     first 70-90 % of the blocks).
//     What decides the launch shape: a CU streams ~25 GB/s whatever the rest of the chip does (256 of them are the chip's
//     6.4 TB/s), so the blocks that DO have work — the last 10-30 % of the index range — must not be left to their own CUs: the
//     first version of this kernel (workgroup b = block b) took as long over a fifth of the table as over all of it.  The wave
//     ranges (the unit whose rows are packed in order) are therefore DEALT OUT: wave w of workgroup g takes wave range
//     w * G + g, i.e. the sixteen waves of a workgroup work in sixteen different blocks spread evenly over the table, and every
//     CU gets its share of whatever part of the table has work.  Nothing of a block lives in one workgroup's LDS any more:
//       k_cut_apply   per wave range: stream cur / aff / load once (12 B/row, two tiles in flight), rebuild the rows' `next`
//                     values, admit and reject wholesale by cut block (one LDS look-up per row: the cut block of its affinity
//                     node, liveness folded in), pack in index order through the wave's LDS ring — PACK: every row that goes
//                     on to the water-fill plus the undecided ones (marked kUndTag | node); else the undecided rows only
//                     (scratch); an undecided row adds its load to Tg[node][wave of its block] (global atomic, no return) and
//                     leaves {packed position, row, load, node} in the range's list (global scratch).
//       k_cut_settle  workgroup b = block b, only where nodes are cut: per node an ordered walk over its sixteen wave sums
//                     finds the wave in which the claim prefix crosses the budget k_resolve left; wave w settles the undecided
//                     rows of wave range (b, w) from its list — earlier wave: admitted, later wave: rejected, the cut wave
//                     itself: exactly, by load prefix in index order.  Inside the cut wave a node's rows of one step (64 rows)
//                     are summed without order first; only the ONE step in which the prefix crosses the budget needs the
//                     ordered scan (one DPP scan per crossing).  More cut nodes in a block than fit the LDS (a nearly full
//                     cluster cuts every node in block 0): groups of kmax, one more walk over the list per further group.
//                     Every workgroup: the block's spill totals from its sixteen ranges.
//     Out: `next` re-marked (PACK: NONE in every row that goes on, so the last round only writes what it places), the packed
//     rows {row, load, spill mark} + per-range counts, the ordered spill totals per range / block (what k_fill<FILL>'s prefix
//     starts from: R / RP are not needed), cutidx / used_cur of the cut nodes, the rejected rows' counters.
//     Real table of one GPU only (the request path's virtual table and the row-sharded solve keep k_cut_find + k_fill<APPLY>).
// ------------------------------------------------------------------------------------------------
constexpr u32 kCbNoClaim = 0xFFFFFFFEu;   // cbt[] entry of an affinity that is no claim target (a node that is not alive — unless
                                          // claims need no live node, Plan::sa —, or no node at all: cbt[m]); kNoCut = no cut
constexpr u32 kUndTag = 0x80000000u;      // packed mark of an undecided row: kUndTag | node
constexpr u32 kCaRingCols = 3;
__host__ __device__ __forceinline__ size_t ca_lds_bytes(u32 m, u32 mwords) {
    return 1024 + (size_t)((m + 4) & ~3u) * 4 + (size_t)((mwords + 3) & ~3u) * 4 + 64;
}
struct UndList { u32* pos; u32* row; u32* load; u32* node; u32* cnt; };  // per wave range, at the front of the range: the undecided rows
struct CutApplyArgs {
    const u32* cur; const u32* load; const u32* aff; u32* next;
    const u32* alive_bits; Plan p;
    const u32* cutblk;
    u64* Tg;                         // [m][16] claim load of node j's undecided rows per wave of its cut block (zeroed by k_resolve)
    u64* wsp_sum_out; u32* wsp_cnt_out;
    const u32* bsp_cnt_in;           // k_scan's spill candidates per block
    DevStats* stats; FxRows fx;
    PackOut pko;                     // PACK: the rows that go on + the undecided ones; else: scratch for the undecided rows
    UndList ul;
};

template <bool PACK, bool ALLALIVE>
__global__ __launch_bounds__(kBlock) void k_cut_apply(const CutApplyArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const Plan& p = a.p;
    const u32 m = p.m;
    const bool sa_ = p.sa != 0;
    // small: [0] first block with a cut | [16..48) rejected rows / load of this workgroup | [64..) per range of this workgroup
    u32* fbp = reinterpret_cast<u32*>(smem);
    u64* red = reinterpret_cast<u64*>(smem + 16);                             // [2]
    u32* wlist = reinterpret_cast<u32*>(smem + 64);                           // [16] which of this workgroup's ranges have work (k), [16] = how many
    u32* pcnt = reinterpret_cast<u32*>(smem + 192);                           // [2][16] rows a wave packs in this step (two steps alive)
    u32* ucnt = pcnt + 2 * kWaves;                                            // [2][16] ... of them undecided
    u64* rsum = reinterpret_cast<u64*>(smem + 512);                           // [16][4] per range: pending load, candidates' load, pending rows, candidates
    u32* cbt = reinterpret_cast<u32*>(smem + 1024);                           // [m + 1] cut block by node | kCbNoClaim
    u32* alv = cbt + ((m + 4) & ~3u);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    RIOGP_KT(p, 6, 0);
    // one round trip: the cut flag, the spill candidates of the blocks of this workgroup's ranges, cutblk and liveness words
    const u64 ncut = a.stats->n_cut;
    // gridDim.x = a small multiple of the plan's G (two workgroups per CU: while one waits for its step's loads and stores the
    // other computes): this workgroup's ranges are k * gridDim.x + blockIdx.x
    const u32 GG = gridDim.x, nk = (p.nw + GG - 1 - blockIdx.x) / GG;          // (ranges of this workgroup: <= 16)
    const u32 kk = (u32)tid & (kWaves - 1);
    const u64 gwk = (u64)kk * GG + blockIdx.x;                                 // the workgroup's k-th range (thread kk looks at it)
    const u32 scand_k = kk < nk ? a.bsp_cnt_in[(u32)(gwk / kWaves)] : 0u;
    const u32 cb0 = a.cutblk[(u32)tid < m ? (u32)tid : m - 1];
    const u32 ak = (u32)tid < p.mwords ? (u32)tid : p.mwords - 1;
    const u32 aw = a.alive_bits[ak];
    // speculative launch behind a solve that needs no fix-up: k_scan's marks, counts and totals stand
    if (ncut == 0 && !PACK) return;
    if (tid < 2) red[tid] = 0;
    if (tid == 0) *fbp = kNoCut;
    if (tid < kWaves * 4) rsum[tid] = 0;
    alv[ak] = aw;
    __syncthreads();
    u32 fmin = kNoCut;
    for (u32 j = tid; j < m; j += kBlock) {
        u32 cb = j == (u32)tid ? cb0 : a.cutblk[j];
        if (!ALLALIVE && !sa_ && !bit_of(alv, j)) cb = kCbNoClaim;  // (a node without claimants has no cut)
        cbt[j] = cb;
        fmin = cb < fmin ? cb : fmin;
    }
    if (tid == 0) cbt[m] = kCbNoClaim;
    {
        u32 t;  // minimum over the wave, then over the workgroup
        t = (u32)__shfl_xor((int)fmin, 1, 64); fmin = t < fmin ? t : fmin;
        t = (u32)__shfl_xor((int)fmin, 2, 64); fmin = t < fmin ? t : fmin;
        t = (u32)__shfl_xor((int)fmin, 4, 64); fmin = t < fmin ? t : fmin;
        t = (u32)__shfl_xor((int)fmin, 8, 64); fmin = t < fmin ? t : fmin;
        t = (u32)__shfl_xor((int)fmin, 16, 64); fmin = t < fmin ? t : fmin;
        t = (u32)__shfl_xor((int)fmin, 32, 64); fmin = t < fmin ? t : fmin;
        if (lane == 0 && fmin != kNoCut) atomicMin(fbp, fmin);
    }
    __syncthreads();
    const u32 fb = *fbp;
    // which ranges have work: a range of a block no node is cut in or before, without spill candidates, stands as k_scan wrote it
    // (blocks are in index order: a node cut in block c makes every block >= c a block with work); wave 0 lists them in order
    if (wave == 0) {
        u64 ws = 0, we = 0;
        bool wk = false;
        if (lane < kWaves && (u32)lane < nk) {
            wave_range_plain(p, gwk, ws, we);
            wk = ws < we && ((u32)(gwk / kWaves) >= fb || scand_k != 0);
            if (!wk) {  // nothing goes on from this range
                a.pko.wcnt[gwk] = 0;
                a.ul.cnt[gwk] = 0;
                a.wsp_sum_out[gwk] = 0;
                a.wsp_cnt_out[gwk] = 0;
            }
        }
        const u64 bal = __ballot(wk);
        if (wk) wlist[__popcll(bal & ((1ull << lane) - 1ull))] = (u32)lane;
        if (lane == 0) wlist[kWaves] = (u32)__popcll(bal);
    }
    __syncthreads();
    const u32 nwork = wlist[kWaves];
    RIOGP_KT(p, 6, 1);

    // ---- the ranges with work, one after the other; a STEP = sixteen consecutive tiles of the range, one per wave, so that a
    //      range's rows are in flight in sixteen waves at once (a wave that walks a whole range alone takes 1.7 us per tile:
    //      instruction and memory latency in series, nothing to overlap them with).  The packed positions need the waves'
    //      counts in order: one barrier per step (counts double-buffered in the LDS).
    constexpr u32 kMark = PACK ? kNone : kSpillMark;
    const u64 lt = (1ull << lane) - 1ull;
    u64 rej_sum = 0, rej_cnt = 0;  // (wave-uniform totals are added at the end)
    // the step the wave works on and the one it has requested
    u32 wi = 0;                    // index into wlist
    u64 rs = 0, re = 0, chunk = 0; // the range and the first row of the step
    auto range_of = [&](const u32 w_i, u64& s_, u64& e_) {
        const u64 g = (u64)wlist[w_i] * GG + blockIdx.x;
        wave_range_plain(p, g, s_, e_);
    };
    uint4 cq = make_uint4(0, 0, 0, 0), aq = cq, lq = cq;  // the requested step's tile
    auto request = [&](const u64 c0, const u64 e_) {
        const u64 t0 = c0 + (u64)wave * kTile;
        const u64 i = (t0 < e_ ? t0 : c0) + (u64)lane * 4;   // (waves past the range's end re-read its first tile: a hit, no over-read)
        cq = *reinterpret_cast<const uint4*>(a.cur + i);
        aq = *reinterpret_cast<const uint4*>(a.aff + i);
        lq = *reinterpret_cast<const uint4*>(a.load + i);
    };
    if (nwork) { range_of(0, rs, re); chunk = rs; request(chunk, re); }
    u32 base = 0, ubase = 0;       // rows packed / undecided rows listed of the range so far (workgroup-uniform)
    u32 par = 0;
    while (wi < nwork) {           // (workgroup-uniform)
        const uint4 c = cq, aa = aq, l = lq;
        const u64 t0 = chunk + (u64)wave * kTile;
        const bool live = t0 < re;                          // this wave has a tile in this step (wave-uniform)
        const u64 i0 = t0 + (u64)lane * 4;
        // the next step: the range's next sixteen tiles, or the next range's first
        u32 nwi = wi;
        u64 nrs = rs, nre = re, nchunk = chunk + (u64)kWaves * kTile;
        if (nchunk >= re) {
            nwi = wi + 1;
            if (nwi < nwork) { range_of(nwi, nrs, nre); nchunk = nrs; }
        }
        if (nwi < nwork) request(nchunk, nre);
        const u32 k = wlist[wi];
        const u64 gw = (u64)k * GG + blockIdx.x;
        const u32 vb = (u32)(gw / kWaves), vw = (u32)(gw % kWaves);
        const bool check = t0 + kTile > re;
        uint4 ov;
        // A row is pending unless it sits on a live node; what becomes of a pending row is ONE table look-up by its affinity (the
        // node's cut block against this range's block): admitted | rejected | undecided | no claim target.
#define RIOGP_ROW(CC, A, LL, O, E, PD, UN, SP, AX)                                                    \
        const bool nk##E = live && (!check || i0 + E < re) && !(CC < m && (ALLALIVE || bit_of(alv, CC < m ? CC : 0u)));  \
        const u32 AX = A < m ? A : m;                                                                 \
        const u32 cb##E = cbt[AX];                                                                    \
        const bool UN = nk##E && cb##E == vb;                                                         \
        const bool SP = nk##E && cb##E == kCbNoClaim && A != kAffInactive;                            \
        const bool PD = (nk##E && cb##E < vb) || SP;                                                  \
        O = !nk##E ? CC : ((cb##E >= vb && cb##E != kCbNoClaim) ? A : (A == kAffInactive ? kNone : kMark));
        RIOGP_ROW(c.x, aa.x, l.x, ov.x, 0, pd0, un0, sp0, ax0)
        RIOGP_ROW(c.y, aa.y, l.y, ov.y, 1, pd1, un1, sp1, ax1)
        RIOGP_ROW(c.z, aa.z, l.z, ov.z, 2, pd2, un2, sp2, ax2)
        RIOGP_ROW(c.w, aa.w, l.w, ov.w, 3, pd3, un3, sp3, ax3)
#undef RIOGP_ROW
        const u64 d0 = __ballot(pd0), d1 = __ballot(pd1), d2 = __ballot(pd2), d3 = __ballot(pd3);
        const u64 u0 = __ballot(un0), u1 = __ballot(un1), u2 = __ballot(un2), u3 = __ballot(un3);
        const bool anyu = (u0 | u1 | u2 | u3) != 0;
        // what is packed: PACK — the rows that go on and the undecided ones; else the undecided rows only
        const u64 b0 = PACK ? d0 | u0 : u0, b1 = PACK ? d1 | u1 : u1, b2 = PACK ? d2 | u2 : u2, b3 = PACK ? d3 | u3 : u3;
        const u32 mycnt = (u32)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));
        const u32 myund = (u32)(__popcll(u0) + __popcll(u1) + __popcll(u2) + __popcll(u3));
        if (lane == 0) { pcnt[par * kWaves + wave] = mycnt; ucnt[par * kWaves + wave] = myund; }
        // the rows' `next` values: a wave that changes anything writes its whole kilobyte (PACK: every row that goes on takes
        // NONE; else: the rejected claimants take the spill mark — the candidates hold it since k_scan)
        if (live && (PACK ? (d0 | d1 | d2 | d3) != 0 : __ballot((pd0 && !sp0) | (pd1 && !sp1) | (pd2 && !sp2) | (pd3 && !sp3)) != 0))
            *reinterpret_cast<uint4*>(a.next + i0) = ov;
        // the range's totals (LDS, per range of this workgroup)
        if (d0 | d1 | d2 | d3) {  // (wave-uniform)
            const u64 ps = wave_sum((pd0 ? (u64)l.x : 0ull) + (pd1 ? (u64)l.y : 0ull) + (pd2 ? (u64)l.z : 0ull) + (pd3 ? (u64)l.w : 0ull));
            const u32 pc = (u32)(__popcll(d0) + __popcll(d1) + __popcll(d2) + __popcll(d3));
            u64 cs = 0;
            u32 cc = 0;
            if (__ballot(sp0 | sp1 | sp2 | sp3)) {  // (rare) spill candidates: counted apart, the rejected rows are the difference
                cs = wave_sum((sp0 ? (u64)l.x : 0ull) + (sp1 ? (u64)l.y : 0ull) + (sp2 ? (u64)l.z : 0ull) + (sp3 ? (u64)l.w : 0ull));
                cc = (u32)(__popcll(__ballot(sp0)) + __popcll(__ballot(sp1)) + __popcll(__ballot(sp2)) + __popcll(__ballot(sp3)));
            }
            if (lane == 0) {
                atomicAdd(&rsum[k * 4 + 0], ps); atomicAdd(&rsum[k * 4 + 2], (u64)pc);
                if (cc) { atomicAdd(&rsum[k * 4 + 1], cs); atomicAdd(&rsum[k * 4 + 3], (u64)cc); }
            }
            rej_sum += ps - cs;
            rej_cnt += pc - cc;
        }
        __syncthreads();  // the step's counts are complete (and the previous step's are dead: the other half of the buffers)
        u32 off = base, uoff = ubase, tot = 0, utot = 0;
        {
            const u32 pv = lane < kWaves ? pcnt[par * kWaves + (lane & (kWaves - 1))] : 0u;
            const u32 uv = lane < kWaves ? ucnt[par * kWaves + (lane & (kWaves - 1))] : 0u;
            const u64 both = wave_incl_scan((u64)pv | ((u64)uv << 32), lane);            // (two 32-bit prefixes in one scan)
            const u64 mine = wave ? shfl64(both, wave - 1) : 0ull;
            const u64 all = shfl64(both, kWaves - 1);
            off += (u32)mine; uoff += (u32)(mine >> 32);
            tot = (u32)all; utot = (u32)(all >> 32);
        }
        if (mycnt) {  // (wave-uniform) this wave's records, in index order = lane-major, then element
            const u64 o0 = rs + off;
            if (PACK && !anyu && mycnt == (u32)kTile) {
                // EVERY row of the tile goes on and none is undecided (the blocks behind the cuts: every tile): three 16-byte
                // stores per lane, straight from the registers
                const u64 o = o0 + (u64)lane * 4;
                u32x4u vi, vl, vm;
                vi.x = (u32)i0; vi.y = (u32)i0 + 1u; vi.z = (u32)i0 + 2u; vi.w = (u32)i0 + 3u;
                vl.x = l.x; vl.y = l.y; vl.z = l.z; vl.w = l.w;
                vm.x = kSpillMark; vm.y = kSpillMark; vm.z = kSpillMark; vm.w = kSpillMark;
                *reinterpret_cast<u32x4u*>(a.pko.idx + o) = vi;
                *reinterpret_cast<u32x4u*>(a.pko.load + o) = vl;
                *reinterpret_cast<u32x4u*>(a.pko.next + o) = vm;
            } else {
                const bool p0 = PACK ? (pd0 || un0) : un0, p1 = PACK ? (pd1 || un1) : un1, p2 = PACK ? (pd2 || un2) : un2,
                           p3 = PACK ? (pd3 || un3) : un3;
                u32 pe = off + (u32)(__popcll(b0 & lt) + __popcll(b1 & lt) + __popcll(b2 & lt) + __popcll(b3 & lt));  // from the range's first
                u64 ue = 0;
                // a HOT node (a skewed cluster: half the rows of a block ask for one server): >= 16 undecided rows of one element
                // share their node — one global atomic for their sum instead of 16+ on one address
                bool hot0 = false, hot1 = false, hot2 = false, hot3 = false;
                u64* const Tw = a.Tg + vw;
                if (anyu) {
                    ue = rs + uoff + (u32)(__popcll(u0 & lt) + __popcll(u1 & lt) + __popcll(u2 & lt) + __popcll(u3 & lt));
#define RIOGP_HOT(UB, UN, AX, LL, HOT)                                                                    \
                    if (__popcll(UB) >= 16) {                                                             \
                        const u32 n0 = (u32)__builtin_amdgcn_readlane((int)AX, __ffsll((long long)UB) - 1); \
                        const bool same = UN && AX == n0;                                                 \
                        if (__popcll(__ballot(same)) >= 16) {                                             \
                            const u64 sum = wave_sum(same ? (u64)LL : 0ull);                              \
                            if (lane == 0) atomicAdd(Tw + (size_t)n0 * kWaves, sum);                      \
                            HOT = same;                                                                   \
                        }                                                                                 \
                    }
                    RIOGP_HOT(u0, un0, ax0, l.x, hot0)
                    RIOGP_HOT(u1, un1, ax1, l.y, hot1)
                    RIOGP_HOT(u2, un2, ax2, l.z, hot2)
                    RIOGP_HOT(u3, un3, ax3, l.w, hot3)
#undef RIOGP_HOT
                }
#define RIOGP_PK(E, P, UN, AX, LL, HOT)                                                                   \
                if (P) {                                                                                  \
                    const u64 o = rs + pe;                                                                \
                    a.pko.idx[o] = (u32)i0 + E; a.pko.load[o] = LL; a.pko.next[o] = UN ? (kUndTag | AX) : kSpillMark;  \
                    if (UN) {                                                                             \
                        if (!HOT) atomicAdd(Tw + (size_t)AX * kWaves, (u64)LL);                           \
                        a.ul.pos[ue] = pe; a.ul.row[ue] = (u32)i0 + E; a.ul.load[ue] = LL; a.ul.node[ue] = AX;  \
                        ++ue;                                                                             \
                    }                                                                                     \
                    ++pe;                                                                                 \
                }
                RIOGP_PK(0, p0, un0, ax0, l.x, hot0)
                RIOGP_PK(1, p1, un1, ax1, l.y, hot1)
                RIOGP_PK(2, p2, un2, ax2, l.z, hot2)
                RIOGP_PK(3, p3, un3, ax3, l.w, hot3)
#undef RIOGP_PK
            }
        }
        base += tot; ubase += utot;
        par ^= 1;
        if (nwi != wi) {  // the range is through: its counts (thread 0), then the next range starts from zero
            if (tid == 0) { a.pko.wcnt[gw] = base; a.ul.cnt[gw] = ubase; }
            base = 0; ubase = 0;
        }
        wi = nwi; rs = nrs; re = nre; chunk = nchunk;
    }
    RIOGP_KT(p, 6, 2);
    __syncthreads();
    // ---- what goes on to the water-fill from each range (k_cut_settle adds the undecided rows it rejects and the block totals)
    if ((u32)tid < nwork) {
        const u32 k = wlist[tid];
        const u64 gw = (u64)k * GG + blockIdx.x;
        a.wsp_sum_out[gw] = rsum[k * 4 + 0];
        a.wsp_cnt_out[gw] = (u32)rsum[k * 4 + 2];
    }
    if (lane == 0 && rej_cnt) { atomicAdd(&red[0], rej_cnt); atomicAdd(&red[1], rej_sum); }
    __syncthreads();
    if (tid == 0 && red[0]) {  // (several workgroups share a counter row: atomics)
        if (a.fx.dev) {
            u64* r = a.fx.dev + (size_t)(blockIdx.x % p.G) * 8;
            atomicAdd(&r[0], red[0]);
            atomicAdd(&r[1], red[1]);
        } else {
            atomicAdd(&a.stats->rejected, red[0]);
            atomicAdd(&a.stats->load_rejected, red[1]);
        }
    }
    RIOGP_KT(p, 6, 7);
}

// k_cut_settle's LDS: small | slot[mr] u16 | node_of[mr] u16 | per group of kmax cut nodes: T[kmax][17] and five words of state
struct CsLds { size_t slot, node_of, grp; u32 kmax; size_t total; };
__host__ __device__ __forceinline__ CsLds cs_lds(u32 m) {
    CsLds L;
    const u32 mr = (m + 7) & ~7u;
    size_t off = 256;
    L.slot = off; off += (size_t)mr * 2;
    L.node_of = off; off += (size_t)mr * 2;
    off = (off + 15) & ~(size_t)15;
    L.grp = off;
    const size_t per = 17 * 8 + 5 * 8 + 2 * 4;             // T row (odd stride) | rem, pre, acc, ssum, adm | cw, cutrow
    const size_t avail = (size_t)144 * 1024 - off;
    u32 k = (u32)(avail / per);
    if (k > m) k = (m + 7) & ~7u;
    if (k > 1024u) k = 1024u;
    if (k < 8u) k = 8u;
    L.kmax = k & ~7u;
    L.total = off + (size_t)L.kmax * per + 64;
    return L;
}
struct CutSettleArgs {
    u32* next; Plan p;
    const u32* cutblk; const u64* budget; const u64* admpre; const u64* used_kept;
    u32* cutidx; u64* used_cur;
    const u64* Tg;
    u64* wsp_sum; u32* wsp_cnt; u64* bsp_sum; u32* bsp_cnt;   // in: the ranges' totals without the undecided rows; out: final (+ the blocks')
    DevStats* stats; FxRows fx;
    u32* pk_next;                    // the packed rows' marks (PACK: what the rounds read; else scratch)
    UndList ul;
    u32 pack;                        // rejected rows take NONE (the rounds run over the packed rows) | the spill mark
};

__global__ __launch_bounds__(kBlock) void k_cut_settle(const CutSettleArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const Plan& p = a.p;
    const u32 m = p.m;
    const CsLds L = cs_lds(m);
    u32& nslot = *reinterpret_cast<u32*>(smem);
    u64* red = reinterpret_cast<u64*>(smem + 16);                             // [4] pending load, pending rows, rejected rows, rejected load
    unsigned short* slot = reinterpret_cast<unsigned short*>(smem + L.slot);  // [mr] slot of a node cut in this block (else 0xFFFF)
    unsigned short* node_of = reinterpret_cast<unsigned short*>(smem + L.node_of);
    const u32 kmax = L.kmax;
    u64* T = reinterpret_cast<u64*>(smem + L.grp);                            // [kmax][17]
    u64* g_rem = T + (size_t)kmax * 17;                                       // [kmax] budget left at the start of the cut wave
    u64* g_pre = g_rem + kmax;                                                // [kmax] admitted in the waves before it (this block)
    u64* g_acc = g_pre + kmax;                                                // [kmax] the cut wave's claim load so far
    u64* g_ssum = g_acc + kmax;                                               // [kmax] ... of the current step
    u64* g_adm = g_ssum + kmax;                                               // [kmax] admitted inside the cut wave
    u32* g_cw = reinterpret_cast<u32*>(g_adm + kmax);                         // [kmax] the cut wave (16: none)
    u32* g_row = g_cw + kmax;                                                 // [kmax] first rejected row
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 b = blockIdx.x;
    const u64 gw = (u64)b * kWaves + wave;
    RIOGP_KT(p, 0, 0);
    const u64 ncut = a.stats->n_cut;
    const u32 cb0 = a.cutblk[(u32)tid < m ? (u32)tid : m - 1];
    if (ncut == 0 && !a.pack) return;  // (k_cut_apply returned as well: k_scan's totals stand)
    u64 sp_sum = lane == 0 ? a.wsp_sum[gw] : 0ull;   // this range's totals so far (lane 0), the settle step adds per lane
    u32 sp_cnt = lane == 0 ? a.wsp_cnt[gw] : 0u;
    const u32 ul_n = a.ul.cnt[gw];
    u64 wstart, wend;
    wave_range_plain(p, gw, wstart, wend);
    // the first 64 entries of this range's list are requested now (clamped to the range's first word: always addressable), not
    // behind the two dependent round trips that locate the cut waves
    u32 po0, ro0, lo0, no0;
    {
        const u64 i0 = wstart + (u64)((u32)lane < ul_n ? (u32)lane : 0u);
        po0 = a.ul.pos[i0]; ro0 = a.ul.row[i0]; lo0 = a.ul.load[i0]; no0 = a.ul.node[i0];
    }
    if (tid == 0) nslot = 0;
    if (tid < 4) red[tid] = 0;
    __syncthreads();
    for (u32 j = tid; j < m; j += kBlock) {
        const u32 cb = j == (u32)tid ? cb0 : a.cutblk[j];
        u32 s = 0xFFFFu;
        if (cb == b) { s = atomicAdd(&nslot, 1u); node_of[s] = (unsigned short)j; }
        slot[j] = (unsigned short)s;
    }
    __syncthreads();
    const u32 ns = nslot;
    u64 rej_sum = 0;
    u32 rej_cnt = 0;
    const u32 mark = a.pack ? kNone : kSpillMark;
    if (ns) {  // (block-uniform)
        u32 g0 = 0, kn = 0;
        // one step = up to 64 undecided rows in index order, one per lane: {is one, node, load, row, packed position}
        auto settle = [&](const bool u, const u32 nd, const u32 lw, const u32 ix, const u64 pos) {
            const u32 sl = u ? (u32)slot[nd] - g0 : ~0u;
            u32 vd = 3;  // 0 admitted | 1 rejected | 2 the node's prefix crosses its budget in this step | 3 not of this group
            bool mine = false;
            if (sl < kn) {
                const u32 cw = g_cw[sl];
                vd = (u32)wave < cw ? 0u : 1u;
                mine = (u32)wave == cw;
            }
            if (__ballot(mine)) {  // rows whose node is cut in THIS wave
                if (mine) atomicAdd(&g_ssum[sl], (u64)lw);
                __builtin_amdgcn_wave_barrier();
                if (mine) {
                    const u64 base = g_acc[sl], tot = g_ssum[sl], rm = g_rem[sl];
                    vd = base > rm ? 1u : (base + tot <= rm ? 0u : 2u);
                }
                u64 cross = __ballot(vd == 2u);
                while (cross) {  // one ordered scan per node whose prefix crosses its budget in this step
                    const int fl = __ffsll((long long)cross) - 1;
                    const u32 s0 = (u32)__builtin_amdgcn_readlane((int)sl, fl);
                    const u64 base = g_acc[s0], rm = g_rem[s0];
                    const bool k = mine && sl == s0;
                    const u64 inc = wave_incl_scan(k ? (u64)lw : 0ull, lane);
                    if (k) vd = base + inc <= rm ? 0u : 1u;
                    cross = __ballot(vd == 2u);
                }
                __builtin_amdgcn_wave_barrier();
                if (mine) {
                    atomicAdd(&g_acc[sl], (u64)lw);
                    g_ssum[sl] = 0;
                    if (vd == 0u) atomicAdd(&g_adm[sl], (u64)lw);
                    else atomicMin(&g_row[sl], ix);
                }
                __builtin_amdgcn_wave_barrier();
            }
            // an admitted row's mark becomes inert (its node), a rejected row's the spill mark — and its real row, which still
            // holds the optimistic affinity, takes the mark of a row that goes on
            if (vd == 0u) {
                a.pk_next[pos] = nd;
            } else if (vd == 1u) {
                a.pk_next[pos] = kSpillMark;
                a.next[ix] = mark;
                sp_sum += (u64)lw; ++sp_cnt;
                rej_sum += (u64)lw; ++rej_cnt;
            }
        };
        for (g0 = 0; g0 < ns; g0 += kmax) {
            kn = ns - g0 < kmax ? ns - g0 : kmax;
            if (g0) __syncthreads();  // (the previous group's state is dead from here on)
            // the group's wave sums (k_cut_apply's atomics, complete since the launch boundary)
            for (u32 k = tid; k < kn * kWaves; k += kBlock) {
                const u32 ls = k / kWaves, w = k % kWaves;
                const u32 nd = node_of[g0 + ls];
                T[(size_t)ls * 17 + w] = a.Tg[(size_t)nd * kWaves + w];
                if (w == 0) g_rem[ls] = a.budget[nd];   // (the same round trip as the sums)
            }
            __syncthreads();
            for (u32 ls = tid; ls < kn; ls += kBlock) {  // the wave in which the node's claim prefix crosses the budget
                const u64 bud = g_rem[ls];
                const u64* Tj = T + (size_t)ls * 17;
                u64 acc = 0, pre = 0;
                u32 cw = kWaves;
#pragma unroll
                for (int w = 0; w < kWaves; ++w) {
                    const u64 nv = acc + Tj[w];
                    const bool hit = cw == (u32)kWaves && nv > bud;
                    pre = hit ? acc : pre;
                    cw = hit ? (u32)w : cw;
                    acc = nv;
                }
                if (cw == (u32)kWaves) pre = acc;  // (cannot happen: k_resolve found the block's prefix crossing the budget)
                g_cw[ls] = cw;
                g_pre[ls] = pre;
                g_rem[ls] = bud - pre;
                g_acc[ls] = 0; g_ssum[ls] = 0; g_adm[ls] = 0;
                g_row[ls] = kNoCut;
            }
            __syncthreads();
            RIOGP_KT(p, 0, 3);
            // this range's undecided rows, 64 a step, the next step's words in flight
            if (ul_n) {
                u32 po = po0, ro = ro0, lo = lo0, no = no0;
                for (u32 q0 = 0; q0 < ul_n; q0 += 64) {  // (wave-uniform)
                    const u32 q = q0 + (u32)lane;
                    const u32 pc = po, rc = ro, lc = lo, nc = no;
                    {
                        const u32 qn = q + 64 < ul_n ? q + 64 : (ul_n - 1);
                        const u64 i1 = wstart + qn;
                        po = a.ul.pos[i1]; ro = a.ul.row[i1]; lo = a.ul.load[i1]; no = a.ul.node[i1];
                    }
                    settle(q < ul_n, nc, lc, rc, wstart + pc);
                }
            }
            __syncthreads();
            for (u32 ls = tid; ls < kn; ls += kBlock) {
                const u32 nd = node_of[g0 + ls];
                a.cutidx[nd] = g_row[ls];
                a.used_cur[nd] = a.used_kept[nd] + a.admpre[nd] + g_pre[ls] + g_adm[ls];
            }
        }
    }
    RIOGP_KT(p, 0, 4);
    // ---- the range's and the block's final spill totals (the rounds' ordered prefix starts from these)
    sp_sum = wave_sum(sp_sum);
    sp_cnt = wave_sum32(sp_cnt);
    rej_sum = wave_sum(rej_sum);
    rej_cnt = wave_sum32(rej_cnt);
    if (lane == 0) {
        a.wsp_sum[gw] = sp_sum;
        a.wsp_cnt[gw] = sp_cnt;
        if (sp_cnt) { atomicAdd(&red[0], sp_sum); atomicAdd(&red[1], (u64)sp_cnt); }
        if (rej_cnt) { atomicAdd(&red[2], (u64)rej_cnt); atomicAdd(&red[3], rej_sum); }
    }
    __syncthreads();
    if (tid == 0) {
        if (red[2]) fx_add_rejected(a.fx, a.stats, red[2], red[3]);
        a.bsp_sum[b] = red[0];
        a.bsp_cnt[b] = (u32)red[1];
    }
    RIOGP_KT(p, 0, 7);
}

// committed `used` = U0 + the rounds' admitted loads (see k_fill); D rows are left as they are (k_resolve zeroes them)
__global__ void k_used_fold(u64* __restrict__ used, const u64* __restrict__ D, u32 m, u32 rounds) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    u64 u = used[j];
    for (u32 r = 0; r < rounds; ++r) u += D[(size_t)r * m + j];
    used[j] = u;
}

// ------------------------------------------------------------------------------------------------
// CRUD kernels over the assignment column (local.rs:22-68)
// ------------------------------------------------------------------------------------------------
__global__ void k_fill_u32(u32* p, u64 n, u32 v) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) p[i] = v;
}

// liveness push (rio_gp_set_alive*): the host packs the bitmap and hands it over BY VALUE in the kernel arguments
// (<= 1 KiB at 8 192 nodes) — no staging buffer whose lifetime would need a wait, no copy engine hand-off; the update
// is ordered on the handle's stream like any kernel and the call returns without synchronising.
__global__ void k_store_words(WordPack pack, u32 nwords, u32* __restrict__ dst) {
    const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nwords) dst[w] = pack.w[w];
}

__global__ void k_pack_alive(const uint8_t* alive, u32 m, u32* bits) {
    const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= (m + 31) / 32) return;
    u32 v = 0;
    for (u32 b = 0; b < 32; ++b) {
        const u32 j = w * 32 + b;
        if (j < m && alive[j]) v |= 1u << b;
    }
    bits[w] = v;
}

// Completion word of a single-workgroup call (micro-batches): every thread calls it after its last result store; the
// host spins on the mapped pinned word instead of going through hipStreamSynchronize (measured: 7.3 us instead of 12.6 us
// for launch + wait, tools/sync_probe.py).  done == nullptr: the caller waits for the stream.
__device__ __forceinline__ void signal_done(u32* done, u32 seq) {
    if (!done) return;
    __threadfence_system();  // this thread's result stores are on their way to host memory before ...
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // ... the word is
}

// The same for a call of several workgroups (medium batches): every workgroup fences its result stores and takes a ticket;
// the last one resets the ticket and stores the completion word.
__device__ __forceinline__ void signal_done_grid(unsigned int* ticket, u32* done, u32 seq) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Requests of the smallest calls (n <= 4: the reference's one-object-per-request flow) travel in the kernel arguments
// instead of mapped pinned memory: the kernel saves a PCIe read round trip (1.3 us of a 7-9 us call, tools/sync_probe.py).
__device__ __forceinline__ u32 inl_sel(const uint4 v, u32 k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }

// lookup, micro-batch (n <= kSmallBatch): one workgroup; requests inline (ninl = n <= 4) or in mapped pinned memory,
// results into mapped pinned memory, then the completion word
__device__ __forceinline__ void dev_lookup_small(const u32* __restrict__ assign, u64 n_obj, const u32* __restrict__ idx, u32 n,
                                                 u32* __restrict__ out, DevStats* st, u32 ninl, uint4 ia) {
    const u32 k = threadIdx.x;
    if (k < n) {
        const u32 i = ninl ? inl_sel(ia, k) : idx[k];
        if (i < n_obj) out[k] = assign[i];
        else { out[k] = kNone; atomicAdd(&st->err, 1ull); }
    }
}
__global__ __launch_bounds__(kSmallBatch) void k_lookup_small(const u32* __restrict__ assign, u64 n_obj,
                                                              const u32* __restrict__ idx, u32 n, u32* __restrict__ out,
                                                              DevStats* st, u32* done, u32 seq, u32 ninl, uint4 ia) {
    dev_lookup_small(assign, n_obj, idx, n, out, st, ninl, ia);
    signal_done(done, seq);
}

// lookup (local.rs:42-49): 12 B/lookup — idx read, assign gather, out write.  A lane takes FOUR consecutive lookups: one
// dwordx4 index read, four independent gathers in flight before the first is used, one dwordx4 store (the scalar form —
// one dependent 4-byte gather per lane and iteration — reached 40 % of the roofline on sequential indices and 7.7 % on
// random ones, where every gather pulls a whole 128-byte line through the fabric for 4 useful bytes).
__global__ __launch_bounds__(256) void k_lookup4(const u32* __restrict__ assign, u64 n_obj, const u32* __restrict__ idx, u64 n,
                                                 u32* __restrict__ out, DevStats* st, u32* done, u32 seq,
                                                 unsigned int* ticket) {
    const u64 nvec = n >> 2;
    u32 bad = 0;
    const u64 stride = (u64)gridDim.x * 256;
    u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    for (; v + stride < nvec; v += 2 * stride) {  // two vectors per trip: eight gathers in flight per lane
        const uint4 ia = *reinterpret_cast<const uint4*>(idx + 4 * v);
        const uint4 ib = *reinterpret_cast<const uint4*>(idx + 4 * (v + stride));
        uint4 ra, rb;
#define RIOGP_G(I, R) { const bool ok = I < n_obj; R = assign[ok ? I : 0]; R = ok ? R : kNone; bad += !ok; }
        RIOGP_G(ia.x, ra.x) RIOGP_G(ia.y, ra.y) RIOGP_G(ia.z, ra.z) RIOGP_G(ia.w, ra.w)
        RIOGP_G(ib.x, rb.x) RIOGP_G(ib.y, rb.y) RIOGP_G(ib.z, rb.z) RIOGP_G(ib.w, rb.w)
        *reinterpret_cast<uint4*>(out + 4 * v) = ra;
        *reinterpret_cast<uint4*>(out + 4 * (v + stride)) = rb;
    }
    for (; v < nvec; v += stride) {
        const uint4 ia = *reinterpret_cast<const uint4*>(idx + 4 * v);
        uint4 ra;
        RIOGP_G(ia.x, ra.x) RIOGP_G(ia.y, ra.y) RIOGP_G(ia.z, ra.z) RIOGP_G(ia.w, ra.w)
        *reinterpret_cast<uint4*>(out + 4 * v) = ra;
    }
    const u64 k = nvec * 4 + (u64)blockIdx.x * 256 + threadIdx.x;  // ragged tail (< 4 entries)
    if (k < n) {
        const u32 i = idx[k];
        u32 r;
        RIOGP_G(i, r)
        out[k] = r;
    }
#undef RIOGP_G
    if (bad) atomicAdd(&st->err, (u64)bad);
    if (ticket) signal_done_grid(ticket, done, seq);  // medium batches in mapped pinned memory: several workgroups
    else signal_done(done, seq);
}
// the same for index / result arrays that are not 16-byte aligned
__global__ void k_lookup(const u32* __restrict__ assign, u64 n_obj, const u32* __restrict__ idx, u64 n,
                         u32* __restrict__ out, DevStats* st, u32* done, u32 seq) {
    for (u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (u64)gridDim.x * blockDim.x) {
        const u32 i = idx[k];
        if (i < n_obj) out[k] = assign[i];
        else { out[k] = kNone; atomicAdd(&st->err, 1ull); }
    }
    signal_done(done, seq);
}

// update (local.rs:22-40), sequential last-writer-wins: phase 1 elects, per row, the highest
// batch position (atomicMin of the reversed position in a row-sized scratch), phase 2 lets the
// winner write and put its scratch slot back to all-ones.
__global__ void k_update_elect(u64 n_obj, u32 m, const u32* __restrict__ idx, const u32* __restrict__ node, u64 n,
                               u32* __restrict__ pos, DevStats* st) {
    for (u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (u64)gridDim.x * blockDim.x) {
        const u32 i = idx[k], nd = node[k];
        if (i < n_obj && (nd == kNone || nd < m)) atomicMin(&pos[i], (u32)(n - 1 - k));
        else atomicAdd(&st->err, 1ull);
    }
}
// the elected writer publishes and puts the scratch slot back to all-ones itself (a loser that reads the slot after
// that sees NONE != its own position and does nothing) — no third pass
// aff_life (row lifecycle, nullptr otherwise): the written row becomes an object whose affinity is its node, a deleted
// one (node NONE, local.rs:36-37) stops being one
// used / load (medium batches only, nullptr otherwise): the per-node load vector follows the write — the row's load leaves
// its old node and arrives on the new one (a handful of global atomics; big batches invalidate `used` instead)
__global__ void k_update_apply(u32* __restrict__ assign, u64 n_obj, u32 m, const u32* __restrict__ idx,
                               const u32* __restrict__ node, u64 n, u32* __restrict__ pos, u32* __restrict__ aff_life,
                               unsigned int* ticket, u32* done, u32 seq, u64* __restrict__ used,
                               const u32* __restrict__ load) {
    for (u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (u64)gridDim.x * blockDim.x) {
        const u32 i = idx[k], nd = node[k];
        if (i < n_obj && (nd == kNone || nd < m) && pos[i] == (u32)(n - 1 - k)) {
            if (used) {  // (one winner per row: nobody else touches assign[i] in this launch)
                const u32 old = assign[i], li = load[i];
                if (old != nd && li) {
                    if (old < m) atomicAdd(&used[old], (u64)0 - (u64)li);
                    if (nd < m) atomicAdd(&used[nd], (u64)li);
                }
            }
            assign[i] = nd;
            pos[i] = kNone;
            if (aff_life) aff_life[i] = nd == kNone ? kAffInactive : nd;
        }
    }
    if (done) signal_done_grid(ticket, done, seq);  // medium batches from mapped pinned memory: the host spins on the word
}
// update, micro-batch (n <= kSmallBatch): one workgroup, one launch; entries were validated by the host and may sit in
// mapped host memory.  Sequential last-writer-wins inside the batch: an entry loses to any LATER entry for the same row.
// (the body: every one of the workgroup's kSmallBatch threads calls it; hkey / hpos = 2 * kSmallBatch LDS words each)
__device__ __forceinline__ void dev_update_small(u32* __restrict__ assign, const u32* __restrict__ idx,
                                                 const u32* __restrict__ node, u32 n, u32* __restrict__ aff_life, u32 ninl,
                                                 uint4 ia, uint4 ib, u64* __restrict__ used, const u32* __restrict__ load,
                                                 u32 m, u32* hkey, u32* hpos) {
    // Last writer per row through a small open-addressing table in LDS (2 x kSmallBatch slots, linear probing): the row id
    // claims a slot with a compare-and-swap, the batch positions meet in an atomic max.  (Comparing every entry with every
    // later one, the first version, is 256 dependent LDS reads for the first entry of a full batch: 7.5 us of kernel time.)
    constexpr u32 kSlots = 2 * kSmallBatch;
    const u32 k = threadIdx.x;
    for (u32 q = k; q < kSlots; q += kSmallBatch) { hkey[q] = kNone; hpos[q] = 0; }
    u32 i = kNone, nd = kNone, old = kNone, li = 0;
    if (k < n) {
        i = ninl ? inl_sel(ia, k) : idx[k];
        nd = ninl ? inl_sel(ib, k) : node[k];
        if (used) { old = assign[i]; li = load[i]; }  // requested with the election, not after it: one round trip
    }
    __syncthreads();
    u32 slot = (i * 2654435761u) >> 23 & (kSlots - 1);  // (row ids are < 2^31: kNone never is one)
    if (k < n) {
        for (;;) {
            const u32 old = atomicCAS(&hkey[slot], kNone, i);
            if (old == kNone || old == i) break;
            slot = (slot + 1) & (kSlots - 1);
        }
        atomicMax(&hpos[slot], k + 1);
    }
    __syncthreads();
    if (k < n && hpos[slot] == k + 1) {  // this entry is the row's last one in the batch
        assign[i] = nd;
        if (aff_life) aff_life[i] = nd == kNone ? kAffInactive : nd;
        // `used` follows the write (round-2 advisor finding: invalidating it made the next place_pending re-stream the
        // whole table): every entry of the row read the same old node; the winner moves the row's load
        if (used && old != nd && li) {
            if (old < m) atomicAdd(&used[old], (u64)0 - (u64)li);
            if (nd < m) atomicAdd(&used[nd], (u64)li);
        }
    }
}
__global__ __launch_bounds__(kSmallBatch) void k_update_small(u32* __restrict__ assign, const u32* __restrict__ idx,
                                                              const u32* __restrict__ node, u32 n,
                                                              u32* __restrict__ aff_life, u32* done, u32 seq, u32 ninl,
                                                              uint4 ia, uint4 ib, u64* __restrict__ used,
                                                              const u32* __restrict__ load, u32 m) {
    __shared__ u32 hkey[2 * kSmallBatch];
    __shared__ u32 hpos[2 * kSmallBatch];
    dev_update_small(assign, idx, node, n, aff_life, ninl, ia, ib, used, load, m, hkey, hpos);
    signal_done(done, seq);  // the host may reuse the staging rows once the word is there
}

// remove, micro-batch (n <= kSmallBatch): one small workgroup, the released load goes straight to `used` (a handful of atomics;
// the big kernel's per-node LDS histogram — 1 024 threads, two passes over m words — was 12 us of a one-entry call)
__device__ __forceinline__ void dev_remove_small(u32* __restrict__ assign, u32 m, const u32* __restrict__ load,
                                                 const u32* __restrict__ idx, u32 n, u64* __restrict__ used,
                                                 u32* __restrict__ aff_life, u32 ninl, uint4 ia) {
    const u32 k = threadIdx.x;
    if (k < n) {
        const u32 i = ninl ? inl_sel(ia, k) : idx[k];
        const u32 li = used ? load[i] : 0u;
        const u32 old = atomicExch(&assign[i], kNone);  // duplicates inside the batch: only the first sees the node
        if (used && old < m) atomicAdd(&used[old], (u64)0 - (u64)li);
        if (aff_life) aff_life[i] = kAffInactive;  // row lifecycle: a removed key is no longer an object
    }
}
__global__ __launch_bounds__(kSmallBatch) void k_remove_small(u32* __restrict__ assign, u32 m, const u32* __restrict__ load,
                                                              const u32* __restrict__ idx, u32 n, u64* __restrict__ used,
                                                              u32* __restrict__ aff_life, u32* done, u32 seq, u32 ninl,
                                                              uint4 ia) {
    dev_remove_small(assign, m, load, idx, n, used, aff_life, ninl, ia);
    signal_done(done, seq);
}

// The update / remove / lookup parts of a mixed micro-batch (rio_gp_mixed_batch), each <= kSmallBatch validated entries, run
// in that order by ONE workgroup of kSmallBatch threads — alone (k_crud_small) or in front of the place_pending part
// (k_pp_one<kSmallBatch, 1>): the callers of one generation asked for different things, the device is asked once.  Between
// the parts the workgroup fences its stores at device scope and meets at a barrier: the next part reads what this one wrote.
struct CrudSmall {
    u32 nu, nr, nl;               // entries per part (0: the part is absent)
    u32 u_ninl, r_ninl, l_ninl;   // != 0: the part's entries ride in the kernel arguments (n <= 4)
    const u32 *u_idx, *u_node, *r_idx, *l_idx;
    u32* l_out;
    uint4 u_ia, u_ib, r_ia, l_ia;
    u64 n_obj;
    DevStats* st;
};
__device__ __forceinline__ void dev_crud_small(const CrudSmall& c, u32* __restrict__ assign, const u32* __restrict__ load, u32 m,
                                               u64* __restrict__ used, u32* __restrict__ aff_life, u32* hkey, u32* hpos,
                                               bool more_follows) {
    if (c.nu) {
        dev_update_small(assign, c.u_idx, c.u_node, c.nu, aff_life, c.u_ninl, c.u_ia, c.u_ib, used, load, m, hkey, hpos);
        if (c.nr | c.nl || more_follows) { __threadfence(); __syncthreads(); }
    }
    if (c.nr) {
        dev_remove_small(assign, m, load, c.r_idx, c.nr, used, aff_life, c.r_ninl, c.r_ia);
        if (c.nl || more_follows) { __threadfence(); __syncthreads(); }
    }
    if (c.nl) {
        dev_lookup_small(assign, c.n_obj, c.l_idx, c.nl, c.l_out, c.st, c.l_ninl, c.l_ia);
        if (more_follows) __syncthreads();  // (reads only: nothing to fence; the LDS tables are reused)
    }
}
__global__ __launch_bounds__(kSmallBatch) void k_crud_small(CrudSmall c, u32* __restrict__ assign, const u32* __restrict__ load,
                                                            u32 m, u64* __restrict__ used, u32* __restrict__ aff_life, u32* done,
                                                            u32 seq) {
    __shared__ u32 hkey[2 * kSmallBatch];
    __shared__ u32 hpos[2 * kSmallBatch];
    dev_crud_small(c, assign, load, m, used, aff_life, hkey, hpos, false);
    signal_done(done, seq);
}

// remove (local.rs:60-68): exchange makes duplicate removals of one row decrement `used` once.  The load released per
// node is gathered in an LDS histogram and flushed once per workgroup: a per-row global atomic on `used` serialises a
// million removals on at most m addresses (measured 55 us per million rows).
__global__ __launch_bounds__(kBlock) void k_remove(u32* __restrict__ assign, u64 n_obj, u32 m,
                                                   const u32* __restrict__ load, const u32* __restrict__ idx, u64 n,
                                                   u64* __restrict__ used, DevStats* st, u32* __restrict__ aff_life,
                                                   u32* done, u32 seq, u32 ninl, uint4 ia, unsigned int* ticket) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* rel = reinterpret_cast<u64*>(smem);  // [m] load released per node (only when `used` is maintained)
    if (used) {
        for (u32 j = threadIdx.x; j < m; j += kBlock) rel[j] = 0;
        __syncthreads();
    }
    u32 bad = 0;
    for (u64 k = (u64)blockIdx.x * kBlock + threadIdx.x; k < n; k += (u64)gridDim.x * kBlock) {
        const u32 i = ninl ? inl_sel(ia, (u32)k) : idx[k];
        if (i >= n_obj) { ++bad; continue; }
        const u32 li = used ? load[i] : 0u;  // requested with the exchange, not after it (one round trip, not two)
        const u32 old = atomicExch(&assign[i], kNone);
        if (used && old < m) atomicAdd(&rel[old], (u64)li);
        if (aff_life) aff_life[i] = kAffInactive;  // row lifecycle: a removed key is no longer an object
    }
    if (bad) atomicAdd(&st->err, (u64)bad);
    if (used) {
        __syncthreads();
        for (u32 j = threadIdx.x; j < m; j += kBlock)
            if (rel[j]) atomicAdd(&used[j], (u64)0 - rel[j]);
    }
    if (ticket) signal_done_grid(ticket, done, seq);
    else signal_done(done, seq);
}

// ------------------------------------------------------------------------------------------------
// Big random CRUD batches, partitioned by row window (update_batch / remove_batch of >= 2^18 entries).
//   A random entry of the plain kernels above touches three or four 128-byte lines of row-sized arrays (election word,
//   re-read, assignment, scratch reset) for 8 useful bytes: counters show 5x (elect) and 21x (apply) the algorithmic
//   bytes, 1.4 % of the HBM roofline on 10 M entries.  Here the batch is first SORTED BY ROW WINDOW in chunks:
//     k_part_bin    one workgroup per chunk of 8 192 consecutive entries (8 per lane, two dwordx4 per column): LDS
//                   histogram over the windows (W = 16 384 rows) with the returned count as the entry's rank, exclusive
//                   scan, the records {row in window | node code, batch position} go to their sorted place in an LDS
//                   staging buffer and leave as ONE coalesced copy — a first version scattered them straight to global
//                   memory, 611 write streams per workgroup: 88 us of partial-line writes for 10 M entries; the chunk's
//                   window boundaries go to a u16 table [window][chunk] (the transposed table — a row of its own per workgroup, whole lines — was
//                   measured in round 6: no difference);
//     k_part_update one workgroup per window: the window's piece of every chunk (~13 records each on a 10 M batch) is
//                   addressed through a prefix over the chunk table, one lane per record, and streams through an LDS
//                   table of W u64 words — last-writer-wins is a ds_max_u64 on {position + 1 | node code}; the window's rows
//                   are then written DENSELY, coalesced, once;
//     k_part_remove the same with a flag per row; the released load goes through an LDS histogram per node.
//   Batches of more than 2 048 chunks are applied slice by slice, in order.
// ------------------------------------------------------------------------------------------------
constexpr u32 kPartSub = 8192;        // entries per chunk = workgroup of k_part_bin: the small form (8 per lane) ...
constexpr u32 kPartSubBig = 16384;    // ... and the big one (16 per lane; round 6): a window's piece of a chunk is twice as long — ~27
                                      // records on a 10 M-row table instead of ~13, i.e. 1.6x instead of 2.2x the records' bytes in
                                      // 128-byte lines for the apply kernels' walk — and there are half as many pieces and descriptors.
                                      // The sorted chunk's staging buffer is then 128 KiB of LDS: tables of up to kPartBigMaxBins windows.
constexpr u32 kPartSliceMax = 2048u * kPartSub;  // entries per slice of the batch (16.7 M): 2 048 small / 1 024 big chunks
constexpr u32 kPartMaxChunks = 2048;  // (small) chunks per slice
constexpr u32 kPartShiftMax = 14;     // rows per window <= 16 384: W u64 election words = 128 KiB of LDS
constexpr u32 kPartRowMask = (1u << kPartShiftMax) - 1u;  // a record's row inside its window
constexpr u32 kPartCUs = 256;         // workgroups of one round: the apply kernels (one window each) and k_part_bin (one chunk each) run
                                      // one workgroup per CU
// Geometry of one partitioned batch (round 6: neither size has to be a power of two).  611 windows of 16 384 rows on 256 CUs are
// three rounds of workgroups, the last one 39 % full; 766 windows of 13 056 rows are three FULL rounds of workgroups that are each
// a fifth shorter.  The same for the chunks of the batch (768 chunks of 13 024 entries instead of 611 of 16 384).  The window of a
// row is one multiply-high and a shift: row * ceil(2^42 / W) >> 42, exact for rows < 2^27 and 2^12 <= W <= 2^14 (the magic number
// fits 32 bits; error term row / 2^42 < 2^-15 < 1 / W).
struct PartGeo {
    u32 W;       // rows per window (a multiple of 256, 4 096 .. 1 << kPartShiftMax)
    u32 sub;     // entries per chunk (a multiple of 4, <= 8 192 or 16 384: the PER of k_part_bin / the LP of the apply kernels)
    u32 wmagic;  // ceil(2^42 / W)
};
__host__ __device__ __forceinline__ u32 part_win_of(const PartGeo& pg, u32 row) { return (u32)(((u64)row * pg.wmagic) >> 42); }
constexpr u32 kPartMaxBins = 8192;    // 134 M rows at the largest window
constexpr u32 kNodeNoneCode = 0x3FFFu;  // 14-bit code of RIO_GP_NONE (node ids are < 8 192)

// records: updates {row in window | node code << 14, position in the slice} (8 bytes), removals the row in window (4 bytes)
template <bool UPDATE, int PER>
__global__ __launch_bounds__(kBlock) void k_part_bin(u64 n_obj, u32 m, const u32* __restrict__ idx,
                                                     const u32* __restrict__ node, u64 n, u32 nbins, const PartGeo pg,
                                                     u32* __restrict__ rec, uint2* __restrict__ rec2,
                                                     unsigned short* __restrict__ start16, DevStats* st, const bool none_ok,
                                                     u32* __restrict__ host_err = nullptr, u32* __restrict__ zero_flags = nullptr,
                                                     u32* __restrict__ zero_bits = nullptr, const u32 zero_words = 0,
                                                     u64* __restrict__ zero_u64 = nullptr, const u32 zero_u64_words = 0) {
    // host_err (optional, mapped host memory): set when the chunk holds an invalid entry, so the caller learns it without a
    // copy-back — the kernels it has enqueued behind this one look at st->err and do nothing
    // zero_flags / zero_bits (place_pending): the per-request flag column (this chunk's slice of it, densely) and the bitmap
    // of dead nodes the requests run into are cleared here — two memset launches less in front of the window kernel
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* part = reinterpret_cast<u64*>(smem);                 // [16] block-scan partials
    u32* hist = reinterpret_cast<u32*>(smem + kSmall);        // [nbins] entries of this chunk per window
    u32* off = hist + nbins;                                  // [nbins + 1] where a window's records start in the sorted chunk
    unsigned char* stage = smem + kSmall + (((size_t)2 * nbins + 1) * sizeof(u32) + 15) / 16 * 16;  // [pg.sub] records
    const int tid = threadIdx.x;
    constexpr int NV = PER / 4;             // dwordx4 loads per column and lane (pg.sub <= PER * 1 024 entries in this chunk)
    const u32 nchunks = gridDim.x, c = blockIdx.x;
    const u64 lo = (u64)c * pg.sub;
    const u64 hi = lo + pg.sub < n ? lo + pg.sub : n;
    // the chunk's columns through a UNIFORM base (idx + lo) and 32-bit in-chunk offsets: one offset register per vector, shared by
    // both columns — 64-bit per-lane addresses for the eight vectors of the 16-per-lane form were 16 registers of its spill
    const u32 cn = (u32)(hi - lo);  // entries of this chunk
    auto load4 = [&](const u32* cb, u32 o) -> uint4 {  // entries o..o+3 of the chunk's column, zero past its end (o is a multiple of 4)
        if (o + 4u <= cn) return *reinterpret_cast<const uint4*>(cb + o);
        uint4 r = make_uint4(0, 0, 0, 0);
        if (o + 0u < cn) r.x = cb[o + 0u];
        if (o + 1u < cn) r.y = cb[o + 1u];
        if (o + 2u < cn) r.z = cb[o + 2u];
        return r;
    };
    const u32* const ib = idx + lo;  // vector q of this lane: entries tid * 4 + q * 4 096 .. + 3 of the chunk
    const u32* const nb = UPDATE ? node + lo : nullptr;
    uint4 iv[NV], nv[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        iv[q] = load4(ib, (u32)tid * 4u + (u32)q * (kBlock * 4u));
        nv[q] = UPDATE ? load4(nb, (u32)tid * 4u + (u32)q * (kBlock * 4u)) : make_uint4(0, 0, 0, 0);
    }
    for (u32 b = tid; b < nbins; b += kBlock) hist[b] = 0;
    if (zero_flags) {
        u32* const zb = zero_flags + lo;  // (uniform base, 32-bit in-chunk offsets: as the loads above)
        const bool al = (reinterpret_cast<uintptr_t>(zb) & 15u) == 0;
        for (int q = 0; q < NV; ++q) {
            const u32 o = (u32)tid * 4u + (u32)q * (kBlock * 4u);
            if (al && o + 4u <= cn) *reinterpret_cast<uint4*>(zb + o) = make_uint4(0, 0, 0, 0);
            else
                for (u32 e = 0; e < 4 && o + e < cn; ++e) zb[o + e] = 0;
        }
    }
    if (zero_bits && c == 0)
        for (u32 w = tid; w < zero_words; w += kBlock) zero_bits[w] = 0;
    if (zero_u64 && c == (nchunks > 1 ? 1u : 0u))  // (place_pending: the per-requester claim loads + the window kernel's counter)
        for (u32 w = tid; w < zero_u64_words; w += kBlock) zero_u64[w] = 0;
    __syncthreads();
    u32 I[PER], N[PER];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        I[4 * q] = iv[q].x; I[4 * q + 1] = iv[q].y; I[4 * q + 2] = iv[q].z; I[4 * q + 3] = iv[q].w;
        N[4 * q] = nv[q].x; N[4 * q + 1] = nv[q].y; N[4 * q + 2] = nv[q].z; N[4 * q + 3] = nv[q].w;
    }
    // Counting and placing are two rounds of LDS atomics (round 6): the count needs no return value, and the entry's place in the
    // sorted chunk is the returned value of a second atomic on the window's running offset, after the scan.  (One returning atomic
    // — count = rank — before: sixteen ranks in registers across the scan next to the sixteen entries spilled 76 bytes a lane in
    // the 16-per-lane form.)  The order of a window's entries inside the chunk is whatever the atomics make it, as before.
    u32 okmask = 0;  // bit j: entry j of this lane is valid
    u32 c2[PER / 2];  // the entries' 14-bit node codes, two to a register (what is left of the node column after this loop)
    u32 bad = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const bool in = (u32)tid * 4u + (u32)(j >> 2) * (kBlock * 4u) + (u32)(j & 3) < (u32)(hi - lo);  // (32-bit: in-chunk offsets)
        const bool ok = in && I[j] < (u32)n_obj && (!UPDATE || (none_ok && N[j] == kNone) || N[j] < m);  // (n_obj <= 2^27 here)
        if (ok) { atomicAdd(&hist[part_win_of(pg, I[j])], 1u); okmask |= 1u << j; }
        bad += in && !ok;
        const u32 code = UPDATE ? (N[j] == kNone ? kNodeNoneCode : (N[j] & kNodeNoneCode)) : 0u;
        if (j & 1) c2[j >> 1] |= code << 16; else c2[j >> 1] = code;
    }
    if (bad) {
        atomicAdd(&st->err, (u64)bad);
        if (host_err) __hip_atomic_store(host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    {   // exclusive scan over the windows: up to 8 per thread
        const u32 per_t = (nbins + kBlock - 1) / kBlock;
        u32 v[8];
        u64 loc = 0;
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
            const u32 b = tid * per_t + q;
            v[q] = (q < per_t && b < nbins) ? hist[b] : 0u;
            loc += v[q];
        }
        u64 total = 0;
        u64 ex = block_excl_scan_1024(loc, false, part, &total);
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
            const u32 b = tid * per_t + q;
            if (q < per_t && b < nbins) {
                off[b] = (u32)ex;
                start16[(size_t)b * nchunks + c] = (unsigned short)ex;
            }
            ex += v[q];
        }
        if (tid == 0) {
            off[nbins] = (u32)total;
            start16[(size_t)nbins * nchunks + c] = (unsigned short)total;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        if ((okmask >> j) & 1u) {
            u32 ij = I[j];
            asm volatile("" : "+v"(ij));  // the window is computed AGAIN (one multiply): kept from the counting loop, sixteen of them spill
            const u32 wn = part_win_of(pg, ij);
            const u32 pos = atomicAdd(&off[wn], 1u);
            const u32 code = (c2[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
            const u32 r = (ij - wn * pg.W) | (code << kPartShiftMax);
            const u32 k = (u32)lo + (u32)tid * 4u + (u32)(j >> 2) * (kBlock * 4u) + (u32)(j & 3);  // position in the slice
            if (UPDATE) reinterpret_cast<uint2*>(stage)[pos] = make_uint2(r, k);
            else reinterpret_cast<u32*>(stage)[pos] = r;
        }
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // four entries at a time: all sixteen interleaved do not fit the register file
    }
    __syncthreads();
    const u32 total = off[nbins];
    for (u32 t = tid; t < total; t += kBlock) {  // the sorted chunk leaves in one coalesced copy
        if (UPDATE) rec2[lo + t] = reinterpret_cast<const uint2*>(stage)[t];
        else rec[lo + t] = reinterpret_cast<const u32*>(stage)[t];
    }
}

// The apply kernels walk the window's piece of every chunk (~13 records of a small chunk on a 10 M-row table, ~27 of a big
// one): LP lanes per piece — a QUARTER wave (16) for small chunks, a HALF wave (32) for big ones — so a wave instruction covers
// 64 / LP pieces; lane group g of wave w takes the chunks (1024 / LP) i + (64 / LP) w + g.  Every descriptor (two u16
// reads) is requested before the first record, records go out four pieces per lane at a time.  LP * 512 = the chunk's size.
constexpr int kPartIters = kPartMaxChunks / 64;  // 32 pieces per lane group at most (2 048 small or 1 024 big chunks a slice)
constexpr int kPartFlight = 8;   // pieces per lane requested before the first is used (8 was measured: no gain — the loop is bound by
                                 // the partial-line reads of the ~13-record pieces, not by its round trips)
constexpr int kPartRowVecs = (1 << kPartShiftMax) / (kBlock * 4);  // 16-byte vectors of the window's rows per thread (4)

#define RIOGP_PART_DESCRIPTORS()                                                                          \
    u32 pbase[kPartIters], pcnt[kPartIters];                                                              \
    _Pragma("unroll") for (int i = 0; i < kPartIters; ++i) {                                              \
        const u32 f = (u32)i * (1024u / LP) + (u32)wave * (64u / LP) + (u32)lane / LP;                    \
        const bool in = f < nchunks;                                                                      \
        const u32 fc = in ? f : 0u;  /* clamped, not predicated: a load behind a branch makes every later */ \
        const u32 s0 = start16[(size_t)b * nchunks + fc];           /* wait a wait for ALL loads (LESSONS 1): */ \
        const u32 s1 = start16[(size_t)(b + 1) * nchunks + fc];     /* 32 round trips in a row, found in round 6 */ \
        pbase[i] = f * pg.sub + s0;                                                                       \
        pcnt[i] = in ? s1 - s0 : 0u;                                                                      \
    }

template <u32 LP>
__global__ __launch_bounds__(kBlock) void k_part_update(u32* __restrict__ assign, u64 n_obj, const uint2* __restrict__ rec2,
                                                        const unsigned short* __restrict__ start16, u32 nchunks,
                                                        u32* __restrict__ aff_life, const PartGeo pg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 W = pg.W;
    u64* win = reinterpret_cast<u64*>(smem);  // [W] {position + 1 | node code} of the last writer, 0 = untouched
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 b = blockIdx.x, o16 = (u32)lane & (LP - 1u);
    const u64 base = (u64)b * W;
    RIOGP_PART_DESCRIPTORS()
    for (u32 r = tid; r < W; r += kBlock) win[r] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kPartIters; i += kPartFlight) {  // kPartFlight pieces per lane in flight: the loop is a chain of round trips
        if ((u32)i * (1024u / LP) >= nchunks) break;  // (uniform: no chunk this far — the clamped loads below would be dummies)
        uint2 x[kPartFlight];
#pragma unroll
        for (int q = 0; q < kPartFlight; ++q) x[q] = rec2[o16 < pcnt[i + q] ? pbase[i + q] + o16 : 0u];  // (clamped: record 0 exists)
#pragma unroll
        for (int q = 0; q < kPartFlight; ++q)
            if (o16 < pcnt[i + q]) atomicMax(&win[x[q].x & kPartRowMask], ((u64)(x[q].y + 1u) << 16) | (u64)(x[q].x >> kPartShiftMax));
    }
    // this thread's rows of the window as they are now (the columns are padded to whole tiles): requested here, behind the
    // main loop (its registers are free again) and ahead of the tail loop and the barrier
    uint4 curv[kPartRowVecs];
#pragma unroll
    for (int q = 0; q < kPartRowVecs; ++q) {
        const u32 r4 = ((u32)q * kBlock + (u32)tid) * 4u;
        curv[q] = (r4 < W && base + r4 < n_obj) ? *reinterpret_cast<const uint4*>(assign + base + r4) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll 1
    for (int i = 0; i < kPartIters; ++i)  // pieces of more than LP records
        for (u32 o = LP + o16; o < pcnt[i]; o += LP) {
            const uint2 x = rec2[pbase[i] + o];
            atomicMax(&win[x.x & kPartRowMask], ((u64)(x.y + 1u) << 16) | (u64)(x.x >> kPartShiftMax));
        }
    __syncthreads();
    // The window's rows leave as whole 16-byte vectors merged with their current values (requested at the top of the kernel),
    // every lane of a wave that changes anything: full 128-byte lines.  Masked 4-byte stores — 63 % of the rows of a 10 M
    // batch over 10 M rows — touch nearly every 32-byte sector without filling it, a read-modify-write at the memory side.
#pragma unroll
    for (int q = 0; q < kPartRowVecs; ++q) {
        const u32 r4 = ((u32)q * kBlock + (u32)tid) * 4u;
        if (r4 >= W || base + r4 >= n_obj) continue;
        const u64 v0 = win[r4], v1 = win[r4 + 1], v2 = win[r4 + 2], v3 = win[r4 + 3];
        uint4 o = curv[q];
#define RIOGP_MERGE(V, O)                                                                   \
        if (V) { const u32 code = (u32)V & 0xFFFFu; O = code == kNodeNoneCode ? kNone : code; }
        RIOGP_MERGE(v0, o.x) RIOGP_MERGE(v1, o.y) RIOGP_MERGE(v2, o.z) RIOGP_MERGE(v3, o.w)
#undef RIOGP_MERGE
        if (__ballot((v0 | v1 | v2 | v3) != 0)) *reinterpret_cast<uint4*>(assign + base + r4) = o;
        if (aff_life) {
            if (v0) aff_life[base + r4 + 0] = o.x == kNone ? kAffInactive : o.x;
            if (v1) aff_life[base + r4 + 1] = o.y == kNone ? kAffInactive : o.y;
            if (v2) aff_life[base + r4 + 2] = o.z == kNone ? kAffInactive : o.z;
            if (v3) aff_life[base + r4 + 3] = o.w == kNone ? kAffInactive : o.w;
        }
    }
}

template <u32 LP>
__global__ __launch_bounds__(kBlock) void k_part_remove(u32* __restrict__ assign, u64 n_obj, u32 m,
                                                        const u32* __restrict__ load, const u32* __restrict__ rec,
                                                        const unsigned short* __restrict__ start16, u32 nchunks,
                                                        u64* __restrict__ used, u32* __restrict__ aff_life, const PartGeo pg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 W = pg.W;
    u32* flag = reinterpret_cast<u32*>(smem);               // [W] row of this window is in the batch
    u64* rel = reinterpret_cast<u64*>(flag + W);            // [m] load released per node (when `used` is maintained)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 b = blockIdx.x, o16 = (u32)lane & (LP - 1u);
    RIOGP_PART_DESCRIPTORS()
    for (u32 r = tid; r < W; r += kBlock) flag[r] = 0;
    if (used)
        for (u32 j = tid; j < m; j += kBlock) rel[j] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kPartIters; i += kPartFlight) {
        if ((u32)i * (1024u / LP) >= nchunks) break;  // (uniform: no chunk this far)
        u32 x[kPartFlight];
#pragma unroll
        for (int q = 0; q < kPartFlight; ++q) x[q] = rec[o16 < pcnt[i + q] ? pbase[i + q] + o16 : 0u];  // (clamped: record 0 exists)
#pragma unroll
        for (int q = 0; q < kPartFlight; ++q)
            if (o16 < pcnt[i + q]) flag[x[q] & kPartRowMask] = 1u;  // duplicates: the same store
    }
    const u64 base = (u64)b * W;
    uint4 curv[kPartRowVecs];  // the window's rows as they are now (as in k_part_update: whole vectors go back)
#pragma unroll
    for (int q = 0; q < kPartRowVecs; ++q) {
        const u32 r4 = ((u32)q * kBlock + (u32)tid) * 4u;
        curv[q] = (r4 < W && base + r4 < n_obj) ? *reinterpret_cast<const uint4*>(assign + base + r4) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll 1
    for (int i = 0; i < kPartIters; ++i)
        for (u32 o = LP + o16; o < pcnt[i]; o += LP) flag[rec[pbase[i] + o] & kPartRowMask] = 1u;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kPartRowVecs; ++q) {
        const u32 r4 = ((u32)q * kBlock + (u32)tid) * 4u;
        if (r4 >= W || base + r4 >= n_obj) continue;
        const uint4 f = *reinterpret_cast<const uint4*>(flag + r4);
        uint4 o = curv[q];
        // (rows past n_obj inside the vector are never flagged: the entries were validated)
#define RIOGP_RM(F, O, E)                                                                                   \
        if (F) {                                                                                            \
            if (O != kNone) {                                                                               \
                if (used && O < m) atomicAdd(&rel[O], (u64)load[base + r4 + E]);                            \
                O = kNone;                                                                                  \
            }                                                                                               \
            if (aff_life) aff_life[base + r4 + E] = kAffInactive;  /* row lifecycle: no longer an object */ \
        }
        RIOGP_RM(f.x, o.x, 0) RIOGP_RM(f.y, o.y, 1) RIOGP_RM(f.z, o.z, 2) RIOGP_RM(f.w, o.w, 3)
#undef RIOGP_RM
        if (__ballot((f.x | f.y | f.z | f.w) != 0)) *reinterpret_cast<uint4*>(assign + base + r4) = o;
    }
    if (used) {
        __syncthreads();
        for (u32 j = tid; j < m; j += kBlock)
            if (rel[j]) atomicAdd(&used[j], (u64)0 - rel[j]);
    }
}

// ------------------------------------------------------------------------------------------------
// place_pending, big batches (>= 2^18 requests), partitioned by row window like the CRUD batches above.  A request of the
// plain kernels (k_pp_mark_dead / elect / gather / scatter / output) makes eight or nine random 4-byte accesses to row-sized
// arrays — each pulls a 128-byte line through the fabric: 6e9 requests/s for 10 M requests, 2 % of the roofline.  Here the
// requests are sorted by row window once (k_part_bin, records {row in window | requester, batch position}) and every
// row-side step works out of LDS, one workgroup per window:
//   k_pp_win_gather  first request per row (ds_min on the batch position), the row's node and load (reads inside the window's
//                    64 KB of each column), a requested object on a dead node marks that node for clean_server and counts as
//                    pending; the virtual-table row {cur | load} — for a later request of the same object {skip | position of
//                    the first} — leaves as ONE 8-byte store at the request's batch position (the only random access of this
//                    step); the window's rows are read once, densely, and a pending row whose first requester is alive gets
//                    that requester written back in the same pass: its optimistic placement;
//   (solve)          the same kernels as every solve, over the virtual table in batch-position order — the requesters column
//                    is the caller's own array; k_scan<COMPACT 3> reads the 8-byte records, leaves them as the two columns the
//                    kernels behind it read, and the fix-up writes what it changes into the real assignment column through the
//                    caller's object column (a first touch on a requester that is not alive — REF_SELF_ASSIGN — is stored by
//                    the scan: clean_server runs between the window kernel and the solve);
//   k_pp_win_output  in batch-position order, dense: a first request's answer is its virtual row's `next`, a later request
//                    reads the first's (one random read per duplicate).
// Two random accesses per request instead of nine.
// ------------------------------------------------------------------------------------------------
// fast[0] = requests this kernel could not answer by itself (an object on a dead node: clean_server first; a pending object
// whose requester is not an active member: water-fill, or the reference's unconditional self-assignment) — the call then runs
// the solve over the records; claim[m] = load the first touches put on every requester (k_pp_win_verdict checks it against
// the free capacity).  Both zeroed by the binning kernel.
template <u32 LP, int kKeep>
__global__ __launch_bounds__(kBlock) void k_pp_win_gather(u32* __restrict__ assign, const u32* __restrict__ load, u64 n_obj,
                                                          u32 m, const u32* __restrict__ alive_bits,
                                                          const uint2* __restrict__ rec2, const unsigned short* __restrict__ start16,
                                                          u32 nchunks, const PartGeo pg, u32* __restrict__ ans0, u32* __restrict__ ans1,
                                                          u32* __restrict__ dead_bits,
                                                          u32* __restrict__ aff_life, const DevStats* __restrict__ st,
                                                          u64* __restrict__ claim, u64* __restrict__ fast, const u32 lds_hist,
                                                          const u32 trace) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (st->err) return;  // the binning kernel found an invalid entry: the call fails, nothing is touched
    RIOGP_KTF(trace, 7, 0);  // (lab build: phase boundaries of the first 256 windows' workgroups, trace table 7)
    const u32 W = pg.W;
    u64* wfirst = reinterpret_cast<u64*>(smem);  // [W] {first batch position that asks for the row | its requester}, then {.. | the row's node}
    u64* hist = wfirst + W;                      // [m] claim load per requester (lds_hist; else straight into `claim`)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 b = blockIdx.x, o16 = (u32)lane & (LP - 1u);
    const u64 base = (u64)b * W;
    for (u32 r = tid; r < W; r += kBlock) wfirst[r] = ~0ull;
    if (lds_hist)
        for (u32 j = tid; j < m; j += kBlock) hist[j] = 0;
    __syncthreads();
    RIOGP_KTF(trace, 7, 1);
    // The first request of every row decides: ds_min_u64 on {position | requester} (one walk gives both).  The records this
    // lane reads — one per piece, up to kPartIters of them — STAY IN REGISTERS for the answers of the later requests further
    // down: a second walk over the sorted records (what round 4 did) reads the 80 MB again, in ~13-record pieces, 50 of this
    // kernel's 230 us at 10 M requests.  Only the records past a piece's first sixteen (a fifth of the pieces have some) are
    // read twice.
    // ... the first kKeep pieces' records, that is: all 32 are 64 registers next to everything the rows pass needs, and the
    // kernel spilled ~100 bytes a lane into the middle of its walks (round 6: the spill reloads sat between the LDS atomics, each
    // with a wait for every load in flight).  The pieces past kKeep (batches of more than kKeep / 32 of a slice: > 10.4 M
    // requests at 768 chunks of a slice) are read again by the answer walk.
    // (kKeep = 24, or 4 for batches whose chunks fit four steps — up to 2 M requests: the unrolled steps past the last chunk are
    // dummy loads, and a uniform exit from the unrolled loops cost the 10 M batch 90 us when it was tried)
    uint2 xr[kKeep];
    u32 tailmask = 0;  // bit i: piece i of this lane group holds more than LP records
    // Where piece i's records start (and its answers go) is kept ACROSS THE LANES of the group: lane o holds the start of the
    // pieces o, o + LP, ... — one or two registers instead of 32, and the answer walk needs no descriptor round trips (round 6:
    // 4 groups x ~2 us of its 22 us per workgroup).
    u32 pbv[kPartIters / LP];
#pragma unroll
    for (u32 e = 0; e < kPartIters / LP; ++e) pbv[e] = 0;
    constexpr int kGrp = 4;  // pieces per lane whose descriptors, then records, are in flight together (the descriptors die with
                             // their group: all 32 next to the 32 records would not fit the register file)
    auto piece = [&](u32 i, u32& pb, u32& pc) {  // piece i of this lane group: first record and record count
        const u32 f = i * (1024u / LP) + (u32)wave * (64u / LP) + (u32)lane / LP;
        const bool in = f < nchunks;
        const u32 fc = in ? f : 0u;  // (clamped, not predicated: see RIOGP_PART_DESCRIPTORS)
        const u32 s0 = start16[(size_t)b * nchunks + fc];
        const u32 s1 = start16[(size_t)(b + 1) * nchunks + fc];
        pb = f * pg.sub + s0;
        pc = in ? s1 - s0 : 0u;
    };
#pragma unroll
    for (int i = 0; i < kKeep; i += kGrp) {
        u32 pb[kGrp], pc[kGrp];
#pragma unroll
        for (int q = 0; q < kGrp; ++q) piece((u32)(i + q), pb[q], pc[q]);
#pragma unroll
        for (int q = 0; q < kGrp; ++q) {
            xr[i + q] = rec2[o16 < pc[q] ? pb[q] + o16 : 0u];  // (clamped: record 0 exists)
            if (!(o16 < pc[q])) xr[i + q] = make_uint2(0u, kNone);
            tailmask |= (pc[q] > LP ? 1u : 0u) << (i + q);
            if (o16 == (u32)(i + q) % LP) pbv[(u32)(i + q) / LP] = pb[q];
        }
#pragma unroll
        for (int q = 0; q < kGrp; ++q)
            if (xr[i + q].y != kNone) atomicMin(&wfirst[xr[i + q].x & kPartRowMask], ((u64)xr[i + q].y << 32) | (u64)(xr[i + q].x >> kPartShiftMax));
    }
#pragma unroll 1
    for (u32 i = kKeep; i < (u32)kPartIters; i += kGrp) {  // the pieces whose records are not kept
        if (i * (1024u / LP) >= nchunks) break;  // (uniform: no chunk this far)
        u32 pb[kGrp], pc[kGrp];
        uint2 x[kGrp];
#pragma unroll
        for (int q = 0; q < kGrp; ++q) piece(i + (u32)q, pb[q], pc[q]);
#pragma unroll
        for (int q = 0; q < kGrp; ++q) {
            x[q] = rec2[o16 < pc[q] ? pb[q] + o16 : 0u];
            tailmask |= (pc[q] > LP ? 1u : 0u) << (i + (u32)q);
        }
#pragma unroll
        for (int q = 0; q < kGrp; ++q)
            if (o16 < pc[q]) atomicMin(&wfirst[x[q].x & kPartRowMask], ((u64)x[q].y << 32) | (u64)(x[q].x >> kPartShiftMax));
    }
    RIOGP_KTF(trace, 7, 2);
    // the pieces' records past their first LP (an eighth of the big chunks' pieces on a 10 M-row table): only the pieces some
    // lane group of this wave flagged are looked at again — walking all 32 descriptors of every lane for them was 9 + 12 us of
    // a workgroup's 66
#pragma unroll 1
    for (u32 i = 0; i < (u32)kPartIters; ++i) {
        if (!__ballot((tailmask >> i) & 1u)) continue;  // (wave-uniform)
        u32 pb, pc;
        piece(i, pb, pc);
        for (u32 o = LP + o16; o < pc; o += LP) {
            const uint2 x = rec2[pb + o];
            atomicMin(&wfirst[x.x & kPartRowMask], ((u64)x.y << 32) | (u64)(x.x >> kPartShiftMax));
        }
    }
    // the window's rows: vector q of this thread is requested one step ahead of its use, the first one here, ahead of the barrier
    // (the addresses depend on nothing the walk produced; a vector outside the window or the table reads the window's first)
    auto rows_at = [&](int q) -> u64 {
        const u32 r4 = ((u32)q * kBlock + (u32)tid) * 4u;
        return base + ((r4 < W && base + r4 < n_obj) ? r4 : 0u);
    };
    uint4 cvn = *reinterpret_cast<const uint4*>(assign + rows_at(0));
    uint4 lvn = *reinterpret_cast<const uint4*>(load + rows_at(0));
    __syncthreads();
    RIOGP_KTF(trace, 7, 3);
    // The requested rows of the window, in row order, read ONCE and densely.  Each gets its answer: sticky (the node it is on),
    // or — pending: unplaced, or found on a dead node — its first requester when that one is an active member: written into
    // the real column right here, densely, as its placement (what the solve's scan used to do with one scattered 4-byte
    // store per first touch: 130 of its 174 us at 10 M requests).  If the batch turns out to need the solve (k_pp_win_verdict)
    // that placement is the optimistic one the fix-up overwrites through the same row when the claim is rejected; and
    // clean_server (which runs behind this kernel for the dead nodes the requests ran into) cannot take it for a row of a dead
    // node: its value is a live node now.  The row's answer stays in the LDS word; the records are written in the walk below.
    u32 slow = 0;
#pragma unroll
    for (int q = 0; q < kPartRowVecs; ++q) {
        const u32 r4 = ((u32)q * kBlock + (u32)tid) * 4u;
        const uint4 cv = cvn, lv = lvn;
        if (q + 1 < kPartRowVecs) {
            cvn = *reinterpret_cast<const uint4*>(assign + rows_at(q + 1));
            lvn = *reinterpret_cast<const uint4*>(load + rows_at(q + 1));
        }
        if (r4 >= W || base + r4 >= n_obj) continue;
        const u64 e0 = wfirst[r4], e1 = wfirst[r4 + 1], e2 = wfirst[r4 + 2], e3 = wfirst[r4 + 3];
        if (!__ballot((e0 & e1 & e2 & e3) != ~0ull)) continue;  // nobody asks for any of the wave's 256 rows
        uint4 ov = cv;
        bool chg = false;
#define RIOGP_FIRST(EW, C, L, O, E)                                                                                \
        if (EW != ~0ull) {                                                                                         \
            const u32 K = (u32)(EW >> 32), rq = (u32)EW;                                                           \
            const bool dead = C < m && !bit_of(alive_bits, C);  /* service.rs:227-237: clean_server of that node */ \
            u32 fl, nd;                                                                                            \
            if (dead) {  /* (RIO_GP_FLAG_REPLACED travels in the record's flag byte: k_pp_win_output reads it there) */ \
                atomicOr(&dead_bits[C >> 5], 1u << (C & 31));                                                      \
                ++slow;                                                                                            \
            }                                                                                                      \
            /* row lifecycle: an object from its first request on, home = the requester (a row on a dead node: once */ \
            /* clean_server has taken it out, k_pp_win_output) */                                                  \
            if (aff_life && C >= m) aff_life[base + r4 + E] = rq;                                                  \
            if (C < m && !dead) { nd = C; fl = C == rq ? 0u : 1u; }         /* sticky: LOCAL | REDIRECT */          \
            else if (rq < m && bit_of(alive_bits, rq)) {                    /* first touch on the requester: PLACED */ \
                nd = rq; fl = 2u | (dead ? kFlagReplaced : 0u);                                                    \
                O = rq; chg = true;                                                                                \
                if (lds_hist) atomicAdd(&hist[rq], (u64)L); else if (L) atomicAdd(&claim[rq], (u64)L);             \
            } else { nd = kNone; fl = 4u | (dead ? kFlagReplaced : 0u); ++slow; }  /* the solve's to decide */       \
            /* what the row's requests are answered from below: {first position, 24 bits | flag, 8 | load, 16 | node, 16}: */ \
            /* a load of 65 535 or more is looked up in the column by the one request that needs it */            \
            wfirst[r4 + E] = ((u64)K << 40) | ((u64)fl << 32) | ((u64)(L < 0xFFFFu ? L : 0xFFFFu) << 16) |         \
                             (nd == kNone ? 0xFFFFull : (u64)nd);                                                  \
        }
        RIOGP_FIRST(e0, cv.x, lv.x, ov.x, 0)
        RIOGP_FIRST(e1, cv.y, lv.y, ov.y, 1)
        RIOGP_FIRST(e2, cv.z, lv.z, ov.z, 2)
        RIOGP_FIRST(e3, cv.w, lv.w, ov.w, 3)
#undef RIOGP_FIRST
        if (__ballot(chg)) *reinterpret_cast<uint4*>(assign + base + r4) = ov;  // (whole lines; unchanged rows keep their value)
    }
    __syncthreads();
    RIOGP_KTF(trace, 7, 4);
    // Every request's answer record {node | flag | later, load or the first request's position}, written AT THE REQUEST'S OWN
    // PLACE IN THE SORTED ORDER (two 4-byte columns): the sixteen lanes of a quarter wave store sixteen consecutive words of a
    // piece — where this kernel used to store one 8-byte record per request at its BATCH position, 10 M random stores, 2.35x the
    // bytes, two thirds of its time.  k_pp_win_unsort carries them back to batch order chunk by chunk, through the LDS.  The
    // first request of a row takes the row's answer and its load — out of the row's LDS word (round 6; a gather from the column
    // per first request before: a memory round trip between every two stores of this walk, 22 us per workgroup); later requests
    // observe (LOCAL / REDIRECT, or UNPLACED) and carry the position of the first.
    auto answer = [&](const uint2 x, const u32 at) {
        const u64 e = wfirst[x.x & kPartRowMask];
        const u32 elo = (u32)e, ehi = (u32)(e >> 32);
        const u32 k = x.y, f = ehi >> 8, nd16 = elo & 0xFFFFu, l16 = elo >> 16, rq = x.x >> kPartShiftMax;
        const u32 nd = nd16 == 0xFFFFu ? kNone : nd16;
        if (f == k) {
            ans0[at] = pp_ans(nd, ehi & 0xFFu, false);
            ans1[at] = l16 != 0xFFFFu ? l16 : load[base + (x.x & kPartRowMask)];
        } else {
            ans0[at] = pp_ans(nd, nd == kNone ? 4u : (nd == rq ? 0u : 1u), true);
            ans1[at] = f;
        }
    };
#pragma unroll
    for (int i = 0; i < kKeep; ++i) {
        const u32 pbi = (u32)__shfl((int)pbv[(u32)i / LP], (int)(((u32)lane & ~(LP - 1u)) | ((u32)i % LP)), 64);  // (every lane)
        if (xr[i].y != kNone) answer(xr[i], pbi + o16);
    }
#pragma unroll 1
    for (u32 i = kKeep; i < (u32)kPartIters; i += kGrp) {  // the pieces whose records were not kept: read again
        if (i * (1024u / LP) >= nchunks) break;
        u32 pb[kGrp], pc[kGrp];
        uint2 x[kGrp];
#pragma unroll
        for (int q = 0; q < kGrp; ++q) piece(i + (u32)q, pb[q], pc[q]);
#pragma unroll
        for (int q = 0; q < kGrp; ++q) x[q] = rec2[o16 < pc[q] ? pb[q] + o16 : 0u];
#pragma unroll
        for (int q = 0; q < kGrp; ++q)
            if (o16 < pc[q]) answer(x[q], pb[q] + o16);
    }
    RIOGP_KTF(trace, 7, 5);
#pragma unroll 1
    for (u32 i = 0; i < (u32)kPartIters; ++i) {  // the flagged pieces' records past their first LP: descriptors again, then the records
        if (!__ballot((tailmask >> i) & 1u)) continue;
        u32 pb, pc;
        piece(i, pb, pc);
        for (u32 o = LP + o16; o < pc; o += LP) answer(rec2[pb + o], pb + o);
    }
    if (lds_hist)
        for (u32 j = tid; j < m; j += kBlock)
            if (hist[j]) atomicAdd(&claim[j], hist[j]);
    slow = wave_sum32(slow);
    if (lane == 0 && slow) atomicAdd(fast, (u64)slow);
    __syncthreads();
    RIOGP_KTF(trace, 7, 7);
}

// Does the batch need the solve?  No, when every request was answered by the window kernel and every requester's first
// touches fit its free capacity (then every index-ordered prefix of them fits too: nobody is cut, no ordered walk is needed
// — k_pp_one's argument, for any batch size): the answers are final, `used` takes the claims, k_pp_win_split hands the
// answers out.  Otherwise nothing is changed here and the host enqueues the solve over the same records.
// verdict: device word for k_pp_win_split | mapped host word for the caller: 1 final | 2 needs the solve | 3 invalid entry
__global__ __launch_bounds__(kBlock) void k_pp_win_verdict(u32 m, const u64* __restrict__ cap, const u32* __restrict__ alive_bits,
                                                           u64* __restrict__ used, const u64* __restrict__ claim,
                                                           const u64* __restrict__ fast, const DevStats* __restrict__ st,
                                                           u32* __restrict__ verdict_dev, u32* __restrict__ verdict_host) {
    const int tid = threadIdx.x;
    const bool bad = st->err != 0;
    bool over = false;
    for (u32 j = tid; j < m; j += kBlock) {
        const u64 c = claim[j];
        if (!c) continue;
        const u64 cp = cap[j], u = used[j];
        over = over || !bit_of(alive_bits, j) || cp <= u || c > cp - u;
    }
    const bool need = __syncthreads_or(over) || fast[0] != 0;
    if (!bad && !need)
        for (u32 j = tid; j < m; j += kBlock) used[j] += claim[j];
    if (tid == 0) {
        const u32 v = bad ? 3u : (need ? 2u : 1u);
        *verdict_dev = v;
        __hip_atomic_store(verdict_host, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The answers back in BATCH order: a chunk of the batch (kPartSub consecutive positions) was sorted by row window inside the
// chunk, so its answers sit in the chunk's own 8 192 sorted slots — one workgroup reads them densely together with the sorted
// records' batch positions, turns them round in the LDS and writes the chunk's positions densely: the caller's node / flag
// columns when the answers are final (verdict 1: what k_pp_win_split did from the records), the 8-byte records the solve and
// k_pp_win_output read when the batch needs the solve (verdict 2).  An invalid entry (3): nothing.
__global__ __launch_bounds__(kBlock) void k_pp_win_unsort(const uint2* __restrict__ rec2, const u32* __restrict__ ans0,
                                                          const u32* __restrict__ ans1, u64 n, uint2* __restrict__ vrec,
                                                          u32* __restrict__ out_node, u32* __restrict__ out_flag,
                                                          const u32* __restrict__ verdict, const u32 sub) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32* w0 = reinterpret_cast<u32*>(smem);  // [sub]
    u32* w1 = w0 + sub;                      // [sub]
    const u32 v = *verdict;
    if (v != 1u && v != 2u) return;
    const int tid = threadIdx.x;
    const u64 lo = (u64)blockIdx.x * sub;
    const u32 cnt = (u32)(n - lo < (u64)sub ? n - lo : (u64)sub);
    for (u32 j = tid; j < cnt; j += kBlock) {
        const u32 k = rec2[lo + j].y - (u32)lo;  // the record's batch position, inside this chunk
        w0[k] = ans0[lo + j];
        if (v == 2u) w1[k] = ans1[lo + j];
    }
    __syncthreads();
    if (v == 1u) {
        for (u32 j = tid; j < cnt; j += kBlock) {
            out_node[lo + j] = pp_ans_node(w0[j]);
            if (out_flag) out_flag[lo + j] = pp_ans_flag(w0[j]);
        }
    } else {
        for (u32 j = tid; j < cnt; j += kBlock) vrec[lo + j] = make_uint2(w0[j], w1[j]);
    }
}

#undef RIOGP_PART_DESCRIPTORS

// Outputs of a window-sorted batch, four requests per lane.  vnext = the solved virtual table's decisions (k_scan: kept ->
// the row's node, claimant -> its requester; the fix-up: rejected and spilled rows -> their node, or NONE after the last
// round); vload of a later request of an object = the batch position of the first.
__global__ __launch_bounds__(256) void k_pp_win_output(const u32* __restrict__ idx, const u32* __restrict__ req, u64 n,
                                                       const u32* __restrict__ vcur, const u32* __restrict__ vload,
                                                       const u32* __restrict__ vnext, const u32* __restrict__ alive_bits,
                                                       const u32* __restrict__ cutidx, u32 m, u32* __restrict__ out_node,
                                                       u32* __restrict__ out_flag, u32* __restrict__ aff_life,
                                                       const DevStats* __restrict__ st, u32 sa, const uint2* __restrict__ vrec) {
    // vrec: the window kernel's answer records — RIO_GP_FLAG_REPLACED of a request that found its object on a dead node sits in
    // the record's flag byte (the caller's flag column is written here, once; nobody has to clear it first)
    if (st->err) return;  // the batch holds an invalid entry: the call fails
    const u64 nv = (n + 3) >> 2;
    for (u64 v = (u64)blockIdx.x * 256 + threadIdx.x; v < nv; v += (u64)gridDim.x * 256) {
        const u64 k0 = v * 4;
        uint4 c, x, r, f, l;
        if (k0 + 4 <= n) {
            c = *reinterpret_cast<const uint4*>(vcur + k0);
            x = *reinterpret_cast<const uint4*>(vnext + k0);
            r = *reinterpret_cast<const uint4*>(req + k0);
            l = *reinterpret_cast<const uint4*>(vload + k0);
            const uint4 ra = *reinterpret_cast<const uint4*>(vrec + k0), rb = *reinterpret_cast<const uint4*>(vrec + k0 + 2);
            f = make_uint4(pp_ans_flag(ra.x), pp_ans_flag(ra.z), pp_ans_flag(rb.x), pp_ans_flag(rb.z));
        } else {
            u32 t[5][4] = {};
            for (u32 e = 0; k0 + e < n; ++e) {
                t[0][e] = vcur[k0 + e]; t[1][e] = vnext[k0 + e]; t[2][e] = req[k0 + e]; t[3][e] = vload[k0 + e];
                t[4][e] = pp_ans_flag(vrec[k0 + e].x);
            }
            c = make_uint4(t[0][0], t[0][1], t[0][2], t[0][3]); x = make_uint4(t[1][0], t[1][1], t[1][2], t[1][3]);
            r = make_uint4(t[2][0], t[2][1], t[2][2], t[2][3]); l = make_uint4(t[3][0], t[3][1], t[3][2], t[3][3]);
            f = make_uint4(t[4][0], t[4][1], t[4][2], t[4][3]);
        }
        uint4 on, of;
#define RIOGP_OUT(C, X, R, L, F, ON, OF, E)                                                              \
        if (k0 + E < n) {                                                                                \
            const u32 nd = C == kSkipMark ? vnext[L] : X;  /* a later request observes the first's */    \
            u32 fl;                                                                                      \
            if (nd == kNone) fl = 4u;                                       /* UNPLACED */               \
            else if (C == kNone) {                                          /* this request placed it */ \
                const bool claimed = (sa || bit_of(alive_bits, R)) && (u32)(k0 + E) < cutidx[R];         \
                fl = (claimed ? 2u : 3u) | (F & kFlagReplaced);             /* PLACED | SPILLED */       \
                if (aff_life && (F & kFlagReplaced)) aff_life[idx[k0 + E]] = R;                          \
            } else fl = nd == R ? 0u : 1u;                                  /* LOCAL | REDIRECT */       \
            if (C == kNone && nd == kNone) {                                                             \
                fl |= F & kFlagReplaced;                                                                 \
                if (aff_life && (F & kFlagReplaced)) aff_life[idx[k0 + E]] = R;                          \
            }                                                                                            \
            ON = nd; OF = fl;                                                                            \
        } else { ON = 0; OF = 0; }
        RIOGP_OUT(c.x, x.x, r.x, l.x, f.x, on.x, of.x, 0)
        RIOGP_OUT(c.y, x.y, r.y, l.y, f.y, on.y, of.y, 1)
        RIOGP_OUT(c.z, x.z, r.z, l.z, f.z, on.z, of.z, 2)
        RIOGP_OUT(c.w, x.w, r.w, l.w, f.w, on.w, of.w, 3)
#undef RIOGP_OUT
        if (k0 + 4 <= n) {
            *reinterpret_cast<uint4*>(out_node + k0) = on;
            if (out_flag) *reinterpret_cast<uint4*>(out_flag + k0) = of;
        } else {
            const u32 a[4] = {on.x, on.y, on.z, on.w}, b[4] = {of.x, of.y, of.z, of.w};
            for (u32 e = 0; k0 + e < n; ++e) { out_node[k0 + e] = a[e]; if (out_flag) out_flag[k0 + e] = b[e]; }
        }
    }
}

// clean_server(s) (local.rs:51-58): one coalesced pass, 4 B read per row, 4 B written per evicted row
// counter: device accumulator of evicted rows.  ticket (non-null = synchronous call) / host_out: the last workgroup to
// finish copies the total into mapped host memory and resets the counter, so the call needs no memset / copy-back.
__global__ __launch_bounds__(kBlock) void k_clean(u32* __restrict__ assign, u64 n_obj, u32 m,
                                               const u32* __restrict__ dead_bits, u64* __restrict__ used,
                                               u64* __restrict__ counter, unsigned int* __restrict__ ticket,
                                               u64* __restrict__ host_out, u32* __restrict__ aff_life, u32 seq,
                                               const u32* __restrict__ skip_if, const u32 full_from) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (skip_if && *skip_if) return;  // request path: the batch holds an invalid entry, the call changes nothing (asynchronous form only)
    u32* db = reinterpret_cast<u32*>(smem);
    __shared__ u32 any;
    __shared__ u32 ev_total;
    const u32 mwords = (m + 31) / 32;
    const int tid = threadIdx.x;
    // At most 256 workgroups of 1 024 threads in a grid-stride loop: the kernel ends with a returning atomic per workgroup on
    // ONE address, and those serialise at ~10 ns each — 2 048 workgroups of 256 threads spent 40 us there, five times the
    // 8 us of streaming the column.
    const u64 nvec = (n_obj + 3) / 4, stride = (u64)gridDim.x * kBlock;
    const u64 vlast = nvec ? nvec - 1 : 0;  // (an empty table: vector 0 of the padded column, judged by nobody)
    u64 v = (u64)blockIdx.x * kBlock + tid;
    // kCleanFlight vectors per lane in flight ahead of the ones being judged; the first is requested before the bitmap set-up
    // below.  Measured on the 10 M-row column (rocprofv3, one node / 10 % of the nodes): 1 -> 13.3 / 16.8 us, 2 -> 13.7 / 17.5,
    // 4 -> 14.8 / 18.0, 8 -> 17.7 / 19.5; 128 or 512 workgroups instead of 256 lose 2-4 us.  The pass is not latency-bound:
    // ~3 us of launch, ~8 us of stream (the column comes from DRAM), the tail below.
    constexpr int kCleanFlight = 1;
    uint4 c[kCleanFlight];
#pragma unroll
    for (int q = 0; q < kCleanFlight; ++q) {
        const u64 vq = v + (u64)q * stride;  // (past the end: the last vector again, unconditionally — a load behind a
        c[q] = *reinterpret_cast<const uint4*>(assign + (vq < nvec ? vq : vlast) * 4);     // branch would end the overlap)
    }
    if (tid == 0) { any = 0; ev_total = 0; }
    __syncthreads();
    u32 mine = 0;
    for (u32 k = tid; k < mwords; k += kBlock) { const u32 w = dead_bits[k]; db[k] = w; mine |= w; }
    if (mine) any = 1;
    __syncthreads();
    if (!any) return;
    if (blockIdx.x == 0 && used)
        for (u32 j = tid; j < m; j += kBlock)
            if (bit_of(db, j)) used[j] = 0;
    u32 ev = 0;
    for (; v < nvec; v += (u64)kCleanFlight * stride) {
        uint4 cn[kCleanFlight];
#pragma unroll
        for (int q = 0; q < kCleanFlight; ++q) {
            const u64 vq = v + (u64)(kCleanFlight + q) * stride;
            cn[q] = *reinterpret_cast<const uint4*>(assign + (vq < nvec ? vq : vlast) * 4);
        }
#pragma unroll
        for (int q = 0; q < kCleanFlight; ++q) {
            const u64 i0 = (v + (u64)q * stride) * 4;  // (past the end: i0 >= n_obj, nothing is evicted or written)
            uint4 x = c[q];
            const bool e0 = i0 + 0 < n_obj && x.x < m && bit_of(db, x.x), e1 = i0 + 1 < n_obj && x.y < m && bit_of(db, x.y);
            const bool e2 = i0 + 2 < n_obj && x.z < m && bit_of(db, x.z), e3 = i0 + 3 < n_obj && x.w < m && bit_of(db, x.w);
            const bool mine_ev = e0 | e1 | e2 | e3;
            // A wave with many evictions writes its whole kilobyte back, every lane its 16 bytes (unchanged rows keep their
            // value; nothing else writes the column while this runs): full 128-byte lines instead of scattered 16-byte pieces
            // of them.  With a few (one node of a thousand: a fifth of the waves hold one or two) only those lanes write —
            // the kernel takes the same time either way, the column is not rewritten for nothing.  Padded past n_obj.
            const u64 bal = __ballot(mine_ev);
            if (bal) {
                x.x = e0 ? kNone : x.x; x.y = e1 ? kNone : x.y; x.z = e2 ? kNone : x.z; x.w = e3 ? kNone : x.w;
                if (i0 < n_obj && (mine_ev || (u32)__popcll(bal) > full_from)) *reinterpret_cast<uint4*>(assign + i0) = x;
            }
            if (mine_ev) {
                if (aff_life) {  // row lifecycle: retain() drops the entries (local.rs:51-58); they come back on their next request
                    if (e0) aff_life[i0 + 0] = kAffInactive;
                    if (e1) aff_life[i0 + 1] = kAffInactive;
                    if (e2) aff_life[i0 + 2] = kAffInactive;
                    if (e3) aff_life[i0 + 3] = kAffInactive;
                }
                ev += e0 + e1 + e2 + e3;
            }
        }
#pragma unroll
        for (int q = 0; q < kCleanFlight; ++q) c[q] = cn[q];
    }
    ev = wave_sum32(ev);
    if ((tid & 63) == 0 && ev) atomicAdd(&ev_total, ev);
    __syncthreads();
    if (tid == 0) {
        if (!ticket) {  // asynchronous callers: a plain accumulator
            if (ev_total) atomicAdd(counter, (u64)ev_total);
            return;
        }
        // Synchronous call: count (bits 0..39) and arrival ticket (bits 40..63) travel in ONE returning atomic per workgroup.
        // Returning atomics on one address serialise at ~12 ns each and every workgroup arrives at the same moment, so they
        // go through two levels: eight counters on lines of their own (workgroups of one XCD share one), then the last
        // arrival of each group adds its group's sum to counter[0] — 32 + 8 in a row instead of 256.
        constexpr u64 kCnt = (1ull << 40) - 1;
        const u32 groups = gridDim.x < 8u ? gridDim.x : 8u, grp = blockIdx.x % groups;
        const u32 gsize = (gridDim.x - grp + groups - 1) / groups;
        u64* gc = counter + (size_t)(1 + grp) * 16;
        const u64 before = __hip_atomic_fetch_add(gc, (u64)ev_total + (1ull << 40), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((before >> 40) != (u64)gsize - 1) return;
        __hip_atomic_store(gc, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 gsum = ((before & kCnt) + ev_total) & kCnt;
        const u64 b2 = __hip_atomic_fetch_add(counter, gsum + (1ull << 40), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((b2 >> 40) != (u64)groups - 1) return;
        // the total, tagged with the caller's sequence number in bits 40..63: the host spins on the tag (one 8-byte store)
        *host_out = (((b2 & kCnt) + gsum) & kCnt) | ((u64)seq << 40);
        __hip_atomic_store(counter, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// scatter new load / affinity values into individual rows
__global__ void k_set_attrs(u32* __restrict__ load, u32* __restrict__ aff, u64 n_obj, const u32* __restrict__ idx,
                            const u32* __restrict__ nload, const u32* __restrict__ naff, u64 n, DevStats* st) {
    for (u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (u64)gridDim.x * blockDim.x) {
        const u32 i = idx[k];
        if (i >= n_obj) { atomicAdd(&st->err, 1ull); continue; }
        if (nload) load[i] = nload[k];
        if (naff) aff[i] = naff[k];
    }
}

// number of placed rows: 4 B/row
__global__ __launch_bounds__(256) void k_count_placed(const u32* __restrict__ assign, u64 n_obj, DevStats* st) {
    u32 c = 0;
    const u64 nvec = (n_obj + 3) / 4;
    for (u64 v = (u64)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (u64)gridDim.x * 256) {
        const u64 i0 = v * 4;
        const uint4 a = *reinterpret_cast<const uint4*>(assign + i0);
        c += (i0 + 0 < n_obj && a.x != kNone) + (i0 + 1 < n_obj && a.y != kNone) + (i0 + 2 < n_obj && a.z != kNone) +
             (i0 + 3 < n_obj && a.w != kNone);
    }
    c = wave_sum32(c);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&st->evicted_clean, (u64)c);  // reuses a scratch accumulator
}

// used[j] = sum of load over rows assigned to j (8 B/row)
__global__ __launch_bounds__(kBlock) void k_used(const u32* __restrict__ assign, const u32* __restrict__ load,
                                                 u64 n_obj, u32 m, u64* __restrict__ used) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* hist = reinterpret_cast<u64*>(smem);
    const int tid = threadIdx.x;
    for (u32 k = tid; k < m; k += kBlock) hist[k] = 0;
    __syncthreads();
    const u64 nvec = (n_obj + 3) / 4;
    for (u64 v = (u64)blockIdx.x * kBlock + tid; v < nvec; v += (u64)gridDim.x * kBlock) {
        const u64 i0 = v * 4;
        const uint4 c = *reinterpret_cast<const uint4*>(assign + i0);
        const uint4 l = *reinterpret_cast<const uint4*>(load + i0);
        if (i0 + 0 < n_obj && c.x < m) atomicAdd(&hist[c.x], (u64)l.x);
        if (i0 + 1 < n_obj && c.y < m) atomicAdd(&hist[c.y], (u64)l.y);
        if (i0 + 2 < n_obj && c.z < m) atomicAdd(&hist[c.z], (u64)l.z);
        if (i0 + 3 < n_obj && c.w < m) atomicAdd(&hist[c.w], (u64)l.w);
    }
    __syncthreads();
    for (u32 k = tid; k < m; k += kBlock)
        if (hist[k]) atomicAdd(&used[k], hist[k]);
}

// ------------------------------------------------------------------------------------------------
// place_pending, the general request path (any batch the one-workgroup and the window-sorted forms do not take: 4 097 ...
// 2^18 requests, sparse or unaligned bigger ones, small ones that ran into a dead node or a full requester):
//     k_ppm_first   validate | first request per row (atomicMin of the batch position into the row-sized scratch) | mark the
//                   dead nodes requests run into (service.rs:227-237) | device copies of requests that live in mapped HOST memory
//     [k_clean]     only when a node is not alive
//     k_ppm_gather  the virtual table (rows = requests): cur of the first request, kSkipMark for later ones, load, and the
//                   position of the row's first request (what the output kernel needs instead of the scratch)
//     k_scan<VIRT> -> k_resolve -> [fix-up, speculative when the last batch needed it]
//     k_ppm_output  decisions into the real column (first requests), answers in batch order (duplicates read the first's
//                   virtual row), scratch reset, completion word.
//   Nothing waits on the host in between (round 4: two round trips, 67-70 us for 8 192 requests).  An invalid entry raises
//   *bad (a device word the last workgroup of the output kernel puts back to 0): every kernel behind k_ppm_first looks at it
//   and changes nothing, the output kernel undoes the election marks and reports status 3.  A solve that needs the cut /
//   water-fill when the fix-up was not enqueued leaves everything as it is and reports status 1: the host enqueues the
//   fix-up and this kernel again (one extra round trip, on the first contended batch only).
// ------------------------------------------------------------------------------------------------
constexpr u32 kPpmOk = 0, kPpmNeedFix = 1, kPpmBad = 3;
__global__ __launch_bounds__(256) void k_ppm_first(const u32* __restrict__ assign, u64 n_obj, u32 m,
                                                   const u32* __restrict__ alive_bits, const u32* __restrict__ idx,
                                                   const u32* __restrict__ req, u64 n, u32* __restrict__ pos,
                                                   u32* __restrict__ s_idx, u32* __restrict__ s_req,
                                                   u32* __restrict__ dead_bits, u32* __restrict__ vflag, u32* __restrict__ bad,
                                                   const u32 vec) {  // vec: every array is 16-byte aligned (else one entry per lane)
    // s_idx / s_req: the requests are read ONCE here, 16 bytes per lane (they may live in mapped host memory), and left in the
    // library's own padded device arrays for the kernels behind; dead_bits / vflag (some node is not alive): RIO_GP_FLAG_REPLACED for a request that finds its
    // object on a dead node, 0 otherwise — the output kernel keeps the bit for the FIRST request of the object (service.rs:268-285)
    u32 nbad = 0;
    auto one = [&](u64 k, u32 i, u32 r) -> u32 {
        if (i >= n_obj || r >= m) { ++nbad; return 0u; }
        atomicMin(&pos[i], (u32)k);
        if (!dead_bits) return 0u;
        const u32 c = assign[i];
        const bool dead = c < m && !bit_of(alive_bits, c);
        if (dead) atomicOr(&dead_bits[c >> 5], 1u << (c & 31));
        return dead ? kFlagReplaced : 0u;
    };
    const u64 nvec = vec ? n >> 2 : 0, stride = (u64)gridDim.x * 256;
    for (u64 v = (u64)blockIdx.x * 256 + threadIdx.x; v < nvec; v += stride) {
        const uint4 iv = *reinterpret_cast<const uint4*>(idx + 4 * v), rv = *reinterpret_cast<const uint4*>(req + 4 * v);
        *reinterpret_cast<uint4*>(s_idx + 4 * v) = iv;
        *reinterpret_cast<uint4*>(s_req + 4 * v) = rv;
        uint4 f;
        f.x = one(4 * v + 0, iv.x, rv.x); f.y = one(4 * v + 1, iv.y, rv.y);
        f.z = one(4 * v + 2, iv.z, rv.z); f.w = one(4 * v + 3, iv.w, rv.w);
        if (dead_bits) *reinterpret_cast<uint4*>(vflag + 4 * v) = f;
    }
    for (u64 k = nvec * 4 + (u64)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {  // ragged tail, or arrays that are not aligned
        const u32 i = idx[k], r = req[k];
        s_idx[k] = i;
        s_req[k] = r;
        const u32 f = one(k, i, r);
        if (dead_bits) vflag[k] = f;
    }
    if (nbad) __hip_atomic_store(bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void k_ppm_gather(const u32* __restrict__ assign, const u32* __restrict__ load,
                                                    const u32* __restrict__ idx, u64 n, const u32* __restrict__ pos,
                                                    u32* __restrict__ vcur, u32* __restrict__ vload, u32* __restrict__ vfirst,
                                                    const u32* __restrict__ bad, const u32 vec) {
    if (*bad) return;  // (every entry is valid from here on)
    const u64 nvec = vec ? n >> 2 : 0, stride = (u64)gridDim.x * 256;
    for (u64 v = (u64)blockIdx.x * 256 + threadIdx.x; v < nvec; v += stride) {
        const uint4 iv = *reinterpret_cast<const uint4*>(idx + 4 * v);
        // twelve independent gathers in flight per lane before the first is used
        const u32 p0 = pos[iv.x], p1 = pos[iv.y], p2 = pos[iv.z], p3 = pos[iv.w];
        const u32 a0 = assign[iv.x], a1 = assign[iv.y], a2 = assign[iv.z], a3 = assign[iv.w];
        const uint4 lv = make_uint4(load[iv.x], load[iv.y], load[iv.z], load[iv.w]);
        const u32 k0 = (u32)(4 * v);
        *reinterpret_cast<uint4*>(vfirst + 4 * v) = make_uint4(p0, p1, p2, p3);
        *reinterpret_cast<uint4*>(vcur + 4 * v) = make_uint4(p0 == k0 ? a0 : kSkipMark, p1 == k0 + 1 ? a1 : kSkipMark,
                                                             p2 == k0 + 2 ? a2 : kSkipMark, p3 == k0 + 3 ? a3 : kSkipMark);
        *reinterpret_cast<uint4*>(vload + 4 * v) = lv;
    }
    for (u64 k = nvec * 4 + (u64)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {
        const u32 i = idx[k], f = pos[i];
        vfirst[k] = f;
        vcur[k] = f == (u32)k ? assign[i] : kSkipMark;
        vload[k] = load[i];
    }
}

struct PpmOutArgs {
    u32* assign; const u32* idx; const u32* req; u64 n; u64 n_obj;
    const u32* vcur; const u32* vnext; const u32* vfirst; const u32* vflag;  // vflag: nullptr when every node is alive
    u32* pos; const u32* alive_bits; const u32* cutidx; u32 m; u32 sa;
    u32* out_node; u32* out_flag; u32* aff_life;
    u32* bad; const DevStats* stats; const u32* bsp_cnt; u32 G; u32 fixup_done; u32 vec;
    u32* status; unsigned int* ticket; u32* done; u32 seq;
};
__global__ __launch_bounds__(256) void k_ppm_output(const PpmOutArgs a) {
    const u32 isbad = *a.bad;
    // did the solve of the virtual table need the cut / water-fill?  (a node with a cut, or a row k_scan sent to the water-fill)
    const u32 pend = (!a.fixup_done && threadIdx.x < a.G) ? a.bsp_cnt[threadIdx.x] : 0u;
    const bool slow = __syncthreads_or(pend != 0u || (!a.fixup_done && threadIdx.x == 0 && a.stats->n_cut != 0));
    const u32 st = isbad ? kPpmBad : (slow ? kPpmNeedFix : kPpmOk);
    const u64 stride = (u64)gridDim.x * 256;
    if (st == kPpmBad) {  // the valid entries' election marks go back to all-ones; nothing else was changed
        for (u64 k = (u64)blockIdx.x * 256 + threadIdx.x; k < a.n; k += stride) {
            const u32 i = a.idx[k];
            if (i < a.n_obj && a.req[k] < a.m) a.pos[i] = kNone;
        }
    } else if (st == kPpmOk) {
        auto one = [&](u64 k, u32 i, u32 r, u32 first, u32& nd_out) -> u32 {
            const bool mine = first == (u32)k;
            const u32 c0 = a.vcur[first];           // the row as its first request found it: a node (sticky) or NONE (pending)
            const bool placed_now = c0 == kNone;
            u32 nd = c0;
            if (placed_now) {
                const u32 nx = a.vnext[first];
                nd = nx < kSkipMark ? nx : kNone;    // (kSpillMark: no room anywhere)
            }
            if (mine) {
                if (placed_now) {
                    if (nd != kNone) a.assign[i] = nd;
                    // row lifecycle: the first request of a pending row makes it an object (placed or not), home = the requester
                    if (a.aff_life) a.aff_life[i] = r;
                }
                a.pos[i] = kNone;
            }
            u32 fl;
            if (nd == kNone) fl = 4u;                                  // UNPLACED
            else if (mine && placed_now) {                             // this request placed the row
                const bool claimed = (a.sa || bit_of(a.alive_bits, r)) && (u32)k < a.cutidx[r];
                fl = claimed ? 2u : 3u;                                // PLACED | SPILLED
            } else fl = (nd == r) ? 0u : 1u;                           // LOCAL | REDIRECT
            // the first request of an object it found on a server that is not alive keeps that fact, whatever the outcome
            if (mine && placed_now && a.vflag) fl |= a.vflag[k] & kFlagReplaced;
            nd_out = nd;
            return fl;
        };
        const u64 nvec = a.vec ? a.n >> 2 : 0;
        for (u64 v = (u64)blockIdx.x * 256 + threadIdx.x; v < nvec; v += stride) {
            const uint4 iv = *reinterpret_cast<const uint4*>(a.idx + 4 * v), rv = *reinterpret_cast<const uint4*>(a.req + 4 * v);
            const uint4 fv = *reinterpret_cast<const uint4*>(a.vfirst + 4 * v);
            uint4 on, of;
            of.x = one(4 * v + 0, iv.x, rv.x, fv.x, on.x); of.y = one(4 * v + 1, iv.y, rv.y, fv.y, on.y);
            of.z = one(4 * v + 2, iv.z, rv.z, fv.z, on.z); of.w = one(4 * v + 3, iv.w, rv.w, fv.w, on.w);
            *reinterpret_cast<uint4*>(a.out_node + 4 * v) = on;
            if (a.out_flag) *reinterpret_cast<uint4*>(a.out_flag + 4 * v) = of;
        }
        for (u64 k = nvec * 4 + (u64)blockIdx.x * 256 + threadIdx.x; k < a.n; k += stride) {
            u32 nd;
            const u32 fl = one(k, a.idx[k], a.req[k], a.vfirst[k], nd);
            a.out_node[k] = nd;
            if (a.out_flag) a.out_flag[k] = fl;
        }
    }
    // status, then the completion word; the last workgroup to arrive also puts the bad-entry word back to 0 (everybody has read it)
    if (threadIdx.x == 0) __hip_atomic_store(a.status, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (st != kPpmNeedFix) __hip_atomic_store(a.bad, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.done, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}


// ------------------------------------------------------------------------------------------------
// place_pending, batches of up to kOneBatch requests: the whole policy of service.rs:193-298 in ONE workgroup and ONE
// launch — request and result arrays are pinned host memory mapped into the device (no staging copies), so a call is a
// launch and a wait.  Handles the common case completely: sticky hits (LOCAL / REDIRECT), first touch on a live requester
// with room (PLACED), duplicates of one object in the batch.  Anything that needs the heavy machinery — a requested object
// sitting on a dead node (clean_server of that node), a dead requester, a requester whose free capacity the batch's first
// touches exceed (strict prefix cut + water-fill) — makes it return status 1 having changed NOTHING, and the caller runs
// the general path.  Results are identical either way: when every requester's TOTAL claim load fits its free capacity,
// every index-ordered prefix fits too, so no ordered walk is needed to know that everybody is admitted.
//   THREADS x PER requests: 256 x 1 (micro-batches, the reference's one-object-per-request flow) | 1024 x 4.
//   The first request of an object decides (batch order): an open-addressing table in LDS keyed by the row, atomicMin on
//   the batch position; later duplicates read the winner's result from the same slot.
// ------------------------------------------------------------------------------------------------
constexpr u32 kPpTot = 2048;  // nodes up to which k_pp_one keeps a per-requester claim total in LDS (16 KiB)
template <int THREADS, int PER>
__global__ __launch_bounds__(THREADS) void k_pp_one(u32* __restrict__ assign, const u32* __restrict__ load, u32 m,
                                                    const u64* __restrict__ cap, const u32* __restrict__ alive_bits,
                                                    u64* __restrict__ used, const u32* __restrict__ idx,
                                                    const u32* __restrict__ req, u32 n, u32* __restrict__ out_node,
                                                    u32* __restrict__ out_flag, u32* __restrict__ status,
                                                    u32* __restrict__ aff_life, u32* done, u32 seq, u32 ninl, uint4 ia,
                                                    uint4 ib, u32 n_obj_chk, u32 trace, u32 sa, const CrudSmall crud) {
    // crud (micro-batches only): the update / remove / lookup parts of a mixed batch, applied before the requests are looked at
    // n_obj_chk != 0 (requests the host has not seen: rio_gp_place_pending_dev): the number of rows — an object index or a
    // requester out of range ends the call with status 3 and nothing changed; the arrays are then exactly n entries long
    // (no whole vector past the end)
    constexpr u32 kSlots = 2u * THREADS * PER;  // power of two
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* s_tot = reinterpret_cast<u64*>(smem);               // [kPpTot] claim load per requester
    u32* hkey = reinterpret_cast<u32*>(s_tot + kPpTot);      // [kSlots] row of the slot
    u32* hpos = hkey + kSlots;                               // [kSlots] first batch position of the row, then its final node
    __shared__ u32 s_general, s_bad;
    const u32 tid = threadIdx.x;
    (void)trace;
    RIOGP_KTF(trace, 6, 0);
    if (PER == 1 && THREADS == kSmallBatch && (crud.nu | crud.nr | crud.nl))
        dev_crud_small(crud, assign, load, m, used, aff_life, hkey, hpos, true);  // (its tables: this kernel's, kSlots = 2 * kSmallBatch)
    if (tid == 0) { s_general = m > kPpTot ? 1u : 0u; s_bad = 0; }
    for (u32 q = tid; q < kSlots; q += THREADS) { hkey[q] = kNone; hpos[q] = kNone; }
    for (u32 q = tid; q < kPpTot; q += THREADS) s_tot[q] = 0;
    u32 i[PER], r[PER], c[PER], l[PER], slot[PER];
    u64 fre[PER];
    bool valid[PER], r_alive[PER];
    // everything a request may need is requested at once: the call is a chain of device round trips, not bandwidth.
    // A thread takes PER CONSECUTIVE requests (k = PER * tid + q): with PER = 4 its indices and requesters arrive as one
    // 16-byte read each from the mapped host buffer (PCIe read requests are what a 4 096-request call waits for) and its
    // results leave as 16-byte stores.
    uint4 iv4 = make_uint4(0, 0, 0, 0), rv4 = iv4;
    const bool vec = PER == 4 && !ninl && (!n_obj_chk || 4 * tid + 4 <= n);  // (the staging rows are kMidBatch entries long)
    if (vec) {
        iv4 = *reinterpret_cast<const uint4*>(idx + 4 * tid);
        rv4 = *reinterpret_cast<const uint4*>(req + 4 * tid);
    }
    bool bad = false;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const u32 k = tid * PER + (u32)q;
        valid[q] = k < n;
        i[q] = 0; r[q] = 0; c[q] = kNone; l[q] = 0; fre[q] = 0; r_alive[q] = false;
        if (valid[q]) {
            i[q] = ninl ? inl_sel(ia, k) : (vec ? inl_sel(iv4, (u32)q) : idx[k]);
            r[q] = ninl ? inl_sel(ib, k) : (vec ? inl_sel(rv4, (u32)q) : req[k]);
            if (n_obj_chk && (i[q] >= n_obj_chk || r[q] >= m)) { bad = true; valid[q] = false; i[q] = 0; r[q] = 0; continue; }
            c[q] = assign[i[q]];
            l[q] = load[i[q]];
            const u64 cj = cap[r[q]], uj = used[r[q]];
            fre[q] = cj > uj ? cj - uj : 0;
            r_alive[q] = sa || bit_of(alive_bits, r[q]);
        }
    }
    __syncthreads();
    RIOGP_KTF(trace, 6, 1);
    if (bad) s_bad = 1;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        slot[q] = (i[q] * 2654435761u) >> 7 & (kSlots - 1);  // (row ids are < 2^31: kNone never is one)
        if (valid[q]) {
            for (;;) {
                const u32 old = atomicCAS(&hkey[slot[q]], kNone, i[q]);
                if (old == kNone || old == i[q]) break;
                slot[q] = (slot[q] + 1) & (kSlots - 1);
            }
            atomicMin(&hpos[slot[q]], tid * PER + (u32)q);
        }
    }
    __syncthreads();
    RIOGP_KTF(trace, 6, 2);
    bool first[PER], claim[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        first[q] = false; claim[q] = false;
        if (valid[q]) {
            first[q] = hpos[slot[q]] == tid * PER + (u32)q;
            const bool dead_cur = c[q] < m && !bit_of(alive_bits, c[q]);   // service.rs:227-237 -> clean_server: general path
            const bool pending = first[q] && c[q] == kNone;
            claim[q] = pending && r_alive[q];
            if (dead_cur || (pending && !claim[q])) s_general = 1;
            if (claim[q] && r[q] < kPpTot) atomicAdd(&s_tot[r[q]], (u64)l[q]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q)
        if (claim[q] && r[q] < kPpTot && s_tot[r[q]] > fre[q]) s_general = 1;  // the requester cannot take all its first touches
    __syncthreads();
    RIOGP_KTF(trace, 6, 3);
    if (s_general | s_bad) {  // hand over untouched (3: an entry out of range — the call fails)
        if (tid == 0) *status = s_bad ? 3u : 1u;
        signal_done(done, seq);
        return;
    }
#pragma unroll
    for (int q = 0; q < PER; ++q)  // the row's final node, for the later requests of the same object in this batch
        if (valid[q] && first[q]) hpos[slot[q]] = claim[q] ? r[q] : c[q];
    __syncthreads();
    RIOGP_KTF(trace, 6, 4);
    u32 ond[PER], ofl[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        ond[q] = kNone; ofl[q] = 0;
        if (!valid[q]) continue;
        if (claim[q]) {  // first touch (service.rs:244-252)
            assign[i[q]] = r[q];
            atomicAdd(&used[r[q]], (u64)l[q]);
            if (aff_life) aff_life[i[q]] = r[q];  // row lifecycle: the object exists from its first touch, its home is the requester
            ond[q] = r[q];
            ofl[q] = 2u;  // PLACED
        } else {
            ond[q] = hpos[slot[q]];  // (its own current node, or what the first request decided)
            ofl[q] = (ond[q] == kNone) ? 4u : (ond[q] == r[q] ? 0u : 1u);
        }
    }
    if (PER == 4 && (!n_obj_chk || 4 * tid + 4 <= n)) {
        if (valid[0]) {  // (the result rows are kMidBatch entries long: a whole vector is always addressable)
            *reinterpret_cast<uint4*>(out_node + 4 * tid) = make_uint4(ond[0], ond[PER > 1 ? 1 : 0], ond[PER > 2 ? 2 : 0], ond[PER > 3 ? 3 : 0]);
            *reinterpret_cast<uint4*>(out_flag + 4 * tid) = make_uint4(ofl[0], ofl[PER > 1 ? 1 : 0], ofl[PER > 2 ? 2 : 0], ofl[PER > 3 ? 3 : 0]);
        }
    } else {
#pragma unroll
        for (int q = 0; q < PER; ++q)
            if (valid[q]) { out_node[tid * PER + q] = ond[q]; out_flag[tid * PER + q] = ofl[q]; }
    }
    if (tid == 0) *status = 0;
    RIOGP_KTF(trace, 6, 5);
    signal_done(done, seq);
    RIOGP_KTF(trace, 6, 7);
}

// ------------------------------------------------------------------------------------------------
// The same call for batches between 1 024 and kOneBatch requests, in THREE launches.  What the one-workgroup kernel spends
// on such a batch is not the decision: of the 38 us of a 4 096-request first-touch batch 21 go into reading the requests
// over PCIe and gathering every request's row and node operands from ONE compute unit (sixteen thousand random lane-requests
// through one texture path), 10 into draining 8 192 scattered stores and atomics and the results' PCIe writes behind one
// fence (its own phase stamps, round 4: profiles/archive/round4_pp_host_batches.txt).  Both ends are spread over the chip here, the decision stays where
// it was — one workgroup, one LDS table, no cross-workgroup protocol (kernel boundaries order the three):
//   k_pp_stage   256 threads x 1 request: request (mapped host or device memory) -> {row, requester, cur, load | free
//                capacity of the requester, its liveness} in a device staging table; an entry out of range raises the flag
//   k_pp_decide  ONE workgroup, 1 024 x 4: k_pp_one's decision over the staged records (coalesced reads) -> per request
//                {node, flag | claim bit} back into the staging table, the requesters' claim totals into `used`, the status
//   k_pp_apply   256 threads x 1 request, nothing to do unless the status is 0: results to the caller's arrays, first
//                touches into the assignment column (and the lifecycle column); the last workgroup stores the completion word
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pp_stage(const u32* __restrict__ assign, const u32* __restrict__ load, u32 m,
                                                  const u64* __restrict__ cap, const u32* __restrict__ alive_bits,
                                                  const u64* __restrict__ used, const u32* __restrict__ idx,
                                                  const u32* __restrict__ req, u32 n, uint4* __restrict__ rec,
                                                  uint4* __restrict__ rec2, u32* __restrict__ st, u32 n_obj_chk, u32 sa) {
    const u32 k = blockIdx.x * 256u + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) st[1] = 2u;  // status: "not decided" until k_pp_decide says otherwise
    if (k >= n) return;
    u32 i = idx[k], r = req[k];
    bool bad = false;
    if (n_obj_chk && (i >= n_obj_chk || r >= m)) { bad = true; i = 0; r = 0; }
    const u32 c = assign[i], l = load[i];
    const u64 cj = cap[r], uj = used[r];
    const u64 fre = cj > uj ? cj - uj : 0;
    rec[k] = make_uint4(i, r, c, l);
    rec2[k] = make_uint4((u32)fre, (u32)(fre >> 32), (sa || bit_of(alive_bits, r)) ? 1u : 0u, bad ? 1u : 0u);
    if (bad) atomicOr(&st[0], 1u);
}

__global__ __launch_bounds__(kBlock) void k_pp_decide(const uint4* __restrict__ rec, const uint4* __restrict__ rec2, u32 n,
                                                      u32 m, const u32* __restrict__ alive_bits, u64* __restrict__ used,
                                                      uint2* __restrict__ res, u32* __restrict__ st, u32* __restrict__ status) {
    constexpr int PER = kOneBatch / kBlock;
    constexpr u32 kSlots = 2u * kOneBatch;  // power of two
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* s_tot = reinterpret_cast<u64*>(smem);               // [kPpTot] claim load per requester
    u32* hkey = reinterpret_cast<u32*>(s_tot + kPpTot);      // [kSlots] row of the slot
    u32* hpos = hkey + kSlots;                               // [kSlots] first batch position of the row, then its final node
    __shared__ u32 s_general;
    const u32 tid = threadIdx.x;
    const u32 bad_any = st[0];
    if (tid == 0) s_general = m > kPpTot ? 1u : 0u;
    for (u32 q = tid; q < kSlots; q += kBlock) { hkey[q] = kNone; hpos[q] = kNone; }
    for (u32 q = tid; q < kPpTot; q += kBlock) s_tot[q] = 0;
    u32 i[PER], r[PER], c[PER], l[PER], slot[PER];
    u64 fre[PER];
    bool valid[PER], r_alive[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const u32 k = tid * PER + (u32)q;
        valid[q] = k < n;
        const uint4 a = rec[valid[q] ? k : 0u], b = rec2[valid[q] ? k : 0u];
        i[q] = a.x; r[q] = a.y; c[q] = a.z; l[q] = a.w;
        fre[q] = ((u64)b.y << 32) | b.x;
        r_alive[q] = b.z != 0;
    }
    __syncthreads();
    if (bad_any) {  // an entry out of range: the call fails, nothing is changed (block-uniform)
        if (tid == 0) { st[0] = 0; st[1] = 3u; *status = 3u; }
        return;
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        slot[q] = (i[q] * 2654435761u) >> 7 & (kSlots - 1);
        if (valid[q]) {
            for (;;) {
                const u32 old = atomicCAS(&hkey[slot[q]], kNone, i[q]);
                if (old == kNone || old == i[q]) break;
                slot[q] = (slot[q] + 1) & (kSlots - 1);
            }
            atomicMin(&hpos[slot[q]], tid * PER + (u32)q);
        }
    }
    __syncthreads();
    bool first[PER], claim[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        first[q] = false; claim[q] = false;
        if (valid[q]) {
            first[q] = hpos[slot[q]] == tid * PER + (u32)q;
            const bool dead_cur = c[q] < m && !bit_of(alive_bits, c[q]);   // service.rs:227-237 -> clean_server: general path
            const bool pending = first[q] && c[q] == kNone;
            claim[q] = pending && r_alive[q];
            if (dead_cur || (pending && !claim[q])) s_general = 1;
            if (claim[q] && r[q] < kPpTot) atomicAdd(&s_tot[r[q]], (u64)l[q]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q)
        if (claim[q] && r[q] < kPpTot && s_tot[r[q]] > fre[q]) s_general = 1;  // the requester cannot take all its first touches
    __syncthreads();
    if (s_general) {  // hand over untouched
        if (tid == 0) { st[1] = 1u; *status = 1u; }
        return;
    }
#pragma unroll
    for (int q = 0; q < PER; ++q)  // the row's final node, for the later requests of the same object in this batch
        if (valid[q] && first[q]) hpos[slot[q]] = claim[q] ? r[q] : c[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        if (!valid[q]) continue;
        u32 nd, fl;
        if (claim[q]) { nd = r[q]; fl = 2u | 0x100u; }  // PLACED (+ "this request writes the table")
        else {
            nd = hpos[slot[q]];  // (its own current node, or what the first request decided)
            fl = (nd == kNone) ? 4u : (nd == r[q] ? 0u : 1u);
        }
        res[tid * PER + (u32)q] = make_uint2(nd, fl);
    }
    // the requesters' new `used`: one plain read-modify-write per requester that was claimed (this workgroup is alone)
    for (u32 j = tid; j < m && j < kPpTot; j += kBlock) {
        const u64 t = s_tot[j];
        if (t) used[j] += t;
    }
    if (tid == 0) { st[1] = 0u; *status = 0u; }
}

__global__ __launch_bounds__(256) void k_pp_apply(u32* __restrict__ assign, const uint4* __restrict__ rec,
                                                  const uint2* __restrict__ res, u32 n, u32* __restrict__ out_node,
                                                  u32* __restrict__ out_flag, const u32* __restrict__ st,
                                                  u32* __restrict__ aff_life, unsigned int* ticket, u32* done, u32 seq) {
    const u32 k = blockIdx.x * 256u + threadIdx.x;
    if (st[1] == 0u && k < n) {
        const uint2 x = res[k];
        out_node[k] = x.x;
        out_flag[k] = x.y & 0xFFu;
        if (x.y & 0x100u) {  // first touch (service.rs:244-252)
            const uint4 a = rec[k];
            assign[a.x] = a.y;
            if (aff_life) aff_life[a.x] = a.y;  // row lifecycle: the object exists from its first touch, its home is the requester
        }
    }
    if (done) signal_done_grid(ticket, done, seq);
}

// ------------------------------------------------------------------------------------------------
// Row-sharded solve: rank r owns the contiguous rows [off_r, off_r + n_r); shard order
// = index order, node tables are replicated.  The only cross-rank data are M-vectors:
//   exchange #1  X_r = [kept_local[m] | claim_local[m] | 8 counters]          (every solve)
//   exchange #2+ Y_r = [admitted-load delta[m] | spill load | spill rows]     (fix-up path only)
// all-gathered by the caller (RCCL over xGMI; ≤ 64 KiB per rank, latency-bound).  These kernels turn
// the gathered records into the local solver's view; every reduction is an integer sum in rank
// order, so the composed result is bit-identical to the unsharded solve.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_shard_pack1(const u64* __restrict__ used_kept,
                                                        const u64* __restrict__ claim_tot,
                                                        const u64* __restrict__ partial, u32 nb, u32 m,
                                                        u64* __restrict__ X) {
    __shared__ u64 red[8];
    const int tid = threadIdx.x;
    if (tid < 8) red[tid] = 0;
    __syncthreads();
    for (u32 j = tid; j < m; j += kBlock) {
        X[j] = used_kept[j];
        X[m + j] = claim_tot[j];
    }
    const u32 c = tid & 7;
    u64 acc = 0;
    for (u32 r = tid >> 3; r < nb; r += kBlock / 8) acc += partial[(size_t)r * 8 + c];
    acc += shfl_xor64(acc, 8); acc += shfl_xor64(acc, 16); acc += shfl_xor64(acc, 32);
    if ((tid & 63) < 8) atomicAdd(&red[c], acc);
    __syncthreads();
    if (tid < 8) X[2 * (size_t)m + tid] = red[tid];
}

// verdict words: 0 cut nodes (global) | 1 spill-candidate rows (global) | 2 this rank has rows to re-mark |
//                3 kept rows | 4 evicted rows | 5 claimant rows | 6 kept load | 7 claim load (all global)
// --- peer-to-peer exchange over xGMI: payload + sequence flag stored straight into every peer's window ---
// Window memory is uncached / fine-grained (hipExtMallocWithFlags) and IPC-mapped by the peers; every access is a
// system-scope 8-byte atomic, the flag is published after a system-scope release, and consumed with ONE relaxed poll
// loop + ONE system-scope acquire (cdna_hip_programming.md Guideline 16, lifted from agent to system scope).
#define RIOGP_SYS_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#define RIOGP_SYS_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
constexpr u64 kP2PTimeoutTicks = 300000000ull;  // wall_clock64 runs at 100 MHz: 3 s, then the error word is set

// all threads of the block call this; thread r < R waits for peer r's flag of this step
__device__ __forceinline__ void p2p_wait(const u64* flags, u32 R, u64 seq, u64* err) {
    if (threadIdx.x < R) {
        const u64* f = flags + (size_t)threadIdx.x * 8;
        const u64 t0 = wall_clock64();
        for (u32 tries = 0; RIOGP_SYS_LOAD(f) != seq; ++tries) {
            // (a peer that is gone: the first wait of the chain runs into the time-out and raises the word; every wait behind it
            //  — the rest of this tick's exchanges, the ticks enqueued after it — gives up as soon as it sees the word)
            if (wall_clock64() - t0 > kP2PTimeoutTicks || ((tries & 63u) == 63u && RIOGP_SYS_LOAD(err) != 0)) {
                RIOGP_SYS_STORE(err, 1ull);
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
}

// block b delivers this rank's record to peer b: words of payload, then the flag
__global__ __launch_bounds__(256) void k_p2p_put(const u64* __restrict__ src, u32 words, u64* const* __restrict__ peers,
                                                 size_t data_off, size_t flag_off, u64 seq) {
    u64* base = peers[blockIdx.x];
    u64* dst = base + data_off;
    for (u32 i = threadIdx.x; i < words; i += 256) RIOGP_SYS_STORE(dst + i, src[i]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) RIOGP_SYS_STORE(base + flag_off, seq);
}

// wait for every rank's record of this step, then copy them out of the window: out[r][words]
__global__ __launch_bounds__(kBlock) void k_p2p_wait_copy(const u64* __restrict__ win_slot, size_t W, u32 R, u32 words,
                                                          const u64* __restrict__ flags, u64 seq, u64* __restrict__ err,
                                                          u64* __restrict__ out) {
    p2p_wait(flags, R, seq, err);
    for (u32 r = 0; r < R; ++r)
        for (u32 i = threadIdx.x; i < words; i += kBlock)
            if (out) out[(size_t)r * words + i] = RIOGP_SYS_LOAD(win_slot + (size_t)r * W + i);
}

// K2x k_resolve_xchg — the whole fast-path exchange of the row-sharded solve in ONE launch over peer-to-peer windows.
//     Node tables couple the ranks only node by node, so every workgroup exchanges and resolves ITS OWN node group
//     (k_resolve's 8 nodes): local column sums of H for 8 nodes x {kept, claim} + its slice of the row counters go
//     straight into every peer's window row of this rank as DATA-TAGGED granules — each u64 travels as two 8-byte
//     words {low half | tag}, {high half | tag}, tag = the step's sequence number — so there is no flag, no store
//     drain and no release fence on the path: a consumer polls the very words it needs until both carry this step's
//     tag (an 8-byte store is delivered whole; a stale word carries the tag of four steps ago).  Then the nodes are
//     resolved against the global sums (k_shard_import's per-node maths) and one partial verdict row goes to the host
//     slot (the host adds the rows up, as after k_resolve).  No workgroup waits for another one of its own rank.
//     Row layout (u64 words, row stride W >= shard_xchg_words(m)): value v of {kept[m] | claim[m] | nb x 8 counters}
//     at words 2v, 2v+1.
__device__ __forceinline__ void xchg_put(u64* row, size_t v, u64 x, u32 tag) {
    RIOGP_SYS_STORE(row + 2 * v, (x & 0xFFFFFFFFull) | ((u64)tag << 32));
    RIOGP_SYS_STORE(row + 2 * v + 1, (x >> 32) | ((u64)tag << 32));
}
__device__ __forceinline__ u64 xchg_get(const u64* row, size_t v, u32 tag, u64* err) {
    const u64 t0 = wall_clock64();
    for (int tries = 0;; ++tries) {
        const u64 g0 = RIOGP_SYS_LOAD(row + 2 * v), g1 = RIOGP_SYS_LOAD(row + 2 * v + 1);
        if ((u32)(g0 >> 32) == tag && (u32)(g1 >> 32) == tag) return (g0 & 0xFFFFFFFFull) | (g1 << 32);
        if (wall_clock64() - t0 > kP2PTimeoutTicks || ((tries & 63) == 63 && RIOGP_SYS_LOAD(err) != 0)) {  // (p2p_wait's comment)
            RIOGP_SYS_STORE(err, 1ull);
            return 0;
        }
        // back off: tens of thousands of lanes polling uncached memory compete with the very stores they wait for
        if (tries < 8) __builtin_amdgcn_s_sleep(2);
        else if (tries < 32) __builtin_amdgcn_s_sleep(8);
        else __builtin_amdgcn_s_sleep(32);
    }
}

constexpr int kXchgVals = 3 * kResNodes;  // per workgroup: 8 kept sums | 8 claim sums | 8 counters

__global__ __launch_bounds__(256) void k_resolve_xchg(const u64* __restrict__ H, const u64* __restrict__ blkstat, Plan p,
                                                      u64* const* __restrict__ peers, u32 R, u32 rank,
                                                      size_t my_row_off /* of THIS rank's row, in every window */,
                                                      const u64* __restrict__ win_rows /* row 0 of the slot, own window */,
                                                      size_t W, u64 seq, u64* __restrict__ p2p_err,
                                                      const u64* __restrict__ cap, const u32* __restrict__ alive_bits,
                                                      u64* __restrict__ used_kept, u64* __restrict__ used_cur,
                                                      u64* __restrict__ claim_tot, u32* __restrict__ cutblk,
                                                      u32* __restrict__ cutidx, u64* __restrict__ gprev,
                                                      u64* __restrict__ gfinal, u32* __restrict__ forced_bits,
                                                      u64* __restrict__ rank_base, u64* __restrict__ partial,
                                                      u64* __restrict__ host_partial, const u32 nb /* node groups */,
                                                      DevStats* __restrict__ stats) {
    // gridDim.x == nb on a GPU of its own: one node group per workgroup, nobody waits for a workgroup of its own rank.
    // Ranks that SHARE a device get a bounded grid (launch_resolve_xchg) and every workgroup walks its node groups: all
    // sends first, then the waits — the spinning workgroups of all co-resident ranks then fit the chip next to the scan
    // of a rank that is still on its way, whatever the number of nodes.
    __shared__ u64 part[kResRowGroups][16];
    __shared__ u64 vals[kXchgVals];      // what this workgroup sends (the first 16 are resolve_column_sums' tot[])
    __shared__ u64 got[32][kXchgVals];   // the same values of every rank (R <= 32)
    __shared__ u64 outc[8];              // partial verdict row
    __shared__ u32 nib;
    const int tid = threadIdx.x, lane = tid & 63;
    const u32 m = p.m, G = p.G;
    const u32 tag = (u32)seq;
    for (u32 b = blockIdx.x; b < nb; b += gridDim.x) {  // ---- send
        // value index of word w (0..23) of node group b in a row
        auto vidx = [&](u32 w) -> size_t {
            return w < 8 ? (size_t)b * kResNodes + w : w < 16 ? (size_t)m + (size_t)b * kResNodes + (w - 8)
                                                              : 2 * (size_t)m + (size_t)b * 8 + (w - 16);
        };
        auto wvalid = [&](u32 w) -> bool { return w >= 16 || b * kResNodes + (w & 7) < m; };
        if (tid >= 16 && tid < kXchgVals) vals[tid] = tid == kXchgVals - 1 ? 1ull : 0ull;
        u64 v[kResRows][2];
        resolve_column_sums(H, b, G, v, part, vals);   // vals[0..7] kept sums, vals[8..15] claim sums
        if (tid >= 64 && tid < 128) {  // slice of the k_scan row counters: rows b, b+nb, ... of blkstat
            u64 acc = 0;
            for (u32 r = b + nb * (lane >> 2); r < G; r += nb * 16) acc += blkstat[(size_t)r * 4 + (lane & 3)];
            acc += shfl_xor64(acc, 4); acc += shfl_xor64(acc, 8); acc += shfl_xor64(acc, 16); acc += shfl_xor64(acc, 32);
            if (lane < 4) vals[16 + 3 + lane] = acc;  // 3 kept  4 evicted  5 claimants  6 spill candidates (rows)
        }
        if (tid == 0) {
            u64 a = 0, bb = 0;
            for (int q = 0; q < kResNodes; ++q)
                if (b * kResNodes + q < m) { a += vals[q]; bb += vals[q + kResNodes]; }
            vals[16 + 0] = a;   // load kept
            vals[16 + 1] = bb;  // load claimed
        }
        __syncthreads();
        if (tid < 8) partial[(size_t)b * 8 + tid] = vals[16 + tid];  // this shard's own counters (rio_gp_shard_finish)
        // this node group's values into this rank's row of EVERY window
        for (u32 i = tid; i < (u32)kXchgVals * R; i += 256) {
            const u32 w = i % kXchgVals;
            if (wvalid(w)) xchg_put(peers[i / kXchgVals] + my_row_off, vidx(w), vals[w], tag);
        }
        __syncthreads();  // vals / part go round again
    }
    for (u32 b = blockIdx.x; b < nb; b += gridDim.x) {  // ---- receive and resolve
        auto vidx = [&](u32 w) -> size_t {
            return w < 8 ? (size_t)b * kResNodes + w : w < 16 ? (size_t)m + (size_t)b * kResNodes + (w - 8)
                                                              : 2 * (size_t)m + (size_t)b * 8 + (w - 16);
        };
        auto wvalid = [&](u32 w) -> bool { return w >= 16 || b * kResNodes + (w & 7) < m; };
        if (tid < 8) outc[tid] = 0;
        if (tid == 0) nib = 0;
        // the same values of every rank, polled word by word
        for (u32 i = tid; i < (u32)kXchgVals * R; i += 256) {
            const u32 w = i % kXchgVals, r = i / kXchgVals;
            got[r][w] = wvalid(w) ? xchg_get(win_rows + (size_t)r * W, vidx(w), tag, p2p_err) : 0ull;
        }
        __syncthreads();
        // resolve this group's nodes against the global sums (k_shard_import's maths, node by node)
        if (tid < kResNodes) {
            const u32 jn = b * kResNodes + tid;
            if (jn < m) {
                u64 kept_glob = 0, claim_pre = 0, claim_glob = 0, claim_local = 0;
                for (u32 r = 0; r < R; ++r) {
                    const u64 kx = got[r][tid], cx = got[r][kResNodes + tid];
                    kept_glob += kx;
                    if (r < rank) claim_pre += cx;
                    if (r == rank) claim_local = cx;
                    claim_glob += cx;
                }
                const u64 cj = cap[jn];
                const u64 fre = (bit_of(alive_bits, jn) && cj > kept_glob) ? cj - kept_glob : 0;
                const bool forced = claim_pre > fre;  // the node's prefix overflowed on a lower rank: everyone here is rejected
                const u64 ukp = kept_glob + (forced ? fre : claim_pre);
                used_kept[jn] = ukp;
                claim_tot[jn] = claim_local;
                used_cur[jn] = ukp + claim_local;
                cutblk[jn] = kNoCut;
                cutidx[jn] = kNoCut;
                gprev[jn] = kept_glob;
                gfinal[jn] = kept_glob + claim_glob;
                if (forced) atomicOr(&nib, 1u << tid);
                if (claim_glob > fre) atomicAdd(&outc[0], 1ull);                          // cut nodes (global)
                if (forced || claim_local > fre - claim_pre) atomicAdd(&outc[2], 1ull);   // this rank has a local fix-up
            }
        } else if (tid >= 16 && tid < kXchgVals) {  // global counters: word tid of this group's slice, summed over the ranks
            const int k = tid - 16;
            u64 sum = 0;
            for (u32 r = 0; r < R; ++r) sum += got[r][tid];
            // verdict row: 0 cut nodes 1 spill rows 2 local fix-up 3 kept 4 evicted 5 claimants 6 load kept 7 load claimed
            const int dst = k == 0 ? 6 : k == 1 ? 7 : k == 3 ? 3 : k == 4 ? 4 : k == 5 ? 5 : k == 6 ? 1 : -1;
            if (dst >= 0) atomicAdd(&outc[dst], sum);
        }
        __syncthreads();
        if (tid == 0) {  // eight bits of the forced bitmap belong to this node group alone (8 | 32)
            const u32 w = (b * kResNodes) >> 5, sh = (b * kResNodes) & 31;
            atomicAnd(&forced_bits[w], ~(0xFFu << sh));
            if (nib) atomicOr(&forced_bits[w], nib << sh);
            if (b == 0) *rank_base = 0;
        }
        if (tid < 8) host_partial[(size_t)b * 8 + tid] = outc[tid];
        if (tid == 0 && outc[2]) atomicAdd(&stats->local_fixup, outc[2]);  // (zeroed by k_scan) the guard of the asynchronous tick's re-marking pass
        if (tid == 0 && (outc[0] | outc[1])) atomicAdd(&stats->global_slow, 1ull);  // cut nodes or spill rows anywhere: the same on every rank
        __syncthreads();  // got / outc / nib go round again
    }
}

// global resolve of the all-gathered X records (RCCL paths): one workgroup, every node
__global__ __launch_bounds__(kBlock) void k_shard_import(const u64* __restrict__ Xg, size_t W, u32 rank, u32 R, u32 m,
                                                         const u64* __restrict__ cap,
                                                         const u32* __restrict__ alive_bits,
                                                         u64* __restrict__ used_kept, u64* __restrict__ used_cur,
                                                         u64* __restrict__ claim_tot, u32* __restrict__ cutblk,
                                                         u32* __restrict__ cutidx, u64* __restrict__ gprev,
                                                         u64* __restrict__ gfinal, u32* __restrict__ forced_bits,
                                                         u64* __restrict__ rank_base, u64* __restrict__ verdict_dev,
                                                         u64* __restrict__ verdict_host) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32* fb = reinterpret_cast<u32*>(smem);  // [mwords]
    __shared__ u64 red[8];
    const int tid = threadIdx.x;
    const u32 mwords = (m + 31) / 32;
    for (u32 k = tid; k < mwords; k += kBlock) fb[k] = 0;
    if (tid < 8) red[tid] = 0;
    __syncthreads();
    u32 ncut = 0, need = 0;
    for (u32 j = tid; j < m; j += kBlock) {
        u64 kept_glob = 0, claim_pre = 0, claim_glob = 0, claim_local = 0;
        for (u32 r = 0; r < R; ++r) {
            const u64 kx = Xg[r * W + j], cx = Xg[r * W + m + j];
            kept_glob += kx;
            if (r < rank) claim_pre += cx;
            if (r == rank) claim_local = cx;
            claim_glob += cx;
        }
        const u64 cj = cap[j];
        const u64 fre = (bit_of(alive_bits, j) && cj > kept_glob) ? cj - kept_glob : 0;
        const bool forced = claim_pre > fre;  // the node's prefix overflowed on a lower rank: everyone here is rejected
        const u64 ukp = kept_glob + (forced ? fre : claim_pre);
        used_kept[j] = ukp;  // local view: free = cap - ukp = what is left for THIS rank's claimants
        claim_tot[j] = claim_local;
        used_cur[j] = ukp + claim_local;
        cutblk[j] = kNoCut;
        cutidx[j] = kNoCut;
        gprev[j] = kept_glob;
        gfinal[j] = kept_glob + claim_glob;  // the committed `used` when nothing is cut and nothing spills
        if (forced) atomicOr(&fb[j >> 5], 1u << (j & 31));
        ncut += claim_glob > fre;
        need += forced || claim_local > fre - claim_pre;
    }
    ncut = wave_sum32(ncut);
    need = wave_sum32(need);
    if ((tid & 63) == 0) {
        if (ncut) atomicAdd(&red[0], (u64)ncut);
        if (need) atomicAdd(&red[2], (u64)need);
    }
    if (tid < 8) {  // global counters: column tid of every rank's record
        u64 s = 0;
        for (u32 r = 0; r < R; ++r) s += Xg[r * W + 2 * (size_t)m + tid];
        // k_resolve's partial columns: 0 load_kept 1 load_claim 2 (local n_cut, unused) 3 kept 4 evicted 5 claimants 6 spillcand
        const int dst = tid == 0 ? 6 : tid == 1 ? 7 : tid == 3 ? 3 : tid == 4 ? 4 : tid == 5 ? 5 : tid == 6 ? 1 : -1;
        if (dst >= 0) atomicAdd(&red[dst], s);
    }
    __syncthreads();
    for (u32 k = tid; k < mwords; k += kBlock) forced_bits[k] = fb[k];
    if (tid == 0) *rank_base = 0;
    if (tid < 8) {
        verdict_dev[tid] = red[tid];
        if (verdict_host) verdict_host[tid] = red[tid];
    }
}

// forced nodes: every local claimant is rejected, load-0 ones included (the strict prefix cut already fired)
__global__ __launch_bounds__(kBlock) void k_shard_force(u32 m, const u32* __restrict__ forced_bits,
                                                        const u64* __restrict__ used_kept, u32* __restrict__ cutidx,
                                                        u64* __restrict__ used_cur) {
    for (u32 j = threadIdx.x; j < m; j += kBlock)
        if (bit_of(forced_bits, j)) {
            cutidx[j] = 0;
            used_cur[j] = used_kept[j];
        }
}

// Y = [used_cur - base | spill load still pending on this rank | its row count]
__global__ __launch_bounds__(kBlock) void k_shard_export_delta(const u64* __restrict__ used_cur,
                                                               const u64* __restrict__ base,
                                                               const u64* __restrict__ wsp_sum,
                                                               const u32* __restrict__ wsp_cnt, u32 nw, u32 m,
                                                               u64* __restrict__ Y) {
    __shared__ u64 red[2];
    const int tid = threadIdx.x;
    if (tid < 2) red[tid] = 0;
    __syncthreads();
    for (u32 j = tid; j < m; j += kBlock) Y[j] = used_cur[j] - base[j];
    u64 s = 0, c = 0;
    for (u32 w = tid; w < nw; w += kBlock) { s += wsp_sum[w]; c += wsp_cnt[w]; }
    s = wave_sum(s);
    c = wave_sum(c);
    if ((tid & 63) == 0) { atomicAdd(&red[0], s); atomicAdd(&red[1], c); }
    __syncthreads();
    if (tid < 2) Y[m + tid] = red[tid];
}

// used_cur = global used so far; rank_base = spill load pending on lower ranks; verdict = [rows, load] pending globally
__global__ __launch_bounds__(kBlock) void k_shard_import_delta(const u64* __restrict__ Yg, u32 rank, u32 R, u32 m,
                                                               u64* __restrict__ gprev, u64* __restrict__ used_cur,
                                                               u64* __restrict__ rank_base,
                                                               u64* __restrict__ verdict_dev,
                                                               u64* __restrict__ verdict_host) {
    const int tid = threadIdx.x;
    const size_t W = (size_t)m + 2;
    for (u32 j = tid; j < m; j += kBlock) {
        u64 g = gprev[j];
        for (u32 r = 0; r < R; ++r) g += Yg[r * W + j];
        gprev[j] = g;
        used_cur[j] = g;
    }
    if (tid == 0) {
        u64 base = 0, load = 0, rows = 0;
        for (u32 r = 0; r < R; ++r) {
            const u64 l = Yg[r * W + m];
            if (r < rank) base += l;
            load += l;
            rows += Yg[r * W + m + 1];
        }
        *rank_base = base;
        verdict_dev[0] = rows; verdict_dev[1] = load;
        if (verdict_host) { verdict_host[0] = rows; verdict_host[1] = load; }
    }
}

// Asynchronous row-sharded tick, one exchange of the fix-up record in two launches.  k_shard_export_put = k_shard_export_delta
// + k_p2p_put: Y = [used_cur - base | spill load still pending here | its rows] straight into this rank's row of every
// window, then the flag.  k_shard_wait_import = k_p2p_wait_copy + k_shard_import_delta: everyone's record out of the own
// window -> global `used`, this rank's spill base, rows pending anywhere.  When the solve needs no fix-up on ANY rank
// (DevStats::global_slow == 0, the same on every rank) neither stores to a peer nor waits for one: `used` = the fast path's
// global vector, nothing pending.
__global__ __launch_bounds__(kBlock) void k_shard_export_put(const u64* __restrict__ used_cur, const u64* __restrict__ base,
                                                             const u64* __restrict__ wsp_sum, const u32* __restrict__ wsp_cnt,
                                                             u32 nw, u32 m, u64* const* __restrict__ peers, u32 R, size_t data_off,
                                                             size_t flag_off, u64 seq, const DevStats* __restrict__ st) {
    __shared__ u64 red[2];
    if (st->global_slow == 0) return;
    const int tid = threadIdx.x;
    if (tid < 2) red[tid] = 0;
    __syncthreads();
    for (u32 j = tid; j < m; j += kBlock) {
        const u64 y = used_cur[j] - base[j];
        for (u32 r = 0; r < R; ++r) RIOGP_SYS_STORE(peers[r] + data_off + j, y);
    }
    u64 s = 0, c = 0;
    for (u32 w = tid; w < nw; w += kBlock) { s += wsp_sum[w]; c += wsp_cnt[w]; }
    s = wave_sum(s);
    c = wave_sum(c);
    if ((tid & 63) == 0) { atomicAdd(&red[0], s); atomicAdd(&red[1], c); }
    __syncthreads();
    if (tid < 2)
        for (u32 r = 0; r < R; ++r) RIOGP_SYS_STORE(peers[r] + data_off + m + tid, red[tid]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if ((u32)tid < R) RIOGP_SYS_STORE(peers[tid] + flag_off, seq);
}

__global__ __launch_bounds__(kBlock) void k_shard_wait_import(const u64* __restrict__ win_slot, size_t W,
                                                              const u64* __restrict__ flags, u64 seq, u64* __restrict__ err,
                                                              u32 rank, u32 R, u32 m, u64* __restrict__ gprev,
                                                              const u64* __restrict__ gfinal, u64* __restrict__ used_cur,
                                                              u64* __restrict__ rank_base, u64* __restrict__ verdict_dev,
                                                              u64* __restrict__ verdict_host, const DevStats* __restrict__ st) {
    const int tid = threadIdx.x;
    if (st->global_slow == 0) {  // fast path everywhere: nothing was sent, nothing is pending
        for (u32 j = tid; j < m; j += kBlock) used_cur[j] = gfinal[j];
        if (tid == 0) {
            *rank_base = 0;
            verdict_dev[0] = 0; verdict_dev[1] = 0;
            if (verdict_host) { verdict_host[0] = 0; verdict_host[1] = 0; }
        }
        return;
    }
    p2p_wait(flags, R, seq, err);
    for (u32 j = tid; j < m; j += kBlock) {
        u64 g = gprev[j];
        for (u32 r = 0; r < R; ++r) g += RIOGP_SYS_LOAD(win_slot + (size_t)r * W + j);
        gprev[j] = g;
        used_cur[j] = g;
    }
    if (tid == 0) {
        u64 base = 0, load = 0, rows = 0;
        for (u32 r = 0; r < R; ++r) {
            const u64 l = RIOGP_SYS_LOAD(win_slot + (size_t)r * W + m);
            if (r < rank) base += l;
            load += l;
            rows += RIOGP_SYS_LOAD(win_slot + (size_t)r * W + m + 1);
        }
        *rank_base = base;
        verdict_dev[0] = rows; verdict_dev[1] = load;
        if (verdict_host) { verdict_host[0] = rows; verdict_host[1] = load; }
    }
}

// Asynchronous row-sharded tick: this rank's counters of the tick into its pinned record — k_resolve_xchg's partial rows
// added up (0 load kept 1 load claimed 3 kept 4 evicted 5 claimants 6 spill candidates) | the fix-up counters of DevStats
// (8 rejected 9 its load 10 spilled 11 its load 12 unplaced 13 its load) | 15 the tick's mark
__global__ __launch_bounds__(64) void k_shard_tick_stats(const u64* __restrict__ partial, u32 nb, const DevStats* __restrict__ st,
                                                         u64* __restrict__ out, u64 mark) {
    const int tid = threadIdx.x;
    if (tid < 8) {
        u64 acc = 0;
        for (u32 r = 0; r < nb; ++r) acc += partial[(size_t)r * 8 + tid];
        out[tid] = acc;
    } else if (tid < 16) {
        const int k = tid - 8;
        out[tid] = k == 0 ? st->rejected : k == 1 ? st->load_rejected : k == 2 ? st->spilled : k == 3 ? st->load_spilled
                 : k == 4 ? st->unplaced : k == 5 ? st->load_unplaced : k == 7 ? mark : 0ull;
    }
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
static inline unsigned grid_for(u64 n, unsigned block, unsigned cap) {
    u64 g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

template <bool VIRT, bool AA, int TPI, int COMPACT = 0, bool NT = false, bool CHAIN = false>
static void launch_scan_t(const Plan& p, const Table& t, const NodeTab& nt, const SolveBufs& b, hipStream_t s,
                          hipEvent_t e0, hipEvent_t e1, const PackOut* pack = nullptr, const ScanChain* chain = nullptr) {
    const ScanChain ch = chain ? *chain : ScanChain{nullptr, nullptr, 0, 0, 0};
    const size_t lds = scan_lds_bytes(p.m) + (COMPACT == 2 ? (size_t)kWaves * 4 * kStageCap * sizeof(u32) : 0);
    const PackOut pko = pack ? *pack : PackOut{nullptr, nullptr, nullptr, nullptr, nullptr};
    Plan pp = p;
    pp.alive_dst = nt.alive_src ? const_cast<u32*>(nt.alive_bits) : nullptr;
    const u32* abits = nt.alive_src ? nt.alive_src : nt.alive_bits;
    if (e0 || e1)  // start / stop events taken from the dispatch packet itself: the kernel's own duration, or (stop event alone) the
                   // completion another stream waits for, without a marker packet behind the kernel
        hipExtLaunchKernelGGL((k_scan<VIRT, AA, TPI, COMPACT, NT, CHAIN>), dim3(p.G), dim3(kBlock), (uint32_t)lds, s, e0, e1, 0,
                              t.cur, t.load, t.aff, t.next, abits, pp, b.H, b.blkstat, b.wsp_sum[0], b.wsp_cnt[0],
                              b.stats, pko, b.fx, b.bsp_sum[0], b.bsp_cnt[0], b.R, b.RP, ch);
    else
        hipLaunchKernelGGL((k_scan<VIRT, AA, TPI, COMPACT, NT, CHAIN>), dim3(p.G), dim3(kBlock), lds, s, t.cur, t.load, t.aff,
                           t.next, abits, pp, b.H, b.blkstat, b.wsp_sum[0], b.wsp_cnt[0], b.stats, pko, b.fx, b.bsp_sum[0],
                           b.bsp_cnt[0], b.R, b.RP, ch);
}

// Tiles per wave-iteration: 2 on the plain whole-table scan (measured +2.5 % over 1; 4 loses to its unpipelined tail),
// 1 where the row body is heavier (packing) or the table is small (virtual table of place_pending).
// Non-temporal streams once the four columns no longer fit the 256 MiB Infinity Cache.
constexpr u64 kScanNtRows = (u64)20 << 20;  // 16 B/row * 20 Mi rows = 320 MiB of columns
int g_scan_nt_mode = 0;  // 0 by size | 1 always | 2 never (lab builds: rio_gp_debug_set_scan_nt, A/B runs)
int g_scan_stage = 1;    // packing through LDS rings (0: straight from registers; lab builds, A/B runs)
int g_inc_tpi = 2;       // tiles per wave-iteration of k_inc_scan: 1 | 2 | 4 (lab builds: bits 5-6 of rio_gp_debug_set_scan_nt, A/B runs)
void set_scan_nt(int mode) {
    g_scan_nt_mode = mode & 3;
    g_scan_stage = (mode & 16) ? 0 : 1;
    g_inc_tpi = ((mode >> 5) & 3) == 1 ? 1 : ((mode >> 5) & 3) == 2 ? 4 : 2;
}

// Two launches of the chain are in flight at any time and the later one's workgroups WAIT, resident, for the earlier one's:
// that cannot deadlock only if both fit the chip at once — two workgroups per CU (64 VGPRs: __launch_bounds__(kBlock, 8)).
// Tiles per wave-iteration of the chained scan: 1 (measured, same run: 26.7-27.0 us per tick against 28.9-29.4 with 2, whose
// 64-register form waits for a group's loads before it requests the next one)
static int chain_tpi() {
#ifdef RIO_GP_LAB
    static const int v = [] { const char* e = getenv("RIO_GP_CHAIN_TPI"); return e && atoi(e) == 2 ? 2 : 1; }();
    return v;
#else
    return 1;
#endif
}
bool scan_chain_fits(u32 m) {
    const size_t lds = scan_lds_bytes(m);
    int nb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipError_t e = hipSuccess;
#define RIOGP_OCC(K, AA, TPI_, NT_) if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb[K], k_scan<false, AA, TPI_, 0, NT_, true>, kBlock, lds)
    RIOGP_OCC(0, true, 2, false); RIOGP_OCC(1, false, 2, false); RIOGP_OCC(2, true, 2, true); RIOGP_OCC(3, false, 2, true);
    RIOGP_OCC(4, true, 1, false); RIOGP_OCC(5, false, 1, false); RIOGP_OCC(6, true, 1, true); RIOGP_OCC(7, false, 1, true);
#undef RIOGP_OCC
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    for (int k = 0; k < 8; ++k) if (nb[k] < 2) return false;
    return true;
}

void launch_scan(const Plan& p, const Table& t, const NodeTab& nt, const SolveBufs& b, bool virt, bool all_alive,
                 hipStream_t s, hipEvent_t e0, hipEvent_t e1, const PackOut* pack, const ScanChain* chain) {
    if (chain && !pack && !virt) {  // a chained quiet tick: the plain whole-table scan, handed over workgroup by workgroup
        const bool ntl = g_scan_nt_mode == 1 || (g_scan_nt_mode == 0 && p.n >= kScanNtRows);
#define RIOGP_CH(TPI_) do { \
            if (ntl) { if (all_alive) launch_scan_t<false, true, TPI_, 0, true, true>(p, t, nt, b, s, e0, e1, nullptr, chain); \
                       else launch_scan_t<false, false, TPI_, 0, true, true>(p, t, nt, b, s, e0, e1, nullptr, chain); } \
            else { if (all_alive) launch_scan_t<false, true, TPI_, 0, false, true>(p, t, nt, b, s, e0, e1, nullptr, chain); \
                   else launch_scan_t<false, false, TPI_, 0, false, true>(p, t, nt, b, s, e0, e1, nullptr, chain); } } while (0)
        if (chain_tpi() == 1) RIOGP_CH(1); else RIOGP_CH(2);
#undef RIOGP_CH
        return;
    }
    if (pack && !virt) {  // k_scan that also packs the pending rows of every wave (adaptive fix-up, rio_gp_capi.hip)
        // through per-wave LDS rings when they fit next to the histograms (m up to ~4 800), straight from registers else
        const bool staged = g_scan_stage && scan_lds_bytes(p.m) + (size_t)kWaves * 4 * kStageCap * sizeof(u32) <= (size_t)160 * 1024;
        if (staged) {
            if (all_alive) launch_scan_t<false, true, 1, 2>(p, t, nt, b, s, e0, e1, pack);
            else launch_scan_t<false, false, 1, 2>(p, t, nt, b, s, e0, e1, pack);
        } else {
            if (all_alive) launch_scan_t<false, true, 1, 1>(p, t, nt, b, s, e0, e1, pack);
            else launch_scan_t<false, false, 1, 1>(p, t, nt, b, s, e0, e1, pack);
        }
        return;
    }
    if (virt) {
        if (t.pk_idx && t.real_next) {  // big place_pending batch: first touches also go straight into the real column
            // (load: a flag, never dereferenced — the window kernel has written the alive requesters' first touches into the real
            //  column already; aff: the virtual table as 8-byte records, split into t.cur / t.load by this scan)
            const PackOut sc{const_cast<u32*>(t.pk_idx), t.prewritten ? reinterpret_cast<u32*>(uintptr_t(16)) : nullptr,
                             const_cast<u32*>(reinterpret_cast<const u32*>(t.vrec)), t.real_next, reinterpret_cast<u32*>(&b.stats->err)};
            if (all_alive) launch_scan_t<true, true, 1, 3>(p, t, nt, b, s, e0, e1, &sc);
            else launch_scan_t<true, false, 1, 3>(p, t, nt, b, s, e0, e1, &sc);
            return;
        }
        const PackOut guard{nullptr, nullptr, nullptr, nullptr, const_cast<u32*>(t.skip_if)};  // (wcnt: solve an empty table when set)
        if (all_alive) launch_scan_t<true, true, 1>(p, t, nt, b, s, e0, e1, &guard);
        else launch_scan_t<true, false, 1>(p, t, nt, b, s, e0, e1, &guard);
        return;
    }
    if (g_scan_nt_mode == 1 || (g_scan_nt_mode == 0 && p.n >= kScanNtRows)) {
        if (all_alive) launch_scan_t<false, true, 2, false, true>(p, t, nt, b, s, e0, e1);
        else launch_scan_t<false, false, 2, false, true>(p, t, nt, b, s, e0, e1);
        return;
    }
    if (all_alive) launch_scan_t<false, true, 2>(p, t, nt, b, s, e0, e1);
    else launch_scan_t<false, false, 2>(p, t, nt, b, s, e0, e1);
}

// The scan of a committed tick over a mostly-placed table (k_inc_scan): the assignment column is read and updated in place,
// the pending rows go to `pack` (per wave range of p); launch_rebal deals them out evenly and builds the fix-up's histograms.
static size_t inc_lds_bytes(u32 m) { return inc_lds_base(m) + (size_t)kWaves * 3 * kIncCap * sizeof(u32); }
bool inc_scan_fits(u32 m) { return inc_lds_bytes(m) <= (size_t)160 * 1024; }
void launch_inc_scan(const Plan& p, u32* assign, const u32* load, const u32* aff, const NodeTab& nt, const SolveBufs& b,
                     const PackOut& pack, hipStream_t s) {
    const size_t lds = inc_lds_bytes(p.m);
    Plan pp = p;
    pp.wcnt = nullptr;
    pp.alive_dst = nt.alive_src ? const_cast<u32*>(nt.alive_bits) : nullptr;
    const u32* abits = nt.alive_src ? nt.alive_src : nt.alive_bits;
    const bool ntl = g_scan_nt_mode == 1 || (g_scan_nt_mode == 0 && p.n >= kScanNtRows);
#define RIOGP_INC(TPI_, NT_) hipLaunchKernelGGL((k_inc_scan<TPI_, NT_>), dim3(p.G), dim3(kBlock), lds, s, assign, load, aff, abits, pp, \
                                                b.blkstat, b.stats, pack, b.fx, b.R, b.RP)
    if (g_inc_tpi == 1) { if (ntl) RIOGP_INC(1, true); else RIOGP_INC(1, false); }
    else if (g_inc_tpi == 4) { if (ntl) RIOGP_INC(4, true); else RIOGP_INC(4, false); }
    else { if (ntl) RIOGP_INC(2, true); else RIOGP_INC(2, false); }
#undef RIOGP_INC
}
// The balanced table of the rows k_inc_scan packed: uniform wave ranges of ceil(tiles / nw) tiles each, as many
// waves and blocks as the real table's plan (the columns hold rebal_rows(p) rows + the usual padding).
Plan rebal_plan(const Plan& p) {
    const u64 ctiles = (p.tiles + p.nw - 1) / p.nw;
    Plan pv = make_plan((u64)p.nw * ctiles * kTile, p.m, p.G);
    pv.mark = p.mark;
    pv.sa = p.sa;
    return pv;
}
u64 rebal_rows(u64 n) {
    const Plan p = make_plan(n, 1, 0);
    return (u64)p.nw * ((p.tiles + p.nw - 1) / p.nw) * kTile;
}
void launch_rebal(const Plan& p, const Plan& pv, const PackOut& src, const NodeTab& nt, const PackOut& dst, const SolveBufs& b,
                  hipStream_t s) {
    hipLaunchKernelGGL(k_rebal, dim3(p.G), dim3(kBlock), rebal_lds_bytes(p.m, p.nw), s, src.idx, src.load, src.aff, src.wcnt, p, pv,
                       nt.alive_bits, dst.idx, dst.load, dst.aff, dst.next, dst.wcnt, b.H, b.wsp_sum[0], b.wsp_cnt[0],
                       b.bsp_sum[0], b.bsp_cnt[0]);
}

unsigned resolve_blocks(u32 m) {
    unsigned g = (m + kResNodes - 1) / kResNodes;
    return g ? g : 1;
}

void launch_resolve(const Plan& p, const NodeTab& nt, const SolveBufs& b, u64* host_partial, hipStream_t s,
                    hipEvent_t e0, hipEvent_t e1, const PackOut* search, u64* fold_into, u32 fold_rounds, const u64* kept_from) {
    const unsigned grid = resolve_blocks(p.m);
    ResolveArgs a;
    a.H = b.H; a.blkstat = b.blkstat; a.p = p;
    a.cap = nt.cap; a.alive_bits = nt.alive_bits; a.used_base = nt.used_base;
    a.used_kept = b.used_kept; a.used_cur = b.used_cur; a.claim_tot = b.claim_tot; a.cutblk = b.cutblk; a.cutidx = b.cutidx;
    a.partial = b.partial; a.host_partial = host_partial; a.budget = b.budget; a.admpre = b.admpre; a.stats = b.stats;
    a.RP = b.RP; a.D = b.D; a.fold_into = b.D ? fold_into : nullptr; a.fold_rounds = fold_rounds;
    a.pk_aff = search ? search->aff : nullptr; a.pk_load = search ? search->load : nullptr;
    a.Tg = b.Tg;
    a.kept_from = kept_from;
    const bool srch = search != nullptr && p.wcnt != nullptr;
    if (e0 && e1) {
        if (srch) hipExtLaunchKernelGGL(k_resolve<true>, dim3(grid), dim3(kBlock), 0, s, e0, e1, 0, a);
        else hipExtLaunchKernelGGL(k_resolve<false>, dim3(grid), dim3(256), 0, s, e0, e1, 0, a);
    } else {
        if (srch) hipLaunchKernelGGL(k_resolve<true>, dim3(grid), dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL(k_resolve<false>, dim3(grid), dim3(256), 0, s, a);
    }
}

// dynamic LDS of k_cut_find: the fixed tables + as many u64 words of T region as fit (T rows [K][S|1] + 3 state words
// per node of a group; every local node in one group at the finest fan-out needs m * ((255|1) + 3) words)
static size_t cut_find_lds(const Plan& p, u32* tcap_out) {
    const size_t fixed = cut_fused_fixed(p.m, p.mwords);
    size_t slots = (150 * 1024 - fixed) / sizeof(u64);
    const size_t most = (size_t)p.m * 259;
    if (slots > most) slots = most;
    if (slots < 64) slots = 64;
    *tcap_out = (u32)slots;
    return fixed + slots * sizeof(u64);
}

// The exact cut search as a launch of its own: k_cut_find, spread over the chip (whole-table solves, the virtual table of
// place_pending, the row-sharded solve; the packed fix-up searches inside k_resolve).  have_cutblk: launch_resolve of the
// same solve (same bufs) has already located the cut blocks; else k_cutblk does (row-sharded path: the global resolve
// changed the free capacities after the local sums).  Guards itself on the device (stats->n_cut): cheap to enqueue
// speculatively behind a solve whose verdict the host has not read yet.
void launch_cut_find(const Plan& p, const Table& t, const NodeTab& nt, const SolveBufs& b, bool virt, hipStream_t s,
                     bool have_cutblk) {
    const unsigned gcb = (p.m + kCbNodes - 1) / kCbNodes;
    if (!have_cutblk)
        hipLaunchKernelGGL(k_cutblk, dim3(gcb ? gcb : 1), dim3(256), 0, s, b.H, p, nt.cap, nt.alive_bits, b.used_kept,
                           b.claim_tot, b.cutblk, b.budget, b.admpre, b.stats);
    u32 tcap = 0;
    const size_t ldsf = cut_find_lds(p, &tcap);
    u64* R = have_cutblk ? b.R : nullptr;  // (k_cutblk does not maintain R: the row-sharded path does not use it)
    if (virt)
        hipLaunchKernelGGL(k_cut_find<true>, dim3(kCutFindGrid), dim3(kBlock), ldsf, s, t.cur, t.load, t.aff, nt.alive_bits, p,
                           b.cutblk, b.budget, b.admpre, b.used_kept, b.forced_bits, b.cutidx, b.used_cur, b.stats, tcap, R);
    else
        hipLaunchKernelGGL(k_cut_find<false>, dim3(kCutFindGrid), dim3(kBlock), ldsf, s, t.cur, t.load, t.aff, nt.alive_bits, p,
                           b.cutblk, b.budget, b.admpre, b.used_kept, b.forced_bits, b.cutidx, b.used_cur, b.stats, tcap, R);
}

static size_t fill_lds_bytes(u32 m, u32 mwords, bool pack) { return fill_lds_off_x(m) + fill_lds_x(m, mwords, pack) + 16; }
bool fill_can_pack(u32 m) { return fill_lds_bytes(m, (m + 31) / 32, true) <= (size_t)160 * 1024; }

// One launch of k_fill:
//   apply && fill   round 0 of a solve: re-mark + (pack) + water-fill
//   fill            a later round
//   apply           the row-sharded solve's cut step (the Y exchange sits between it and the rounds)
// The round orders the nodes by used_cur + D[0..round) and adds what it admits into D[round] — or, row-sharded solve
// (b.used_snap set, b.D == nullptr), orders them by the snapshot and adds into used_cur.
void launch_fill(const Plan& p, const Table& t, const NodeTab& nt, const SolveBufs& b, bool virt, bool apply, bool fill,
                 int round, bool last, hipStream_t s, const PackOut* pack) {
    const int in = (apply ? 0 : (round & 1)), out = fill ? (in ^ 1) : 0;
    FillArgs a;
    a.cur = t.cur; a.load = t.load; a.aff = t.aff; a.next = t.next;
    a.alive_bits = nt.alive_bits;
    a.p = p;
    a.cutidx = b.cutidx; a.forced_bits = b.forced_bits;
    a.cap = nt.cap; a.round = (u32)round;
    if (b.D) { a.Uread = b.used_cur; a.D = b.D; a.Uadd = b.D + (size_t)round * p.m; }
    else { a.Uread = b.used_snap ? b.used_snap : b.used_cur; a.D = nullptr; a.Uadd = b.used_cur; }
    a.wsp_sum_in = b.wsp_sum[in]; a.wsp_cnt_in = b.wsp_cnt[in]; a.wsp_sum_out = b.wsp_sum[out]; a.wsp_cnt_out = b.wsp_cnt[out];
    a.bsp_sum_in = b.bsp_sum[in]; a.bsp_cnt_in = b.bsp_cnt[in]; a.bsp_sum_out = b.bsp_sum[out]; a.bsp_cnt_out = b.bsp_cnt[out];
    a.R = b.R; a.RP = b.RP; a.ng = resolve_blocks(p.m);
    a.last = last ? ((t.none_prewritten || pack) ? 2 : 1) : 0;
    a.stats = b.stats;
    a.pk_idx = t.pk_idx; a.real_next = pack ? t.next : t.real_next;
    a.rank_base = b.rank_base; a.pending_global = b.pending_global; a.run_if = b.run_if;
    a.fx = b.fx;
    a.pko = pack ? *pack : PackOut{nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t lds = fill_lds_bytes(p.m, p.mwords, pack != nullptr);
    if (apply && fill) {
        if (virt) hipLaunchKernelGGL((k_fill<true, true, true, false>), dim3(p.G), dim3(kBlock), lds, s, a);
        else if (pack) hipLaunchKernelGGL((k_fill<false, true, true, true>), dim3(p.G), dim3(kBlock), lds, s, a);
        else hipLaunchKernelGGL((k_fill<false, true, true, false>), dim3(p.G), dim3(kBlock), lds, s, a);
    } else if (fill) {
        hipLaunchKernelGGL((k_fill<false, false, true, false>), dim3(p.G), dim3(kBlock), lds, s, a);
    } else {
        if (virt) hipLaunchKernelGGL((k_fill<true, true, false, false>), dim3(p.G), dim3(kBlock), lds, s, a);
        else hipLaunchKernelGGL((k_fill<false, true, false, false>), dim3(p.G), dim3(kBlock), lds, s, a);
    }
}

// k_cut_apply + k_cut_settle: exact cuts + re-marking (+ packing) of a whole-table solve of the REAL table in one pass.  pack:
// the rows that go on to the water-fill are packed into pk (the rounds then run over pk with Plan::wcnt = pk.wcnt); else pk is
// only the scratch of the undecided rows and the rounds run over the table.  ul: four more scratch columns + one count per wave
// range (the undecided rows' list); Tg: [m][16] u64, zeroed for every cut node by k_resolve (ResolveArgs::Tg).  The rounds that
// follow are k_fill<FILL> from round 0 on.
bool cut_apply_fits(u32 m) { return m >= 1 && ca_lds_bytes(m, (m + 31) / 32) <= (size_t)150 * 1024; }
void launch_cut_apply(const Plan& p, const Table& t, const NodeTab& nt, const SolveBufs& b, const PackOut& pk, const PackOut& ul,
                      u64* Tg, bool pack, bool all_alive, hipStream_t s) {
    CutApplyArgs a;
    a.cur = t.cur; a.load = t.load; a.aff = t.aff; a.next = t.next;
    a.alive_bits = nt.alive_bits;
    a.p = p;
    a.p.wcnt = nullptr;
    a.cutblk = b.cutblk;
    a.Tg = Tg;
    a.wsp_sum_out = b.wsp_sum[0]; a.wsp_cnt_out = b.wsp_cnt[0];
    a.bsp_cnt_in = b.bsp_cnt[0];
    a.stats = b.stats; a.fx = b.fx;
    a.pko = pk;
    a.ul = UndList{ul.idx, ul.load, ul.aff, ul.next, ul.wcnt};
    const size_t lds = ca_lds_bytes(p.m, p.mwords);
    const unsigned grid = p.G;       // (two workgroups per CU are not co-resident: ~100 SGPRs cap a SIMD at 6 waves)
    if (pack) {
        if (all_alive) hipLaunchKernelGGL((k_cut_apply<true, true>), dim3(grid), dim3(kBlock), lds, s, a);
        else hipLaunchKernelGGL((k_cut_apply<true, false>), dim3(grid), dim3(kBlock), lds, s, a);
    } else {
        if (all_alive) hipLaunchKernelGGL((k_cut_apply<false, true>), dim3(grid), dim3(kBlock), lds, s, a);
        else hipLaunchKernelGGL((k_cut_apply<false, false>), dim3(grid), dim3(kBlock), lds, s, a);
    }
    CutSettleArgs c;
    c.next = t.next; c.p = a.p;
    c.cutblk = b.cutblk; c.budget = b.budget; c.admpre = b.admpre; c.used_kept = b.used_kept;
    c.cutidx = b.cutidx; c.used_cur = b.used_cur;
    c.Tg = Tg;
    c.wsp_sum = b.wsp_sum[0]; c.wsp_cnt = b.wsp_cnt[0]; c.bsp_sum = b.bsp_sum[0]; c.bsp_cnt = b.bsp_cnt[0];
    c.stats = b.stats; c.fx = b.fx;
    c.pk_next = pk.next;
    c.ul = a.ul;
    c.pack = pack ? 1u : 0u;
    hipLaunchKernelGGL(k_cut_settle, dim3(p.G), dim3(kBlock), cs_lds(p.m).total, s, c);
}

void launch_used_fold(u64* used, const u64* D, u32 m, u32 rounds, hipStream_t s) {
    if (!m) return;
    hipLaunchKernelGGL(k_used_fold, dim3((m + 255) / 256), dim3(256), 0, s, used, D, m, rounds);
}

void launch_lookup(const u32* assign, u64 n_obj, const u32* idx, u64 n, u32* out, DevStats* st, hipStream_t s, u32* done,
                   u32 seq, unsigned int* ticket) {
    if (!n) return;
    const bool vec = (((uintptr_t)idx | (uintptr_t)out) & 15u) == 0;
    const unsigned g = vec ? grid_for((n + 3) / 4, 256, 2048) : grid_for(n, 256, 4096);
    if (!vec) ticket = nullptr;  // (the several-workgroup completion protocol is k_lookup4's)
    if (g != 1 && !ticket) done = nullptr;  // without a ticket the completion word is a single-workgroup protocol
    if (!done) ticket = nullptr;
    if (vec) hipLaunchKernelGGL(k_lookup4, dim3(g), dim3(256), 0, s, assign, n_obj, idx, n, out, st, done, seq, ticket);
    else hipLaunchKernelGGL(k_lookup, dim3(g), dim3(256), 0, s, assign, n_obj, idx, n, out, st, done, seq);
}
void launch_update(u32* assign, u64 n_obj, u32 m, const u32* idx, const u32* node, u64 n, u32* pos, DevStats* st,
                   hipStream_t s, u32* aff_life, unsigned int* ticket, u32* done, u32 seq, u64* used, const u32* load) {
    if (!n) return;
    const unsigned g = grid_for(n, 256, 4096);
    hipLaunchKernelGGL(k_update_elect, dim3(g), dim3(256), 0, s, n_obj, m, idx, node, n, pos, st);
    hipLaunchKernelGGL(k_update_apply, dim3(g), dim3(256), 0, s, assign, n_obj, m, idx, node, n, pos, aff_life, ticket,
                       ticket ? done : nullptr, seq, used, load);
}
// inl (every micro-batch launcher): the n <= 4 requests themselves (a = indices, b = nodes / requesters), or nullptr
static inline uint4 inl_a(const SmallInline* inl) { return inl ? make_uint4(inl->a[0], inl->a[1], inl->a[2], inl->a[3]) : make_uint4(0, 0, 0, 0); }
static inline uint4 inl_b(const SmallInline* inl) { return inl ? make_uint4(inl->b[0], inl->b[1], inl->b[2], inl->b[3]) : make_uint4(0, 0, 0, 0); }
void launch_lookup_small(const u32* assign, u64 n_obj, const u32* idx, u32 n, u32* out, DevStats* st, hipStream_t s,
                         u32* done, u32 seq, const SmallInline* inl) {
    if (!n) return;
    hipLaunchKernelGGL(k_lookup_small, dim3(1), dim3(kSmallBatch), 0, s, assign, n_obj, idx, n, out, st, done, seq,
                       inl ? n : 0u, inl_a(inl));
}
void launch_update_small(u32* assign, const u32* idx, const u32* node, u32 n, hipStream_t s, u32* aff_life, u32* done,
                         u32 seq, const SmallInline* inl, u64* used, const u32* load, u32 m) {
    if (!n) return;
    hipLaunchKernelGGL(k_update_small, dim3(1), dim3(kSmallBatch), 0, s, assign, idx, node, n, aff_life, done, seq,
                       inl ? n : 0u, inl_a(inl), inl_b(inl), used, load, m);
}
void launch_remove_small(u32* assign, u32 m, const u32* load, const u32* idx, u32 n, u64* used, hipStream_t s, u32* aff_life,
                         u32* done, u32 seq, const SmallInline* inl) {
    if (!n) return;
    hipLaunchKernelGGL(k_remove_small, dim3(1), dim3(kSmallBatch), 0, s, assign, m, load, idx, n, used, aff_life, done, seq,
                       (inl && n <= 4) ? n : 0u, inl_a(inl));
}
void launch_remove(u32* assign, u64 n_obj, u32 m, const u32* load, const u32* idx, u64 n, u64* used, DevStats* st,
                   hipStream_t s, u32* aff_life, u32* done, u32 seq, const SmallInline* inl, unsigned int* ticket) {
    if (!n) return;
    const unsigned g = grid_for(n, kBlock * 4, 256);
    if (g != 1 && !ticket) done = nullptr;  // without a ticket the completion word is a single-workgroup protocol
    if (!done) ticket = nullptr;
    hipLaunchKernelGGL(k_remove, dim3(g), dim3(kBlock), used ? (size_t)m * sizeof(u64) : 0, s, assign,
                       n_obj, m, load, idx, n, used, st, aff_life, done, seq, (inl && n <= 4) ? (u32)n : 0u, inl_a(inl), ticket);
}
// The partitioned forms (see k_part_bin).  scratch: rec[n] | kk[n] (updates) | frag_off[nbins * 256] | frag_cnt[nbins * 256] u32 words,
// provided by the caller (part_scratch_words).  false: this batch / table does not qualify — use the plain kernels.
int g_part_shift = 14;  // rows per window <= 1 << shift (rio_gp_debug_set_part_shift: 12..14)
bool g_part_balance = true;   // window and chunk sizes that fill whole rounds of workgroups (bit 5 of the knob: powers of two, round 5's)
bool g_part_big = true; // 16 384-entry chunks where part_form() says so (bit 6 of rio_gp_debug_set_part_shift: small chunks only; A/B runs)
bool g_part_big_all = false;  // ... (bit 7) wherever they fit: CRUD batches and small request batches too (A/B runs, parity tests)
int g_pp_staged_from = kPpStagedFrom;  // host-buffer batches (lab builds: bits 8.. of rio_gp_debug_set_part_shift, A/B runs)
void set_part_shift(int v) {
    const int shift = v & 0x1F;
    g_part_shift = shift < 12 ? 12 : shift > (int)kPartShiftMax ? (int)kPartShiftMax : shift;
    g_part_balance = !(v & 0x20);
    g_part_big = !(v & 0x40);
    g_part_big_all = (v & 0x80) != 0;
    if (v >> 8) g_pp_staged_from = v >> 8;
}
// rows per window: the largest (1 << g_part_shift), or — balanced — the size at which the windows fill R whole rounds of kPartCUs
// workgroups, R = the rounds the largest size needs (10 M rows: 611 windows of 16 384 -> 766 of 13 056)
static u32 part_window(u64 n_obj) {
    const u64 wmax = (u64)1 << g_part_shift;
    if (!g_part_balance) return (u32)wmax;
    const u64 full = (n_obj + wmax - 1) / wmax;
    const u64 R = (full + kPartCUs - 1) / kPartCUs;
    u64 W = (n_obj + (u64)kPartCUs * R - 1) / ((u64)kPartCUs * R);
    W = (W + 255) & ~(u64)255;
    if (W < 4096) W = 4096;
    if (W > wmax || (n_obj + W - 1) / W > kPartMaxBins) W = wmax;
    return (u32)W;
}
static inline u64 part_bins(u64 n_obj) { const u64 W = part_window(n_obj); return (n_obj + W - 1) / W; }
static size_t part_bin_lds(u32 nbins, size_t rec_bytes, u32 sub) {
    return kSmall + (((size_t)2 * nbins + 1) * sizeof(u32) + 15) / 16 * 16 + (size_t)sub * rec_bytes;
}
// The chunk FORM (the largest chunk: 8 192 entries, 8 per lane of k_part_bin and a quarter wave per piece in the apply kernels |
// 16 384, 16 per lane and a half wave).  Big chunks make the apply kernels' pieces twice as long (less of every 128-byte line read
// for nothing, half as many descriptors) and k_part_bin's workgroups twice as long; measured on the 10 M x 1 024 table with 10 M
// entries (round 6, profiles/round6_crud_ab.json, round6_pp_chunks.json): they pay from 4 M entries on (below that k_part_bin is a
// single round of workgroups either way, and twice as long with big chunks).
static inline u32 part_form(u64 n_obj, u64 n) {
    const bool fits = part_bin_lds((u32)part_bins(n_obj), sizeof(uint2), kPartSubBig) <= (size_t)150 * 1024;
    if (g_part_big_all && fits) return kPartSubBig;
    return (g_part_big && fits && n >= ((u64)1 << 22)) ? kPartSubBig : kPartSub;
}
// entries per chunk for a slice of ns entries: the form's size, or — balanced, from two rounds of chunks on — the size at which the
// chunks fill whole rounds (10 M entries: 611 chunks of 16 384 -> 768 of 13 024)
static u32 part_chunk(u64 ns, u32 form) {
    if (!g_part_balance) return form;
    const u64 R = (ns + (u64)kPartCUs * form - 1) / ((u64)kPartCUs * form);
    if (R < 2) return form;
    u64 sub = (ns + (u64)kPartCUs * R - 1) / ((u64)kPartCUs * R);
    sub = (sub + 3) & ~(u64)3;
    return sub > form ? form : (u32)sub;
}
struct PartPlan { PartGeo g; u32 nbins; u32 form; u64 slice_max; };
static PartPlan part_plan(u64 n_obj, u64 n, bool pp) {
    PartPlan q;
    q.g.W = part_window(n_obj);
    q.g.wmagic = (u32)((((u64)1 << 42) + q.g.W - 1) / q.g.W);
    q.nbins = (u32)((n_obj + q.g.W - 1) / q.g.W);
    q.form = part_form(n_obj, n);
    const u64 ns = n < (u64)kPartSliceMax ? n : (u64)kPartSliceMax;
    q.g.sub = part_chunk(ns, q.form);
    q.slice_max = (u64)(kPartSliceMax / q.form) * q.g.sub;  // 2 048 small / 1 024 big chunks a launch: the apply kernels' 32 pieces per lane group
    return q;
}
bool part_applicable(u64 n_obj, u64 n, const void* idx, const void* node) {
    const u64 nbins = part_bins(n_obj);
    // dense enough that rewriting whole windows pays (a sparse batch touches few rows of each), columns 16-byte aligned
    return n >= ((u64)1 << 18) && n <= 0x7FFFFFFFull && n * 8 >= n_obj && nbins >= 32 && nbins <= kPartMaxBins &&
           n_obj <= ((u64)1 << 27) && (((uintptr_t)idx | (uintptr_t)node) & 15u) == 0;
}
// one slice of the batch at a time: records (8 B each, whole chunks) + the u16 chunk table [nbins + 1][chunks]; sized for either form
size_t part_scratch_words(u64 n_obj, u64 n) {
    size_t need = 0;
    for (int pp = 0; pp < 2; ++pp) {
        const PartPlan q = part_plan(n_obj, n, pp != 0);
        const u64 ns = n < q.slice_max ? n : q.slice_max;
        const u64 chunks = (ns + q.g.sub - 1) / q.g.sub;
        const size_t w = (size_t)(2 * chunks * q.g.sub + (((u64)q.nbins + 1) * chunks + 1) / 2 + 64);
        need = w > need ? w : need;
    }
    return need;
}
void launch_update_part(u32* assign, u64 n_obj, u32 m, const u32* idx, const u32* node, u64 n, u32* scratch, DevStats* st,
                        hipStream_t s, u32* aff_life) {
    const PartPlan q = part_plan(n_obj, n, false);
    const u32 nbins = q.nbins, sub = q.g.sub;
    for (u64 at = 0; at < n; at += q.slice_max) {  // slices in batch order: a later slice overwrites an earlier one's rows
        const u64 ns = n - at < q.slice_max ? n - at : q.slice_max;
        const u32 chunks = (u32)((ns + sub - 1) / sub);
        uint2* rec2 = reinterpret_cast<uint2*>(scratch);
        unsigned short* start16 = reinterpret_cast<unsigned short*>(scratch + 2 * (size_t)chunks * sub);
        const size_t blds = part_bin_lds(nbins, sizeof(uint2), sub);
        const size_t lds = (size_t)q.g.W * sizeof(u64);
        if (q.form == kPartSubBig) {
            hipLaunchKernelGGL((k_part_bin<true, 16>), dim3(chunks), dim3(kBlock), blds, s, n_obj, m, idx + at,
                               node + at, ns, nbins, q.g, (u32*)nullptr, rec2, start16, st, true);
            hipLaunchKernelGGL(k_part_update<32>, dim3(nbins), dim3(kBlock), lds, s, assign, n_obj, rec2, start16, chunks, aff_life, q.g);
        } else {
            hipLaunchKernelGGL((k_part_bin<true, 8>), dim3(chunks), dim3(kBlock), blds, s, n_obj, m, idx + at,
                               node + at, ns, nbins, q.g, (u32*)nullptr, rec2, start16, st, true);
            hipLaunchKernelGGL(k_part_update<16>, dim3(nbins), dim3(kBlock), lds, s, assign, n_obj, rec2, start16, chunks, aff_life, q.g);
        }
    }
}
void launch_remove_part(u32* assign, u64 n_obj, u32 m, const u32* load, const u32* idx, u64 n, u32* scratch, u64* used,
                        DevStats* st, hipStream_t s, u32* aff_life) {
    const PartPlan q = part_plan(n_obj, n, false);
    const u32 nbins = q.nbins, sub = q.g.sub;
    for (u64 at = 0; at < n; at += q.slice_max) {
        const u64 ns = n - at < q.slice_max ? n - at : q.slice_max;
        const u32 chunks = (u32)((ns + sub - 1) / sub);
        u32* rec = scratch;
        unsigned short* start16 = reinterpret_cast<unsigned short*>(scratch + 2 * (size_t)chunks * sub);
        const size_t blds = part_bin_lds(nbins, sizeof(u32), sub);
        const size_t lds = (size_t)q.g.W * sizeof(u32) + (used ? (size_t)m * sizeof(u64) : 0) + 16;
        if (q.form == kPartSubBig) {
            hipLaunchKernelGGL((k_part_bin<false, 16>), dim3(chunks), dim3(kBlock), blds, s, n_obj, m, idx + at,
                               (const u32*)nullptr, ns, nbins, q.g, rec, (uint2*)nullptr, start16, st, true);
            hipLaunchKernelGGL(k_part_remove<32>, dim3(nbins), dim3(kBlock), lds, s, assign, n_obj, m, load, rec, start16, chunks, used,
                               aff_life, q.g);
        } else {
            hipLaunchKernelGGL((k_part_bin<false, 8>), dim3(chunks), dim3(kBlock), blds, s, n_obj, m, idx + at,
                               (const u32*)nullptr, ns, nbins, q.g, rec, (uint2*)nullptr, start16, st, true);
            hipLaunchKernelGGL(k_part_remove<16>, dim3(nbins), dim3(kBlock), lds, s, assign, n_obj, m, load, rec, start16, chunks, used,
                               aff_life, q.g);
        }
    }
}
// place_pending over a window-sorted batch (k_pp_win_*): scratch = part_scratch_words(n_obj, n) words (records + chunk table)
void launch_pp_bin(u64 n_obj, u32 m, const u32* idx, const u32* req, u64 n, u32* scratch, DevStats* st, u32* host_err, hipStream_t s,
                   u32* dead_bits, u64* claim_fast) {
    const PartPlan q = part_plan(n_obj, n, true);
    const u32 nbins = q.nbins, sub = q.g.sub;
    const u32 chunks = (u32)((n + sub - 1) / sub);
    uint2* rec2 = reinterpret_cast<uint2*>(scratch);
    unsigned short* start16 = reinterpret_cast<unsigned short*>(scratch + 2 * (size_t)chunks * sub);
    const size_t blds = part_bin_lds(nbins, sizeof(uint2), sub);
    if (q.form == kPartSubBig)
        hipLaunchKernelGGL((k_part_bin<true, 16>), dim3(chunks), dim3(kBlock), blds, s, n_obj, m, idx, req, n,
                           nbins, q.g, (u32*)nullptr, rec2, start16, st, false, host_err, (u32*)nullptr, dead_bits, (m + 31) / 32,
                           claim_fast, m + 1);
    else
        hipLaunchKernelGGL((k_part_bin<true, 8>), dim3(chunks), dim3(kBlock), blds, s, n_obj, m, idx, req, n,
                           nbins, q.g, (u32*)nullptr, rec2, start16, st, false, host_err, (u32*)nullptr, dead_bits, (m + 31) / 32,
                           claim_fast, m + 1);
}
// claim_fast: [m] claim loads + [1] the "could not answer by itself" counter, zeroed by launch_pp_bin
void launch_pp_win_gather(u32* assign, const u32* load, u64 n_obj, u32 m, const u32* alive_bits, u64 n, const u32* scratch,
                          u32* ans0, u32* ans1, u32* dead_bits, u32* aff_life, const DevStats* st, u64* claim_fast, hipStream_t s) {
    const PartPlan q = part_plan(n_obj, n, true);
    const u32 nbins = q.nbins, sub = q.g.sub;
    const u32 chunks = (u32)((n + sub - 1) / sub);
    const uint2* rec2 = reinterpret_cast<const uint2*>(scratch);
    const unsigned short* start16 = reinterpret_cast<const unsigned short*>(scratch + 2 * (size_t)chunks * sub);
    // (dead_bits and claim_fast were cleared by the binning kernel: launch_pp_bin)
    const size_t win = (size_t)q.g.W * sizeof(u64);
    const u32 lds_hist = win + (size_t)m * sizeof(u64) <= (size_t)150 * 1024 ? 1u : 0u;  // else: global atomics per first touch
    const size_t lds = win + (lds_hist ? (size_t)m * sizeof(u64) : 0);
    if (q.form == kPartSubBig)
        hipLaunchKernelGGL((k_pp_win_gather<32, 24>), dim3(nbins), dim3(kBlock), lds, s, assign, load,
                           n_obj, m, alive_bits, rec2, start16, chunks, q.g, ans0, ans1, dead_bits, aff_life, st, claim_fast,
                           claim_fast + m, lds_hist, trace_flag());
    else if (chunks <= 4 * 64)
        hipLaunchKernelGGL((k_pp_win_gather<16, 4>), dim3(nbins), dim3(kBlock), lds, s, assign, load,
                           n_obj, m, alive_bits, rec2, start16, chunks, q.g, ans0, ans1, dead_bits, aff_life, st, claim_fast,
                           claim_fast + m, lds_hist, trace_flag());
    else
        hipLaunchKernelGGL((k_pp_win_gather<16, 24>), dim3(nbins), dim3(kBlock), lds, s, assign, load,
                           n_obj, m, alive_bits, rec2, start16, chunks, q.g, ans0, ans1, dead_bits, aff_life, st, claim_fast,
                           claim_fast + m, lds_hist, trace_flag());
}
void launch_pp_win_verdict(u32 m, const u64* cap, const u32* alive_bits, u64* used, const u64* claim_fast, const DevStats* st,
                           u32* verdict_dev, u32* verdict_host, hipStream_t s) {
    hipLaunchKernelGGL(k_pp_win_verdict, dim3(1), dim3(kBlock), 0, s, m, cap, alive_bits, used, claim_fast, claim_fast + m, st,
                       verdict_dev, verdict_host);
}
void launch_pp_win_unsort(u64 n_obj, const u32* scratch, const u32* ans0, const u32* ans1, u64 n, uint2* vrec, u32* out_node,
                          u32* out_flag, const u32* verdict, hipStream_t s) {
    const u32 sub = part_plan(n_obj, n, true).g.sub;
    const u32 chunks = (u32)((n + sub - 1) / sub);
    hipLaunchKernelGGL(k_pp_win_unsort, dim3(chunks), dim3(kBlock), (size_t)2 * sub * sizeof(u32), s,
                       reinterpret_cast<const uint2*>(scratch), ans0, ans1, n, vrec, out_node, out_flag, verdict, sub);
}
void launch_pp_win_output(const u32* idx, const u32* req, u64 n, const u32* vcur, const u32* vload, const u32* vnext,
                          const u32* alive_bits, const u32* cutidx, u32 m, u32* out_node, u32* out_flag, u32* aff_life,
                          const DevStats* st, hipStream_t s, u32 sa, const uint2* vrec) {
    hipLaunchKernelGGL(k_pp_win_output, dim3(grid_for((n + 3) / 4, 256, 4096)), dim3(256), 0, s, idx, req, n, vcur, vload, vnext,
                       alive_bits, cutidx, m, out_node, out_flag, aff_life, st, sa, vrec);
}
bool pp_win_applicable(u64 n_obj, u64 n, const void* idx, const void* req) {
    // (the windows' rows are only READ here, once and densely: it pays for sparser batches than the CRUD forms' n_obj / 8)
    const u64 nbins = part_bins(n_obj);
    return n >= ((u64)1 << 18) && n <= (u64)kPartSliceMax && n * 32 >= n_obj && nbins >= 32 && nbins <= kPartMaxBins &&
           n_obj <= ((u64)1 << 27) && (((uintptr_t)idx | (uintptr_t)req) & 15u) == 0;
}
void launch_clean(u32* assign, u64 n_obj, u32 m, const u32* dead_bits, u64* used, DevStats* st, hipStream_t s,
                  u64* counter, unsigned int* ticket, u64* host_out, u32* aff_life, u32 seq, const u32* skip_if) {
    const size_t lds = (size_t)((m + 31) / 32 + 4) * sizeof(u32);
    // a wave in which more than `full_from` lanes evict writes its whole kilobyte back (k_clean's comment); 64 = only the changed
    // 16-byte vectors ever.  16: measured in round 6 (profiles/round6_clean_ab.json, lab builds read RIO_GP_CLEAN_FULL_FROM)
    u32 full_from = 16;
#ifdef RIO_GP_LAB
    static const char* e = getenv("RIO_GP_CLEAN_FULL_FROM");
    if (e) full_from = (u32)atoi(e);
#endif
    hipLaunchKernelGGL(k_clean, dim3(grid_for((n_obj + 3) / 4, kBlock, 256)), dim3(kBlock), lds, s, assign, n_obj, m,
                       dead_bits, used, counter ? counter : &st->evicted_clean, ticket, host_out, aff_life, seq, skip_if, full_from);
}
void launch_recompute_used(const u32* assign, const u32* load, u64 n_obj, u32 m, u64* used, hipStream_t s) {
    (void)hipMemsetAsync(used, 0, (size_t)m * sizeof(u64), s);
    if (!n_obj) return;
    hipLaunchKernelGGL(k_used, dim3(grid_for((n_obj + 3) / 4, kBlock, 256)), dim3(kBlock), (size_t)m * sizeof(u64), s,
                       assign, load, n_obj, m, used);
}
void launch_set_attrs(u32* load, u32* aff, u64 n_obj, const u32* idx, const u32* nload, const u32* naff, u64 n,
                      DevStats* st, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_set_attrs, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s, load, aff, n_obj, idx, nload, naff, n,
                       st);
}
void launch_count_placed(const u32* assign, u64 n_obj, DevStats* st, hipStream_t s) {
    hipLaunchKernelGGL(k_count_placed, dim3(grid_for((n_obj + 3) / 4, 256, 2048)), dim3(256), 0, s, assign, n_obj, st);
}
void launch_fill_u32(u32* p, u64 n, u32 v, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_fill_u32, dim3(grid_for(n, 256, 8192)), dim3(256), 0, s, p, n, v);
}
// End of a synchronous call of several big kernels: one thread, behind them in the stream, copies the error counter into a
// mapped host word and stores the completion word — the host spins on it instead of a counter copy-back + hipStreamSynchronize
// (~25 us of a 130 us update_batch of 10 M entries).  No fence inside the big kernels (LESSONS 25): the launch boundary orders them.
__global__ void k_finish_err(const DevStats* __restrict__ st, u32* __restrict__ host_err, u32* __restrict__ done, u32 seq) {
    __hip_atomic_store(host_err, st->err ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_finish_err(const DevStats* st, u32* host_err, u32* done, u32 seq, hipStream_t s) {
    hipLaunchKernelGGL(k_finish_err, dim3(1), dim3(1), 0, s, st, host_err, done, seq);
}
void launch_store_words(const WordPack& pack, u32 nwords, u32* dst, hipStream_t s) {
    if (!nwords) return;
    hipLaunchKernelGGL(k_store_words, dim3((nwords + 63) / 64), dim3(64), 0, s, pack, nwords, dst);
}
void launch_pack_alive(const uint8_t* alive_bytes, u32 m, u32* alive_bits, hipStream_t s) {
    const u32 w = (m + 31) / 32;
    if (!w) return;
    hipLaunchKernelGGL(k_pack_alive, dim3((w + 63) / 64), dim3(64), 0, s, alive_bytes, m, alive_bits);
}
static CrudSmall crud_small_of(const CrudSmallArgs& a) {
    CrudSmall c = CrudSmall();
    c.nu = a.nu; c.nr = a.nr; c.nl = a.nl;
    c.u_idx = a.u_idx; c.u_node = a.u_node; c.r_idx = a.r_idx; c.l_idx = a.l_idx; c.l_out = a.l_out;
    c.n_obj = a.n_obj; c.st = a.st;
    if (a.u_inl) { c.u_ninl = a.nu; c.u_ia = inl_a(a.u_inl); c.u_ib = inl_b(a.u_inl); }
    if (a.r_inl) { c.r_ninl = a.nr; c.r_ia = inl_a(a.r_inl); }
    if (a.l_inl) { c.l_ninl = a.nl; c.l_ia = inl_a(a.l_inl); }
    return c;
}
void launch_pp_one(u32* assign, const u32* load, u32 m, const u64* cap, const u32* alive_bits, u64* used,
                   const u32* idx, const u32* req, u32 n, u32* out_node, u32* out_flag, u32* status, hipStream_t s,
                   u32* aff_life, u32* done, u32 seq, const SmallInline* inl, u32 n_obj_chk, void* stage, unsigned int* ticket, u32 sa,
                   bool host_io, const CrudSmallArgs* crud) {
    // Three launches (stage | decide | apply) when the requests and results are mapped HOST memory and the batch is beyond the
    // small kernel's 256: the one-workgroup kernel reads them over PCIe from ONE compute unit.  Same run, us per call,
    // one workgroup -> three launches: 300 requests 26-28 -> 24.7-24.9, 1 000: 28.3-33 -> 25.2-26, 1 024: 28.5-33 -> 24.8-26.4
    // (round 4's A/B: profiles/archive/round4_pp_staged_ab.json).  Device-resident requests (_dev) stay with the single launch up to 1 024.
    if (stage && ticket && done && n > (u32)(host_io ? g_pp_staged_from : kOneBatch / 4) && m <= kPpTot) {
        uint4* rec = static_cast<uint4*>(stage);
        uint4* rec2 = rec + kOneBatch;
        uint2* res = reinterpret_cast<uint2*>(rec2 + kOneBatch);
        u32* st = reinterpret_cast<u32*>(res + kOneBatch);  // [0] bad-entry flag (0 between calls) | [1] status
        const unsigned g = (n + 255u) / 256u;
        hipLaunchKernelGGL(k_pp_stage, dim3(g), dim3(256), 0, s, assign, load, m, cap, alive_bits, used, idx, req, n, rec, rec2, st,
                           n_obj_chk, sa);
        const size_t lds = (size_t)kPpTot * sizeof(u64) + (size_t)2 * (2 * kOneBatch) * sizeof(u32);
        hipLaunchKernelGGL(k_pp_decide, dim3(1), dim3(kBlock), lds, s, rec, rec2, n, m, alive_bits, used, res, st, status);
        hipLaunchKernelGGL(k_pp_apply, dim3(g), dim3(256), 0, s, assign, rec, res, n, out_node, out_flag, st, aff_life, ticket, done, seq);
        return;
    }
    const CrudSmall none = CrudSmall();
    if (n <= (u32)kSmallBatch) {
        const size_t lds = (size_t)kPpTot * sizeof(u64) + (size_t)2 * (2 * kSmallBatch) * sizeof(u32);
        hipLaunchKernelGGL((k_pp_one<kSmallBatch, 1>), dim3(1), dim3(kSmallBatch), lds, s, assign, load, m, cap, alive_bits, used,
                           idx, req, n, out_node, out_flag, status, aff_life, done, seq, inl ? n : 0u, inl_a(inl), inl_b(inl), n_obj_chk, trace_flag(), sa,
                           crud ? crud_small_of(*crud) : none);
    } else {
        const size_t lds = (size_t)kPpTot * sizeof(u64) + (size_t)2 * (2 * kOneBatch) * sizeof(u32);
        hipLaunchKernelGGL((k_pp_one<kBlock, kOneBatch / kBlock>), dim3(1), dim3(kBlock), lds, s, assign, load, m, cap, alive_bits,
                           used, idx, req, n, out_node, out_flag, status, aff_life, done, seq, 0u, inl_a(nullptr), inl_b(nullptr), n_obj_chk, trace_flag(), sa, none);
    }
}
void launch_crud_small(const CrudSmallArgs& c, u32* assign, const u32* load, u32 m, u64* used, u32* aff_life, u32* done, u32 seq,
                       hipStream_t s) {
    hipLaunchKernelGGL(k_crud_small, dim3(1), dim3(kSmallBatch), 0, s, crud_small_of(c), assign, load, m, used, aff_life, done, seq);
}
// every non-null pointer is 16-byte aligned (the kernels' dwordx4 paths; a caller's device arrays may start anywhere)
template <typename... P>
static inline bool aligned16(P... p) { return ((... | reinterpret_cast<uintptr_t>(p)) & 15u) == 0; }
void launch_ppm_first(const u32* assign, u64 n_obj, u32 m, const u32* alive_bits, const u32* idx, const u32* req, u64 n,
                      u32* pos, u32* s_idx, u32* s_req, u32* dead_bits, u32* vflag, u32* bad, hipStream_t s) {
    if (dead_bits) (void)hipMemsetAsync(dead_bits, 0, (size_t)((m + 31) / 32) * sizeof(u32), s);
    const u32 vec = aligned16(idx, req, s_idx, s_req, vflag) ? 1u : 0u;
    hipLaunchKernelGGL(k_ppm_first, dim3(grid_for(vec ? (n + 3) / 4 : n, 256, 4096)), dim3(256), 0, s, assign, n_obj, m, alive_bits, idx, req,
                       n, pos, s_idx, s_req, dead_bits, vflag, bad, vec);
}
void launch_ppm_gather(const u32* assign, const u32* load, const u32* idx, u64 n, const u32* pos, u32* vcur, u32* vload,
                       u32* vfirst, const u32* bad, hipStream_t s) {
    const u32 vec = aligned16(idx, vcur, vload, vfirst) ? 1u : 0u;
    hipLaunchKernelGGL(k_ppm_gather, dim3(grid_for(vec ? (n + 3) / 4 : n, 256, 4096)), dim3(256), 0, s, assign, load, idx, n, pos, vcur,
                       vload, vfirst, bad, vec);
}
void launch_ppm_output(u32* assign, u64 n_obj, const u32* idx, const u32* req, u64 n, const u32* vcur, const u32* vnext,
                       const u32* vfirst, const u32* vflag, u32* pos, const u32* alive_bits, const SolveBufs& b, const Plan& vp,
                       u32* out_node, u32* out_flag, u32* aff_life, u32* bad, bool fixup_done, u32* status, unsigned int* ticket,
                       u32* done, u32 seq, hipStream_t s) {
    PpmOutArgs a;
    a.assign = assign; a.idx = idx; a.req = req; a.n = n; a.n_obj = n_obj;
    a.vcur = vcur; a.vnext = vnext; a.vfirst = vfirst; a.vflag = vflag;
    a.pos = pos; a.alive_bits = alive_bits; a.cutidx = b.cutidx; a.m = vp.m; a.sa = vp.sa;
    a.out_node = out_node; a.out_flag = out_flag; a.aff_life = aff_life;
    a.bad = bad; a.stats = b.stats; a.bsp_cnt = b.bsp_cnt[0]; a.G = vp.G; a.fixup_done = fixup_done ? 1u : 0u;
    a.status = status; a.ticket = ticket; a.done = done; a.seq = seq;
    a.vec = aligned16(idx, req, vfirst, out_node, out_flag) ? 1u : 0u;
    hipLaunchKernelGGL(k_ppm_output, dim3(grid_for(a.vec ? (n + 3) / 4 : n, 256, 4096)), dim3(256), 0, s, a);
}

void launch_shard_pack1(const Plan& p, const SolveBufs& b, u64* X, hipStream_t s) {
    hipLaunchKernelGGL(k_shard_pack1, dim3(1), dim3(kBlock), 0, s, b.used_kept, b.claim_tot, b.partial,
                       resolve_blocks(p.m), p.m, X);
}
void launch_shard_import(const Plan& p, const NodeTab& nt, const SolveBufs& b, const u64* Xg, u32 rank, u32 R,
                         u64* gprev, u64* gfinal, u64* verdict_dev, u64* verdict_host, hipStream_t s) {
    const size_t lds = (size_t)(p.mwords + 4) * sizeof(u32);
    hipLaunchKernelGGL(k_shard_import, dim3(1), dim3(kBlock), lds, s, Xg, shard_words1(p.m), rank, R, p.m, nt.cap,
                       nt.alive_bits, b.used_kept, b.used_cur, b.claim_tot, b.cutblk, b.cutidx, gprev, gfinal, b.forced_bits,
                       b.rank_base, verdict_dev, verdict_host);
}
void launch_resolve_xchg(const Plan& p, const NodeTab& nt, const SolveBufs& b, u64* const* d_peers, u32 R, u32 rank,
                         size_t my_row_off, const u64* win_rows, size_t W, u64 seq, u64* p2p_err, u64* gprev, u64* gfinal,
                         u64* host_partial, u32 co_resident, hipStream_t s) {
    // co_resident = ranks whose exchange kernels run on THIS device (1: a GPU of its own).  Sharing ranks spin on each other
    // while a late one may still be scanning: together they keep to 512 workgroups of 256 threads — 8 of a CU's 32 wave
    // slots and a few KB of its LDS — so that scan always finds room (see k_resolve_xchg).
    const unsigned nb = resolve_blocks(p.m);
    unsigned grid = nb;
    if (co_resident > 1) {
        const unsigned cap = 512u / co_resident > 8u ? 512u / co_resident : 8u;
        if (grid > cap) grid = cap;
    }
    hipLaunchKernelGGL(k_resolve_xchg, dim3(grid), dim3(256), 0, s, b.H, b.blkstat, p, d_peers, R, rank,
                       my_row_off, win_rows, W, seq, p2p_err, nt.cap, nt.alive_bits, b.used_kept, b.used_cur, b.claim_tot,
                       b.cutblk, b.cutidx, gprev, gfinal, b.forced_bits, b.rank_base, b.partial, host_partial, nb, b.stats);
}
void launch_p2p_put(const u64* src, u32 words, u64* const* d_peers, u32 R, size_t data_off, size_t flag_off, u64 seq,
                    hipStream_t s) {
    hipLaunchKernelGGL(k_p2p_put, dim3(R), dim3(256), 0, s, src, words, d_peers, data_off, flag_off, seq);
}
void launch_p2p_wait_copy(const u64* win_slot, size_t W, u32 R, u32 words, const u64* flags, u64 seq, u64* err, u64* out,
                          hipStream_t s) {
    hipLaunchKernelGGL(k_p2p_wait_copy, dim3(1), dim3(kBlock), 0, s, win_slot, W, R, words, flags, seq, err, out);
}
void launch_shard_export_delta(const Plan& p, const SolveBufs& b, const u64* base, int wsp_sel, u64* Y, hipStream_t s) {
    hipLaunchKernelGGL(k_shard_export_delta, dim3(1), dim3(kBlock), 0, s, b.used_cur, base, b.wsp_sum[wsp_sel],
                       b.wsp_cnt[wsp_sel], p.nw, p.m, Y);
}
void launch_shard_export_put(const Plan& p, const SolveBufs& b, const u64* base, int wsp_sel, u64* const* d_peers, u32 R,
                             size_t data_off, size_t flag_off, u64 seq, hipStream_t s) {
    hipLaunchKernelGGL(k_shard_export_put, dim3(1), dim3(kBlock), 0, s, b.used_cur, base, b.wsp_sum[wsp_sel], b.wsp_cnt[wsp_sel],
                       p.nw, p.m, d_peers, R, data_off, flag_off, seq, b.stats);
}
void launch_shard_wait_import(const Plan& p, const SolveBufs& b, const u64* win_slot, size_t W, const u64* flags, u64 seq,
                              u64* err, u32 rank, u32 R, u64* gprev, const u64* gfinal, u64* verdict_dev, u64* verdict_host,
                              hipStream_t s) {
    hipLaunchKernelGGL(k_shard_wait_import, dim3(1), dim3(kBlock), 0, s, win_slot, W, flags, seq, err, rank, R, p.m, gprev, gfinal,
                       b.used_cur, b.rank_base, verdict_dev, verdict_host, b.stats);
}
void launch_shard_tick_stats(const Plan& p, const SolveBufs& b, u64* out_host, u64 mark, hipStream_t s) {
    hipLaunchKernelGGL(k_shard_tick_stats, dim3(1), dim3(64), 0, s, b.partial, resolve_blocks(p.m), b.stats, out_host, mark);
}
void launch_shard_import_delta(const Plan& p, const SolveBufs& b, const u64* Yg, u32 rank, u32 R, u64* gprev,
                               u64* verdict_dev, u64* verdict_host, hipStream_t s) {
    hipLaunchKernelGGL(k_shard_import_delta, dim3(1), dim3(kBlock), 0, s, Yg, rank, R, p.m, gprev, b.used_cur,
                       b.rank_base, verdict_dev, verdict_host);
}

}  // namespace riogp
