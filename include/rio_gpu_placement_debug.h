/*
 * rio_gpu_placement_debug.h — policy knobs and measurement aids of the LAB build, librio_gp_lab.so.
 *
 * NOT part of the drop-in boundary and NOT in the product library: librio_gp.so exports exactly what
 * rio_gpu_placement.h and rio_gpu_object_placement.h declare (tests/test_abi_symbols.py checks it).  The lab build is
 * the same sources compiled with -DRIO_GP_LAB plus stream_probe.hip; it exists so that the parity tests can force every
 * policy the product picks adaptively (packed fix-up, packing at the cut pass, speculative enqueue, partitioned CRUD)
 * through the same inputs, and so that tools/ can take reference bandwidths.  May change between builds.
 */
#ifndef RIO_GPU_PLACEMENT_DEBUG_H
#define RIO_GPU_PLACEMENT_DEBUG_H

#include "rio_gpu_placement.h"

#ifdef __cplusplus
extern "C" {
#endif

/* packed fix-up (fix-up passes over the pending rows only): 0 = adaptive (default: used when the previous solve left
 * <= 25 % of the rows pending), 1 = always, 2 = never.  + 16: big update / remove batches (>= 2^18 entries) go through the plain
 * per-entry kernels instead of the window-partitioned ones.  + (mode << 5), mode 0 | 1 | 2 as above: packing at the cut pass
 * of a whole-table solve (adaptive: when the previous solve sent <= 25 % of the rows to the water-fill).
 * + (inc << 7): the in-place scan of COMMITTED ticks over a mostly-placed table (k_inc_scan: the assignment column is updated
 * in place, then k_rebal deals the pending rows out evenly to the fix-up's workgroups) — 0 or 1 = whenever the packed fix-up
 * is used and the `used` vector is valid (default) | 2 = never (k_scan<COMPACT>).
 * + (ca << 9): the whole-table fix-up by k_cut_apply + k_cut_settle (the exact cuts and the re-marking in one pass over the wave
 * ranges that have work) — 0 = when the solve packs at the cut pass (default) | 1 = always | 2 = never: k_cut_find, then the
 * re-marking pass inside round 0 of k_fill (two passes: round 5's form).
 * + 2048: the k_resolve of a quiet asynchronous tick (rio_gp_tick_async on a table nothing has changed in: k_scan + k_resolve,
 * no fix-up) stays on the main stream; by default it runs on a stream of its own beside the next tick's k_scan.
 * + 4096: the scans of such ticks are not CHAINED (every scan on the main stream, one launch after the other); by default they
 * alternate between two streams and hand their rows over wave range by wave range (ScanChain, placement_kernels.h).
 * Environment, read when a handle of the lab build is created: RIO_GP_OVERLAP_MIN_ROWS (the smallest table whose quiet ticks
 * overlap / chain; 2^18 rows in the product, 2^22 for the form without the chain), RIO_GP_CHAIN_INLINE_BELOW (tables below this
 * many rows run a chained tick's k_resolve in line behind its scan: 5 * 2^20 in the product, 0 = never), RIO_GP_CHAIN_PER_WAVE
 * (0: the hand-over per workgroup instead of per wave range), RIO_GP_CHAIN_TPI (1 | 2 tiles per wave-iteration of the chained scan),
 * RIO_GP_CHAIN_DIAG (timing experiments without the waits: NOT correct, tools/quiet_overlap_ab.py). */
int rio_gp_debug_set_compact(rio_gp_t* h, int mode);
/* chained scans enqueued by this handle so far (0: its quiet ticks have never met the conditions) */
uint64_t rio_gp_debug_chained_scans(rio_gp_t* h);
/* speculative enqueue of the fix-up behind k_resolve, without waiting for the verdict: 0 (default) = when the previous
 * solve needed it | 1 = always | 2 = never. */
int rio_gp_debug_set_speculate(rio_gp_t* h, int speculate);
/* non-temporal column streams in k_scan: 0 = by table size (default) | 1 = always | 2 = never; process-wide. */
void rio_gp_debug_set_scan_nt(int mode);
/* window of the partitioned update / remove batches: 1 << shift rows, shift 12..14 (default 14); process-wide. */
void rio_gp_debug_set_part_shift(int shift);
/* first row of wave range `wave` (0 .. *n_waves; the last value is the end of the table rounded up to a tile) of a table
 * of n_objects rows, exactly as the kernels compute it (a multiply-shift instead of a 64-bit division); host-only, needs
 * no GPU: tests/test_abi_symbols.py checks it against the plain division. */
uint64_t rio_gp_debug_wave_row_lo(uint64_t n_objects, uint32_t n_nodes, uint32_t wave, uint32_t* n_waves);
/* phase traces of the fix-up kernels: switch them on / off (enable) and, when out2048 != NULL, read table 0 = k_resolve with
 * the in-kernel cut search | 1 = k_fill round 0 | 2 = k_fill later rounds: 256 workgroups x 8 words of wall_clock64 (100 MHz) */
int rio_gp_debug_ktrace(rio_gp_t* h, int enable, int table, uint64_t* out2048);
/* pure streaming kernels with k_scan's traffic mix (3 columns in, 1 out) over the handle's own columns; mode
 * 0 grid-stride | 1 block-tiled | 2 wave-contiguous | 3 read-only | 4 1:1 copy.  ms per launch. */
int rio_gp_debug_stream_probe(rio_gp_t* h, int mode, int reps, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* RIO_GPU_PLACEMENT_DEBUG_H */
