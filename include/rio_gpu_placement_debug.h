/*
 * rio_gpu_placement_debug.h — A/B knobs and measurement aids of librio_gp.so.
 *
 * NOT part of the drop-in boundary: nothing a Rust / cgo / ctypes binding of the ObjectPlacement path needs is
 * declared here (that is rio_gpu_placement.h and rio_gpu_object_placement.h).  These entry points exist so that the
 * parity tests can drive every implementation of the fix-up through the same inputs (results are identical in every
 * mode by construction, and the tests check it), and so that tools/ can take phase traces and reference bandwidths.
 * They may change or disappear between builds.
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
 * of a whole-table solve (adaptive: when the previous solve sent <= 25 % of the rows to the water-fill). */
int rio_gp_debug_set_compact(rio_gp_t* h, int mode);
/* cut / water-fill fix-up.  impl: 2 (default) = split launches (k_cut_find spread over the chip, then k_cut_apply) |
 * 1 = one fused launch per solve (k_cut_fused) | 0 = the first, unfused launch chain.  speculate: 0 (default) =
 * enqueue the fix-up behind k_resolve without waiting for the verdict when the previous solve needed it | 1 = always |
 * 2 = never. */
int rio_gp_debug_set_fixup(rio_gp_t* h, int impl, int speculate);
/* non-temporal column streams in k_scan: 0 = by table size (default) | 1 = always | 2 = never; process-wide. */
void rio_gp_debug_set_scan_nt(int mode);
/* window of the partitioned update / remove batches: 1 << shift rows, shift 12..14 (default 14); process-wide. */
void rio_gp_debug_set_part_shift(int shift);
/* first row of wave range `wave` (0 .. *n_waves; the last value is the end of the table rounded up to a tile) of a table
 * of n_objects rows, exactly as the kernels compute it (a multiply-shift instead of a 64-bit division); host-only, needs
 * no GPU: tests/test_abi_symbols.py checks it against the plain division. */
uint64_t rio_gp_debug_wave_row_lo(uint64_t n_objects, uint32_t n_nodes, uint32_t wave, uint32_t* n_waves);
/* read (out2048 != NULL: 256 workgroups x 8 words) and switch the phase trace of the cut kernels */
int rio_gp_debug_cut_trace(rio_gp_t* h, int enable, uint64_t* out2048);
/* phase traces of the other fix-up kernels (switched by rio_gp_debug_cut_trace's enable): table 0 / 1 = k_spill_apply
 * first / last round, 2 = k_cut_apply_rank, 3 = k_cut_find; 256 workgroups x 8 words of wall_clock64 (100 MHz) */
int rio_gp_debug_ktrace(rio_gp_t* h, int table, uint64_t* out2048);
/* pure streaming kernels with k_scan's traffic mix (3 columns in, 1 out) over the handle's own columns; mode
 * 0 grid-stride | 1 block-tiled | 2 wave-contiguous | 3 read-only | 4 1:1 copy.  ms per launch. */
int rio_gp_debug_stream_probe(rio_gp_t* h, int mode, int reps, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* RIO_GPU_PLACEMENT_DEBUG_H */
