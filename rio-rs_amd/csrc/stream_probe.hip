// stream_probe.hip — measurement aid, not part of the product path: pure streaming kernels with the
// SAME traffic mix as k_scan (read three u32 columns, write one) in several access patterns, so
// bench.py / tools/sweep_scan.py can report how far k_scan is from what this chip's memory system
// delivers for that mix (MI355X_MICROARCH.md quotes 6.29 TB/s for a 1:1 float4 copy).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <chrono>

#include "placement_kernels.h"

namespace riogp {

// mode 0: classic grid-stride, 256-thread blocks
__global__ __launch_bounds__(256) void k_probe_gridstride(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                                          const uint4* __restrict__ c, uint4* __restrict__ o, u64 nv) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < nv; i += (u64)gridDim.x * 256) {
        const uint4 x = a[i], y = b[i], z = c[i];
        uint4 r;
        r.x = x.x ^ y.x ^ z.x; r.y = x.y ^ y.y ^ z.y; r.z = x.z ^ y.z ^ z.z; r.w = x.w ^ y.w ^ z.w;
        o[i] = r;
    }
}
// mode 1: 1024-thread blocks, block-contiguous chunk, waves interleaved tile by tile inside the chunk
__global__ __launch_bounds__(1024) void k_probe_blocktile(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                                          const uint4* __restrict__ c, uint4* __restrict__ o, u64 nv) {
    const u64 per = (nv + gridDim.x - 1) / gridDim.x;
    const u64 lo = (u64)blockIdx.x * per;
    u64 hi = lo + per;
    if (hi > nv) hi = nv;
    for (u64 i = lo + threadIdx.x; i < hi; i += 1024) {
        const uint4 x = a[i], y = b[i], z = c[i];
        uint4 r;
        r.x = x.x ^ y.x ^ z.x; r.y = x.y ^ y.y ^ z.y; r.z = x.z ^ y.z ^ z.z; r.w = x.w ^ y.w ^ z.w;
        o[i] = r;
    }
}
// mode 2: 1024-thread blocks, every WAVE owns a contiguous range (k_scan's pattern)
__global__ __launch_bounds__(1024) void k_probe_wavecontig(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                                           const uint4* __restrict__ c, uint4* __restrict__ o, u64 nv) {
    const u64 nw = (u64)gridDim.x * 16, gw = (u64)blockIdx.x * 16 + (threadIdx.x >> 6);
    const u64 tiles = (nv + 63) / 64;
    const u64 lo = (gw * tiles / nw) * 64;
    u64 hi = ((gw + 1) * tiles / nw) * 64;
    if (hi > nv) hi = nv;
    for (u64 i = lo + (threadIdx.x & 63); i < hi; i += 64) {
        const uint4 x = a[i], y = b[i], z = c[i];
        uint4 r;
        r.x = x.x ^ y.x ^ z.x; r.y = x.y ^ y.y ^ z.y; r.z = x.z ^ y.z ^ z.z; r.w = x.w ^ y.w ^ z.w;
        o[i] = r;
    }
}
// mode 3: grid-stride, read-only (three columns), one word per block written
__global__ __launch_bounds__(256) void k_probe_readonly(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                                        const uint4* __restrict__ c, uint4* __restrict__ o, u64 nv) {
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < nv; i += (u64)gridDim.x * 256) {
        const uint4 x = a[i], y = b[i], z = c[i];
        acc ^= x.x ^ y.x ^ z.x ^ x.y ^ y.y ^ z.y ^ x.z ^ y.z ^ z.z ^ x.w ^ y.w ^ z.w;
    }
    if (acc == 0x12345678u) o[blockIdx.x].x = acc;  // practically never; keeps the loads alive
}
// mode 4: 1:1 copy (one column in, one out) — the guide's 6.29 TB/s reference pattern
__global__ __launch_bounds__(256) void k_probe_copy(const uint4* __restrict__ a, uint4* __restrict__ o, u64 nv) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < nv; i += (u64)gridDim.x * 256) o[i] = a[i];
}

// modes 7/8/9: grid-stride with non-temporal loads / stores / both (does the streaming hint help past the MALL?)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_probe_nt(const u32x4* __restrict__ a, const u32x4* __restrict__ b,
                                                  const u32x4* __restrict__ c, u32x4* __restrict__ o, u64 nv) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < nv; i += (u64)gridDim.x * 256) {
        const u32x4 x = NTL ? __builtin_nontemporal_load(a + i) : a[i];
        const u32x4 y = NTL ? __builtin_nontemporal_load(b + i) : b[i];
        const u32x4 z = NTL ? __builtin_nontemporal_load(c + i) : c[i];
        const u32x4 r = x ^ y ^ z;
        if (NTS) __builtin_nontemporal_store(r, o + i);
        else o[i] = r;
    }
}

// ---- host round-trip probes (modes 20..23): what does ONE synchronous call cost besides its kernel?  A one-thread kernel
// stores a sequence number into mapped pinned memory; the host either waits for the stream (what every synchronous entry
// point did up to round 2) or spins on the word.  `from_host`: the kernel first reads a word of mapped pinned memory (a
// request staged by the host) instead of taking it from its arguments.  Returns microseconds per call, as ms / 1000.
__global__ void k_probe_flag(const u32* __restrict__ req, u32* flag, u32 seq, int from_host) {
    u32 v = seq;
    if (from_host) v = __builtin_nontemporal_load(req);
    __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
float sync_probe(int mode, int reps, hipStream_t s) {
    u32* hbuf = nullptr;
    u32* dbuf = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&hbuf), 256, hipHostMallocMapped) != hipSuccess) return -1.f;
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&dbuf), hbuf, 0) != hipSuccess) { (void)hipHostFree(hbuf); return -1.f; }
    volatile u32* flag = hbuf + 32;
    const bool spin = mode == 21 || mode == 22;
    const int from_host = mode == 22 || mode == 23;
    double best = 1e30;
    for (int pass = 0; pass < 3; ++pass) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) {
            const u32 seq = (u32)(pass * reps + r + 1);
            hbuf[0] = seq;
            hipLaunchKernelGGL(k_probe_flag, dim3(1), dim3(1), 0, s, dbuf, dbuf + 32, seq, from_host);
            if (spin) {
                u64 guard = 0;
                while (*flag != seq && ++guard < (1ull << 31)) __builtin_ia32_pause();
            } else if (hipStreamSynchronize(s) != hipSuccess) { (void)hipHostFree(hbuf); return -1.f; }
        }
        if (hipStreamSynchronize(s) != hipSuccess) { (void)hipHostFree(hbuf); return -1.f; }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        if (us < best) best = us;
    }
    (void)hipHostFree(hbuf);
    return (float)(best / 1000.0);
}

// returns per-launch ms (dispatch timestamps) averaged over reps; bytes moved = see bench.py
float stream_probe(int mode, const u32* a, const u32* b, const u32* c, u32* o, u64 n, int reps, hipStream_t s,
                   hipEvent_t e0, hipEvent_t e1) {
    const u64 nv = n / 4;
    float total = 0;
    for (int r = 0; r < reps + 2; ++r) {
        const uint4 *A = (const uint4*)a, *B = (const uint4*)b, *C = (const uint4*)c;
        uint4* O = (uint4*)o;
        switch (mode) {
            case 0: hipExtLaunchKernelGGL(k_probe_gridstride, dim3(2048), dim3(256), 0, s, e0, e1, 0, A, B, C, O, nv); break;
            case 1: hipExtLaunchKernelGGL(k_probe_blocktile, dim3(256), dim3(1024), 0, s, e0, e1, 0, A, B, C, O, nv); break;
            case 2: hipExtLaunchKernelGGL(k_probe_wavecontig, dim3(256), dim3(1024), 0, s, e0, e1, 0, A, B, C, O, nv); break;
            case 10: hipExtLaunchKernelGGL(k_probe_wavecontig, dim3(512), dim3(1024), 0, s, e0, e1, 0, A, B, C, O, nv); break;  // 2 workgroups per CU
            case 11: hipExtLaunchKernelGGL(k_probe_blocktile, dim3(512), dim3(1024), 0, s, e0, e1, 0, A, B, C, O, nv); break;
            case 3: hipExtLaunchKernelGGL(k_probe_readonly, dim3(2048), dim3(256), 0, s, e0, e1, 0, A, B, C, O, nv); break;
            case 5: hipExtLaunchKernelGGL(k_probe_gridstride, dim3(8192), dim3(256), 0, s, e0, e1, 0, A, B, C, O, nv); break;
            case 7: hipExtLaunchKernelGGL((k_probe_nt<true, false>), dim3(2048), dim3(256), 0, s, e0, e1, 0, (const u32x4*)a, (const u32x4*)b, (const u32x4*)c, (u32x4*)o, nv); break;
            case 8: hipExtLaunchKernelGGL((k_probe_nt<false, true>), dim3(2048), dim3(256), 0, s, e0, e1, 0, (const u32x4*)a, (const u32x4*)b, (const u32x4*)c, (u32x4*)o, nv); break;
            case 9: hipExtLaunchKernelGGL((k_probe_nt<true, true>), dim3(2048), dim3(256), 0, s, e0, e1, 0, (const u32x4*)a, (const u32x4*)b, (const u32x4*)c, (u32x4*)o, nv); break;
            case 6: hipExtLaunchKernelGGL(k_probe_gridstride, dim3(1024), dim3(256), 0, s, e0, e1, 0, A, B, C, O, nv); break;
            default: hipExtLaunchKernelGGL(k_probe_copy, dim3(2048), dim3(256), 0, s, e0, e1, 0, A, O, nv); break;
        }
        if (hipStreamSynchronize(s) != hipSuccess) return -1.f;
        float ms = 0;
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1.f;
        if (r >= 2) total += ms;
    }
    return total / reps;
}

}  // namespace riogp
