// common.hpp -- small HIP utilities shared by the device sources.
#ifndef OPMGPU_COMMON_HPP
#define OPMGPU_COMMON_HPP

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/opmgpu.h"

namespace opmgpu {

struct HipError : std::runtime_error {
    int code;
    HipError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define OPMGPU_HIP(expr)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            throw ::opmgpu::HipError(e_ == hipErrorOutOfMemory ? OPMGPU_ENOMEM : OPMGPU_ENODEVICE, \
                                     std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

// RAII device array
template <class T>
struct DevArray {
    T* p = nullptr;
    size_t n = 0;
    DevArray() {}
    DevArray(const DevArray&) = delete;
    DevArray& operator=(const DevArray&) = delete;
    ~DevArray() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    void alloc(size_t count) {
        if (count == n && p) return;
        release();
        if (count == 0) return;
        OPMGPU_HIP(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)));
        n = count;
    }
    void ensure(size_t count) { if (count > n) alloc(count); }
    void upload(const T* h, size_t count, hipStream_t s) {
        ensure(count);
        if (count) OPMGPU_HIP(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void upload(const std::vector<T>& h, hipStream_t s) { upload(h.data(), h.size(), s); }
    void download(T* h, size_t count, hipStream_t s) const {
        if (count) OPMGPU_HIP(hipMemcpyAsync(h, p, count * sizeof(T), hipMemcpyDeviceToHost, s));
    }
    void zero(hipStream_t s) { if (n) OPMGPU_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
};

// In-situ kernel timing (opmgpu_kernel_timing): HIP event pairs recorded on the launch stream around the launches of a kernel
// class DURING a real Newton iteration; off by default (two event records per bracket are not free), switched on by bench.py for a
// separate profiled pass that feeds the per-kernel roofline table.
enum { KT_CELL_PROPS = 0, KT_FLUX, KT_WELLS, KT_CONV, KT_ILU_FACTOR, KT_CPR_SETUP, KT_SPMV1, KT_SPMV2, KT_ILU_APPLY, KT_VCYCLE, KT_CPR_OTHER,
       KT_VECTOR, KT_UPDATE_STATE, KT_COUNT };
struct KernelTimers {
    bool on = false;
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> pool;                    // recycled events
    struct Rec { int id; hipEvent_t a, b; };
    std::vector<Rec> recs;
    double total_ms[KT_COUNT] = {};
    long count[KT_COUNT] = {};
    hipEvent_t get() { if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; } hipEvent_t e; OPMGPU_HIP(hipEventCreate(&e)); return e; }
    hipEvent_t begin() { if (!on) return nullptr; hipEvent_t e = get(); OPMGPU_HIP(hipEventRecord(e, stream)); return e; }
    void end(int id, hipEvent_t a) { if (!on || !a) return; hipEvent_t b = get(); OPMGPU_HIP(hipEventRecord(b, stream)); recs.push_back({ id, a, b }); if (recs.size() > 8192) collect(); }
    void collect() {
        if (recs.empty()) return;
        OPMGPU_HIP(hipEventSynchronize(recs.back().b));
        for (const Rec& r : recs) { float ms = 0.f; if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { total_ms[r.id] += ms; ++count[r.id]; } pool.push_back(r.a); pool.push_back(r.b); }
        recs.clear();
    }
    void reset() { collect(); for (int i = 0; i < KT_COUNT; ++i) { total_ms[i] = 0.0; count[i] = 0; } }
    ~KernelTimers() { for (const Rec& r : recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); } for (hipEvent_t e : pool) (void)hipEventDestroy(e); }
};
struct KtScope {        // RAII bracket
    KernelTimers& kt; int id; hipEvent_t a;
    KtScope(KernelTimers& k, int i) : kt(k), id(i), a(k.begin()) {}
    ~KtScope() { try { kt.end(id, a); } catch (...) {} }
};

// Work redirected to a side stream by swapping the stream member (and the in-situ timers off meanwhile): swapped back on every exit path,
// also when the redirected code throws -- otherwise every later call of the context would run on the side stream without its event ordering
struct StreamSwapGuard {
    hipStream_t& a; hipStream_t& b; bool& timing; bool saved;
    StreamSwapGuard(hipStream_t& main_, hipStream_t& side_, bool& timing_on) : a(main_), b(side_), timing(timing_on), saved(timing_on) { std::swap(a, b); timing = false; }
    ~StreamSwapGuard() { std::swap(a, b); timing = saved; }
    StreamSwapGuard(const StreamSwapGuard&) = delete;
    StreamSwapGuard& operator=(const StreamSwapGuard&) = delete;
};

constexpr int kBlock = 256;          // 4 wavefronts = 4 SELL slices per workgroup
constexpr int kMaxRedBlocks = 2048;  // grid cap of the reduction kernels (256 CUs x 8)

inline int grid_for(long n, int block = kBlock) { return int((n + block - 1) / block); }
// grids of the XCD-aware kernels are multiples of 8 (one slot per XCD and round)
inline int grid8_for(long n, int block = kBlock) { return (grid_for(n, block) + 7) / 8 * 8; }

// XCD-aware chunk mapping.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an XCD, each XCD
// has its own L2): give every XCD one CONTIGUOUS range of 256-row chunks so that the x / property lines its rows
// gather are fetched into one L2 instead of up to eight.  Speed only, never correctness (any mapping covers all chunks).
// Iterate:  for (int c = xcd_first(...); c < xcd_end(...); c += gridDim.x >> 3)
// mode is a kernel argument (0 = plain round-robin chunks, 1 = XCD-contiguous ranges) so both can be A/B-timed.
__device__ __forceinline__ int xcd_per(int nchunks) { return (nchunks + 7) >> 3; }
__device__ __forceinline__ int xcd_first(int nchunks, int mode) { return mode ? int(blockIdx.x & 7) * xcd_per(nchunks) + int(blockIdx.x >> 3) : int(blockIdx.x); }
__device__ __forceinline__ int xcd_end(int nchunks, int mode) { if (!mode) return nchunks; const int e = (int(blockIdx.x & 7) + 1) * xcd_per(nchunks); return e < nchunks ? e : nchunks; }
__device__ __forceinline__ int xcd_stride(int mode) { return mode ? int(gridDim.x >> 3) : int(gridDim.x); }
int xcd_mode();     // host: OPMGPU_XCD env (default chosen by measurement)

// ---- wave64 / workgroup reductions (double accumulators; deterministic order) ----
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}
// sum over a 256-thread block; result valid in thread 0.  `sm` = 4 doubles of LDS per value.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* sm)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double s = wave_sum(v[k]);
        if (lane == 0) sm[k * 4 + w] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = (sm[k * 4] + sm[k * 4 + 1]) + (sm[k * 4 + 2] + sm[k * 4 + 3]);
    }
}

} // namespace opmgpu
#endif
