// bge_route.hip — sharded broadphase on gfx950: route body records to spatial slabs, one slab per rank.
//
// The tick shards the scene by SUBTREE (bge_partition_subtrees), so the bodies of different ranks interleave in space
// and a per-rank broadphase misses every pair that straddles two ranks (SURVEY.md §8(e) "not sharded", §8(f) rank 4).
// The reference has one Bullet world and therefore one global pair set (src/physics/PhysicsSystem.cpp:124, 863); to
// produce that set on N GPUs the broadphase is re-partitioned SPATIALLY for the duration of the pair search:
//
//   * the ranks agree on cuts c_1 <= ... <= c_{N-1} along one axis (slab s = [c_s, c_{s+1}), c_0 = -inf, c_N = +inf);
//   * every body sends one 48-byte record (AABB, global entity id, filter words) to each slab its AABB's extent
//     [min, max] along the axis touches — one all-to-all over xGMI, ~48 B per body plus the ghosts at slab borders;
//   * each rank runs the ordinary single-GPU broadphase on what it received and keeps a pair only if the LOWER END of
//     the pair's overlap interval, max(min_a, min_b), lies in its own slab.  That point belongs to both AABBs'
//     extents, so both bodies were sent to that slab; and it lies in exactly one slab, so the union over ranks is the
//     global pair set with no duplicates.
// The work per rank is 1/N of the global search (plus ghosts); no rank ever holds all bodies.
#include "bge_route.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/bge_world.h"
#include "bge_flatten.hpp"

namespace bge {

namespace {

struct RouteScalars {
    unsigned long long count[kMaxSlabs];
    unsigned long long cursor[kMaxSlabs];
    float cuts[kMaxSlabs + 1];
    uint32_t mn[3], mx[3]; // ordered-uint encoding
    unsigned long long n_bodies;
};

__device__ __forceinline__ bool is_body(uint32_t f) { return (f & kTypeMask) != 0; }

__device__ __forceinline__ uint32_t f2ord(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float ord2f_host(uint32_t o)
{
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// slab of a coordinate: the number of interior cuts <= z (a NaN lands in slab 0)
__device__ __forceinline__ uint32_t slab_of(const float* __restrict__ cuts, uint32_t nranks, float z)
{
    uint32_t s = 0;
    for (uint32_t k = 1; k < nranks; ++k) s += (z >= cuts[k]) ? 1u : 0u;
    return s;
}

__global__ void __launch_bounds__(256) k_route_reset(RouteScalars* sc)
{
    if (threadIdx.x < kMaxSlabs) sc->count[threadIdx.x] = 0;
    if (threadIdx.x < 3) {
        sc->mn[threadIdx.x] = 0xffffffffu;
        sc->mx[threadIdx.x] = 0u;
    }
    if (threadIdx.x == 0) sc->n_bodies = 0;
}

// Pass 1: records per destination slab.  Counts are aggregated in LDS; one global atomic per (workgroup, slab).
__global__ void __launch_bounds__(256) k_route_count(uint64_t n_slots, const uint32_t* __restrict__ flags,
                                                     const float* __restrict__ aabb, uint32_t axis, uint32_t nranks,
                                                     RouteScalars* sc)
{
    __shared__ uint32_t cnt[kMaxSlabs];
    __shared__ float cuts[kMaxSlabs + 1];
    if (threadIdx.x < kMaxSlabs) cnt[threadIdx.x] = 0;
    if (threadIdx.x <= nranks) cuts[threadIdx.x] = sc->cuts[threadIdx.x];
    __syncthreads();
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t s = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; s < n_slots; s += stride) {
        if (!is_body(flags[s])) continue;
        const uint32_t d_lo = slab_of(cuts, nranks, aabb[6 * s + axis]);
        const uint32_t d_hi = max(d_lo, slab_of(cuts, nranks, aabb[6 * s + 3 + axis]));
        for (uint32_t d = d_lo; d <= d_hi; ++d) atomicAdd(&cnt[d], 1u);
    }
    __syncthreads();
    if (threadIdx.x < nranks && cnt[threadIdx.x]) atomicAdd(&sc->count[threadIdx.x], static_cast<unsigned long long>(cnt[threadIdx.x]));
}

// Pass 2: write the records.  Per workgroup: count per slab in LDS, reserve one contiguous range per slab behind a
// single global atomic, then hand out positions inside the range with LDS atomics.
__global__ void __launch_bounds__(256) k_route_pack(uint64_t n_slots, const uint32_t* __restrict__ flags,
                                                    const float* __restrict__ aabb, const uint32_t* __restrict__ group,
                                                    const uint32_t* __restrict__ mask, const uint32_t* __restrict__ global_of_slot,
                                                    uint32_t axis, uint32_t nranks, RouteScalars* sc, float4* __restrict__ send)
{
    __shared__ uint32_t cnt[kMaxSlabs];
    __shared__ unsigned long long base[kMaxSlabs];
    __shared__ float cuts[kMaxSlabs + 1];
    if (threadIdx.x <= nranks) cuts[threadIdx.x] = sc->cuts[threadIdx.x];
    // one workgroup handles one contiguous chunk of 256 slots per round (uniform trip count: barriers inside)
    const uint64_t rounds = (n_slots + static_cast<uint64_t>(gridDim.x) * 256u - 1u) / (static_cast<uint64_t>(gridDim.x) * 256u);
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t s = (r * gridDim.x + blockIdx.x) * 256ull + threadIdx.x;
        __syncthreads();
        if (threadIdx.x < kMaxSlabs) cnt[threadIdx.x] = 0;
        __syncthreads();
        uint32_t f = 0, d_lo = 1, d_hi = 0;
        if (s < n_slots) {
            f = flags[s];
            if (is_body(f)) {
                d_lo = slab_of(cuts, nranks, aabb[6 * s + axis]);
                d_hi = max(d_lo, slab_of(cuts, nranks, aabb[6 * s + 3 + axis]));
            }
        }
        for (uint32_t d = d_lo; d <= d_hi; ++d) atomicAdd(&cnt[d], 1u);
        __syncthreads();
        if (threadIdx.x < nranks) {
            const uint32_t c = cnt[threadIdx.x];
            base[threadIdx.x] = c ? atomicAdd(&sc->cursor[threadIdx.x], static_cast<unsigned long long>(c)) : 0ull;
            cnt[threadIdx.x] = 0;
        }
        __syncthreads();
        if (d_lo <= d_hi) {
            const float* b = aabb + 6 * s;
            const float4 r0 = make_float4(b[0], b[1], b[2], __uint_as_float(global_of_slot[s]));
            const float4 r1 = make_float4(b[3], b[4], b[5], 0.0f);
            const float4 r2 = make_float4(__uint_as_float(group[s]), __uint_as_float(mask[s]),
                                          __uint_as_float((f & kTypeMask) == 1u ? 1u : 0u), 0.0f);
            for (uint32_t d = d_lo; d <= d_hi; ++d) {
                const unsigned long long pos = base[d] + atomicAdd(&cnt[d], 1u);
                send[3ull * pos] = r0;
                send[3ull * pos + 1] = r1;
                send[3ull * pos + 2] = r2;
            }
        }
    }
}

// Received records -> the arrays the broadphase kernels read (flags: valid + Static / Dynamic, aabb, group, mask, id).
__global__ void __launch_bounds__(256) k_route_unpack(uint64_t n, const float4* __restrict__ rec, uint32_t* __restrict__ flags,
                                                      float* __restrict__ aabb, uint32_t* __restrict__ group,
                                                      uint32_t* __restrict__ mask, uint32_t* __restrict__ entity)
{
    const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    const float4 r0 = rec[3 * i], r1 = rec[3 * i + 1], r2 = rec[3 * i + 2];
    flags[i] = kValid | (__float_as_uint(r2.z) ? 1u : 2u);
    float* b = aabb + 6 * i;
    b[0] = r0.x;
    b[1] = r0.y;
    b[2] = r0.z;
    b[3] = r1.x;
    b[4] = r1.y;
    b[5] = r1.z;
    group[i] = __float_as_uint(r2.x);
    mask[i] = __float_as_uint(r2.y);
    entity[i] = __float_as_uint(r0.w);
}

__global__ void __launch_bounds__(256) k_route_bounds(uint64_t n_slots, const uint32_t* __restrict__ flags,
                                                      const float* __restrict__ aabb, RouteScalars* sc)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    unsigned long long cnt = 0;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t s = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; s < n_slots; s += stride) {
        if (!is_body(flags[s])) continue;
        for (int a = 0; a < 3; ++a) {
            mn[a] = fminf(mn[a], aabb[6 * s + a]);
            mx[a] = fmaxf(mx[a], aabb[6 * s + 3 + a]);
        }
        ++cnt;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        for (int a = 0; a < 3; ++a) {
            mn[a] = fminf(mn[a], __shfl_down(mn[a], off, 64));
            mx[a] = fmaxf(mx[a], __shfl_down(mx[a], off, 64));
        }
        cnt += __shfl_down(cnt, off, 64);
    }
    // across the 4 waves through LDS, then ONE set of atomics per workgroup (per-wave atomics on the same 7 words cost
    // 1.4 ms for 2 M bodies: a contended word takes ~10^8 atomics/s)
    __shared__ float red[4][6];
    __shared__ unsigned long long red_cnt[4];
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0) {
        for (int a = 0; a < 3; ++a) {
            red[wave][a] = mn[a];
            red[wave][3 + a] = mx[a];
        }
        red_cnt[wave] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long total = red_cnt[0] + red_cnt[1] + red_cnt[2] + red_cnt[3];
        if (total) {
            for (int a = 0; a < 3; ++a) {
                atomicMin(&sc->mn[a], f2ord(fminf(fminf(red[0][a], red[1][a]), fminf(red[2][a], red[3][a]))));
                atomicMax(&sc->mx[a], f2ord(fmaxf(fmaxf(red[0][3 + a], red[1][3 + a]), fmaxf(red[2][3 + a], red[3][3 + a]))));
            }
            atomicAdd(&sc->n_bodies, total);
        }
    }
}

// Balanced cuts: where do the bodies' min corners lie along the axis?  LDS histogram per workgroup, merged with atomics.
__global__ void __launch_bounds__(256) k_route_hist(uint64_t n_slots, const uint32_t* __restrict__ flags,
                                                    const float* __restrict__ aabb, uint32_t axis, float lo, float inv_width,
                                                    uint32_t bins, unsigned long long* __restrict__ hist)
{
    __shared__ uint32_t h[kMaxHistBins];
    for (uint32_t k = threadIdx.x; k < bins; k += blockDim.x) h[k] = 0;
    __syncthreads();
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t s = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; s < n_slots; s += stride) {
        if (!is_body(flags[s])) continue;
        const float t = (aabb[6 * s + axis] - lo) * inv_width;
        const uint32_t b = t >= 0.0f ? min(static_cast<uint32_t>(fminf(t, 4.0e9f)), bins - 1u) : 0u; // NaN -> bin 0
        atomicAdd(&h[b], 1u);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < bins; k += blockDim.x) {
        if (h[k]) atomicAdd(&hist[k], static_cast<unsigned long long>(h[k]));
    }
}

inline uint32_t grid_for_slots(uint64_t n) { return static_cast<uint32_t>(std::min<uint64_t>((n + 255) / 256, 4096)); }

} // namespace

int ShardRouter::fail(int code, const char* what, hipError_t e)
{
    error_ = std::string(what) + ": " + hipGetErrorString(e);
    return code;
}

#define RT_TRY(expr)                                                                                       \
    do {                                                                                                   \
        const hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? BGE_ERR_OOM : BGE_ERR_HIP, #expr, e_); \
    } while (0)

int ShardRouter::ensure(void** p, size_t* have, size_t need)
{
    if (*have >= need && *p) return BGE_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    const size_t bytes = std::max<size_t>(need + need / 4, 256);
    RT_TRY(hipMalloc(p, bytes));
    *have = bytes;
    return BGE_OK;
}

void ShardRouter::release()
{
    for (void** p : {&scalars_, &hist_, &rx_flags_, &rx_aabb_, &rx_group_, &rx_mask_, &rx_entity_}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    rx_flags_b_ = rx_aabb_b_ = rx_group_b_ = rx_mask_b_ = rx_entity_b_ = 0;
    counted_ = false;
}

int ShardRouter::bounds(hipStream_t stream, const WorldView& w, uint64_t n_slots, float mn[3], float mx[3], uint64_t* n_bodies)
{
    if (!scalars_) RT_TRY(hipMalloc(&scalars_, sizeof(RouteScalars)));
    RouteScalars* sc = static_cast<RouteScalars*>(scalars_);
    hipLaunchKernelGGL(k_route_reset, dim3(1), dim3(256), 0, stream, sc);
    if (n_slots) hipLaunchKernelGGL(k_route_bounds, dim3(std::min<uint32_t>(grid_for_slots(n_slots), 1024)), dim3(256), 0, stream, n_slots, w.flags, w.aabb, sc);
    RT_TRY(hipGetLastError());
    RouteScalars host;
    RT_TRY(hipMemcpyAsync(&host, sc, sizeof host, hipMemcpyDeviceToHost, stream));
    RT_TRY(hipStreamSynchronize(stream));
    for (int a = 0; a < 3; ++a) {
        mn[a] = host.n_bodies ? ord2f_host(host.mn[a]) : INFINITY;
        mx[a] = host.n_bodies ? ord2f_host(host.mx[a]) : -INFINITY;
    }
    if (n_bodies) *n_bodies = host.n_bodies;
    return BGE_OK;
}

int ShardRouter::histogram(hipStream_t stream, const WorldView& w, uint64_t n_slots, uint32_t axis, float lo, float hi, uint32_t bins,
                           uint64_t* hist_host)
{
    if (axis > 2 || bins == 0 || bins > kMaxHistBins || !(hi >= lo)) {
        error_ = "histogram needs axis 0..2, 1 <= bins <= 4096 and lo <= hi";
        return BGE_ERR_INVALID;
    }
    if (!hist_) RT_TRY(hipMalloc(&hist_, kMaxHistBins * sizeof(unsigned long long)));
    RT_TRY(hipMemsetAsync(hist_, 0, bins * sizeof(unsigned long long), stream));
    const float width = (hi - lo) / static_cast<float>(bins);
    const float inv_width = width > 0.0f ? 1.0f / width : 0.0f;
    if (n_slots) {
        // 256 workgroups: each merges up to `bins` LDS counters into the global histogram with atomics
        hipLaunchKernelGGL(k_route_hist, dim3(std::min<uint32_t>(grid_for_slots(n_slots), 256)), dim3(256), 0, stream, n_slots, w.flags,
                           w.aabb, axis, lo, inv_width, bins, static_cast<unsigned long long*>(hist_));
        RT_TRY(hipGetLastError());
    }
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "64-bit counters");
    RT_TRY(hipMemcpyAsync(hist_host, hist_, bins * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    RT_TRY(hipStreamSynchronize(stream));
    return BGE_OK;
}

int ShardRouter::count(hipStream_t stream, const WorldView& w, uint64_t n_slots, uint32_t axis, uint32_t nranks,
                       const float* cuts_host, uint64_t* counts_host)
{
    counted_ = false;
    if (axis > 2 || nranks == 0 || nranks > kMaxSlabs) {
        error_ = "axis must be 0..2 and 1 <= nranks <= 64";
        return BGE_ERR_INVALID;
    }
    for (uint32_t k = 1; k < nranks; ++k) {
        if (std::isnan(cuts_host[k]) || (k > 1 && cuts_host[k] < cuts_host[k - 1])) {
            error_ = "slab cuts must be non-decreasing and not NaN";
            return BGE_ERR_INVALID;
        }
    }
    if (!scalars_) RT_TRY(hipMalloc(&scalars_, sizeof(RouteScalars)));
    RouteScalars* sc = static_cast<RouteScalars*>(scalars_);
    hipLaunchKernelGGL(k_route_reset, dim3(1), dim3(256), 0, stream, sc);
    float cuts[kMaxSlabs + 1];
    for (uint32_t k = 0; k <= kMaxSlabs; ++k) cuts[k] = (k >= 1 && k < nranks) ? cuts_host[k] : (k == 0 ? -INFINITY : INFINITY);
    RT_TRY(hipMemcpyAsync(sc->cuts, cuts, sizeof cuts, hipMemcpyHostToDevice, stream));
    if (n_slots) {
        hipLaunchKernelGGL(k_route_count, dim3(grid_for_slots(n_slots)), dim3(256), 0, stream, n_slots, w.flags, w.aabb, axis, nranks, sc);
    }
    RT_TRY(hipGetLastError());
    unsigned long long counts[kMaxSlabs];
    RT_TRY(hipMemcpyAsync(counts, sc->count, sizeof counts, hipMemcpyDeviceToHost, stream));
    RT_TRY(hipStreamSynchronize(stream));
    unsigned long long cursor[kMaxSlabs] = {};
    total_ = 0;
    for (uint32_t d = 0; d < nranks; ++d) {
        counts_host[d] = counts[d];
        cursor[d] = total_;
        total_ += counts[d];
    }
    RT_TRY(hipMemcpyAsync(sc->cursor, cursor, sizeof cursor, hipMemcpyHostToDevice, stream));
    RT_TRY(hipStreamSynchronize(stream)); // `cursor` is a stack array
    axis_ = axis;
    nranks_ = nranks;
    counted_ = true;
    return BGE_OK;
}

int ShardRouter::pack(hipStream_t stream, const WorldView& w, uint64_t n_slots, const uint32_t* global_of_slot, void* send_device)
{
    if (!counted_) {
        error_ = "bge_world_bp_route has not been called for this tick";
        return BGE_ERR_STATE;
    }
    counted_ = false; // the cursors are consumed
    if (total_ == 0 || n_slots == 0) return BGE_OK;
    if (!send_device) {
        error_ = "send buffer is NULL";
        return BGE_ERR_INVALID;
    }
    hipLaunchKernelGGL(k_route_pack, dim3(grid_for_slots(n_slots)), dim3(256), 0, stream, n_slots, w.flags, w.aabb, w.group, w.mask,
                       global_of_slot, axis_, nranks_, static_cast<RouteScalars*>(scalars_), static_cast<float4*>(send_device));
    RT_TRY(hipGetLastError());
    return BGE_OK;
}

int ShardRouter::unpack(hipStream_t stream, const void* records_device, uint64_t n, WorldView* view, const uint32_t** entity_ids)
{
    const size_t m = std::max<uint64_t>(n, 1);
    if (int rc = ensure(&rx_flags_, &rx_flags_b_, m * 4)) return rc;
    if (int rc = ensure(&rx_aabb_, &rx_aabb_b_, m * 24)) return rc;
    if (int rc = ensure(&rx_group_, &rx_group_b_, m * 4)) return rc;
    if (int rc = ensure(&rx_mask_, &rx_mask_b_, m * 4)) return rc;
    if (int rc = ensure(&rx_entity_, &rx_entity_b_, m * 4)) return rc;
    if (n) {
        hipLaunchKernelGGL(k_route_unpack, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, stream, n,
                           static_cast<const float4*>(records_device), static_cast<uint32_t*>(rx_flags_), static_cast<float*>(rx_aabb_),
                           static_cast<uint32_t*>(rx_group_), static_cast<uint32_t*>(rx_mask_), static_cast<uint32_t*>(rx_entity_));
        RT_TRY(hipGetLastError());
    }
    *view = WorldView{};
    view->flags = static_cast<uint32_t*>(rx_flags_);
    view->aabb = static_cast<float*>(rx_aabb_);
    view->group = static_cast<uint32_t*>(rx_group_);
    view->mask = static_cast<uint32_t*>(rx_mask_);
    *entity_ids = static_cast<const uint32_t*>(rx_entity_);
    return BGE_OK;
}

} // namespace bge
