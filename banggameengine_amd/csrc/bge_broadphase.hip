// bge_broadphase.hip — AABB broadphase on gfx950: uniform grid + counting sort + wave-compacted pair list.
//
// Replaces what the reference gets from Bullet's btDbvtBroadphase (src/physics/PhysicsSystem.cpp:124;
// updateAabbs / calculateOverlappingPairs inside stepSimulation, :863).  The pair SET it produces is the
// history-free core of Bullet's pair cache (see oracle/broadphase_ref.h for the exact specification):
//   AABBs overlap (non-strict) AND (groupA & maskB) && (groupB & maskA) AND not both Static.
//
// Per tick, all on the world's stream, no host round trip:
//   1. k_bp_bounds   wave64 __shfl reductions of the scene bounds + LDS histogram of AABB extents
//   2. k_bp_params   one wave derives the grid: the cell size that minimises the expected number of AABB tests
//                    given the histogram of body extents (grown until the padded grid fits the table);
//                    bodies wider than a cell are "large" and handled by step 7
//   3. k_bp_count    cell index of each body's AABB min corner; atomic histogram, the returned value is the
//                    body's rank inside its cell (so the scatter needs no second atomic)
//   4. scan          exclusive prefix sum of the histogram (3 small kernels)
//   5. k_bp_scatter  bodies sorted by cell into 32-byte records (min xyz | slot, max xyz | cell)
//   6. k_bp_pairs    every body scans the 5 contiguous runs of the 14 "forward" neighbour cells; overlapping
//                    pairs are compacted per wave with a ballot + one atomic per wave per iteration
//   7. k_bp_large    large bodies against everything
// A small body's AABB is narrower than one cell (by a 2^-20 margin that dominates the f64 rounding of the
// cell coordinates), so two overlapping small bodies sit in cells that differ by at most one per axis; scanning only forward neighbours (and, inside the own cell, only later records) reports
// each pair once.  The kernels are bound by L2/Infinity-Cache traffic of the sorted records, not by HBM.
#include "bge_broadphase.hpp"

#include <hip/hip_runtime.h>

#include "../../include/bge_world.h"
#include "bge_flatten.hpp"

namespace bge {

namespace {

struct GridParams {
    float origin[3];
    double inv_cell;     // cell coordinates are formed in f64 so that their error is << 2^-20 of a cell
    float cell;
    float small_limit;   // bodies wider than this (cell * (1 - 2^-20)) go to the large list
    uint32_t dim_x, dim_xy; // padded dims: x, x*y
    uint32_t n_cells;
    uint32_t n_bodies;
};

// Extent histogram: bin = bits 21..30 of the (positive) float = 8 exponent bits + 2 mantissa bits, so
// consecutive bin edges are 19-25 % apart.
constexpr uint32_t kExtentBins = 1024;
__host__ __device__ inline uint32_t extent_bin(float e) { return (__builtin_bit_cast(uint32_t, e) >> 21) & (kExtentBins - 1u); }
__host__ __device__ inline float extent_bin_upper(uint32_t b) { return __builtin_bit_cast(float, (b + 1u) << 21); }

struct Accum {
    uint32_t min_bits[3]; // ordered-uint encoding of floats
    uint32_t max_bits[3];
    uint32_t n_bodies;
    uint32_t n_large;
    unsigned long long n_pairs;
    GridParams grid;
    uint32_t extent_hist[kExtentBins]; // bodies per extent bin (bin = float exponent + 2 mantissa bits)
};

constexpr uint32_t kLargeCell = 0xffffffffu;

__device__ __forceinline__ uint32_t f2ord(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o)
{
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}

__device__ __forceinline__ bool is_body(uint32_t f) { return (f & kValid) && (f & kTypeMask) != 0; }

__global__ void k_bp_reset(Accum* acc)
{
    for (uint32_t b = threadIdx.x; b < kExtentBins; b += blockDim.x) acc->extent_hist[b] = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int a = 0; a < 3; ++a) {
            acc->min_bits[a] = 0xffffffffu;
            acc->max_bits[a] = 0u;
        }
        acc->n_bodies = 0;
        acc->n_large = 0;
        acc->n_pairs = 0;
    }
}

__global__ void __launch_bounds__(256) k_bp_bounds(uint64_t n_slots, const uint32_t* __restrict__ flags,
                                                   const float* __restrict__ aabb, Accum* acc)
{
    __shared__ uint32_t hist[kExtentBins];
    __shared__ float red[4][6];
    __shared__ uint32_t red_cnt[4];
    for (uint32_t k = threadIdx.x; k < kExtentBins; k += blockDim.x) hist[k] = 0;
    __syncthreads();

    float mn[3] = {INFINITY, INFINITY, INFINITY};
    float mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t cnt = 0;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t s = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; s < n_slots; s += stride) {
        if (!is_body(flags[s])) continue;
        const float2* b = reinterpret_cast<const float2*>(aabb + 6 * s);
        const float2 b0 = b[0], b1 = b[1], b2 = b[2]; // min.x min.y | min.z max.x | max.y max.z
        mn[0] = fminf(mn[0], b0.x);
        mn[1] = fminf(mn[1], b0.y);
        mn[2] = fminf(mn[2], b1.x);
        mx[0] = fmaxf(mx[0], b1.y);
        mx[1] = fmaxf(mx[1], b2.x);
        mx[2] = fmaxf(mx[2], b2.y);
        const float e = fmaxf(fmaxf(b1.y - b0.x, b2.x - b0.y), b2.y - b1.x);
        atomicAdd(&hist[(e > 0.0f && e < INFINITY) ? extent_bin(e) : 0u], 1u); // LDS atomic
        cnt += 1;
    }
    // wave64 __shfl reductions, then across the 4 waves through LDS, then one set of atomics per workgroup
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mn[a] = fminf(mn[a], __shfl_down(mn[a], off, 64));
            mx[a] = fmaxf(mx[a], __shfl_down(mx[a], off, 64));
        }
        cnt += __shfl_down(cnt, off, 64);
    }
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
        uint32_t total = 0;
        for (int wv = 0; wv < 4; ++wv) total += red_cnt[wv];
        if (total) {
            for (int a = 0; a < 3; ++a) {
                const float lo = fminf(fminf(red[0][a], red[1][a]), fminf(red[2][a], red[3][a]));
                const float hi = fmaxf(fmaxf(red[0][3 + a], red[1][3 + a]), fmaxf(red[2][3 + a], red[3][3 + a]));
                atomicMin(&acc->min_bits[a], f2ord(lo));
                atomicMax(&acc->max_bits[a], f2ord(hi));
            }
            atomicAdd(&acc->n_bodies, total);
        }
    }
    for (uint32_t k = threadIdx.x; k < kExtentBins; k += blockDim.x) {
        const uint32_t h = hist[k];
        if (h) atomicAdd(&acc->extent_hist[k], h);
    }
}

// One wave: pick the cell size that minimises the expected number of AABB tests,
//   tests(c) = n_small(c)^2 / volume * c^3 * 13.5   (13 forward cells + half the own cell)
//            + n_large(c) * n                        (k_bp_large: every large body against everything)
// over the histogram's bin edges c (a body is "small" when its widest side is < c); then grow the cell
// until the padded grid fits the table.
__global__ void __launch_bounds__(64) k_bp_params(Accum* acc, uint32_t max_cells)
{
    __shared__ uint32_t below[kExtentBins + 1]; // exclusive prefix: bodies in bins < b
    const uint32_t lane = threadIdx.x;
    {
        // exclusive prefix of the histogram: 16 consecutive bins per lane + a wave64 __shfl_up scan of the lane totals
        uint32_t h[16];
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            h[k] = acc->extent_hist[lane * 16 + k];
            sum += h[k];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= static_cast<uint32_t>(off)) incl += t;
        }
        uint32_t run = incl - sum;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            below[lane * 16 + k] = run;
            run += h[k];
        }
        if (lane == 63) below[kExtentBins] = run;
    }
    __syncthreads();
    const uint32_t n = acc->n_bodies;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    double volume = 1.0;
    if (n) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = ord2f(acc->min_bits[a]);
            hi[a] = ord2f(acc->max_bits[a]);
            if (!(lo[a] > -1.0e30f)) lo[a] = -1.0e30f;
            if (!(hi[a] < 1.0e30f)) hi[a] = 1.0e30f;
            volume *= fmax(static_cast<double>(hi[a]) - static_cast<double>(lo[a]), 1.0e-3);
        }
    }
    // each lane evaluates 16 candidate edges, then a wave64 __shfl min-reduction picks the best
    double best_cost = 1.0e300;
    uint32_t best_bin = kExtentBins - 2;
    for (uint32_t b = lane; b < kExtentBins - 1; b += 64) {
        const uint32_t small = below[b + 1];
        if (small == 0 && n != 0 && b + 2 < kExtentBins) continue; // an edge below every body: nothing would be small
        const double c = static_cast<double>(extent_bin_upper(b));
        if (!(c > 0.0) || !(c < 1.0e30)) continue;
        const double large = static_cast<double>(n - small);
        const double cost = static_cast<double>(small) * small / volume * c * c * c * 13.5 + large * n + 1.0e-9 * b;
        if (cost < best_cost) {
            best_cost = cost;
            best_bin = b;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double oc = __shfl_down(best_cost, off, 64);
        const uint32_t ob = __shfl_down(best_bin, off, 64);
        if (oc < best_cost) {
            best_cost = oc;
            best_bin = ob;
        }
    }
    if (lane != 0) return;

    GridParams g;
    g.n_bodies = n;
    float cell = n ? extent_bin_upper(best_bin) : 1.0f;
    cell = fminf(fmaxf(cell, 1.0e-6f), 1.0e30f);
    // padded grid must fit the table; grow the cell until it does
    uint32_t dx = 4, dy = 4, dz = 4;
    for (int it = 0; it < 400; ++it) {
        const float fx = floorf((hi[0] - lo[0]) / cell), fy = floorf((hi[1] - lo[1]) / cell), fz = floorf((hi[2] - lo[2]) / cell);
        const double cells = (static_cast<double>(fx) + 4.0) * (static_cast<double>(fy) + 4.0) * (static_cast<double>(fz) + 4.0);
        const bool axes_ok = fx < 1048576.0f && fy < 1048576.0f && fz < 1048576.0f; // per-axis index stays exact
        if (axes_ok && cells <= static_cast<double>(max_cells)) {
            // one empty cell below, two above (one of them slack for the f32/f64 rounding of the top edge)
            dx = static_cast<uint32_t>(fx) + 4;
            dy = static_cast<uint32_t>(fy) + 4;
            dz = static_cast<uint32_t>(fz) + 4;
            break;
        }
        cell *= 1.25f;
    }
    g.cell = cell;
    g.small_limit = cell * (1.0f - 0x1p-20f);
    g.inv_cell = 1.0 / static_cast<double>(cell);
    for (int a = 0; a < 3; ++a) g.origin[a] = lo[a];
    g.dim_x = dx;
    g.dim_xy = dx * dy;
    g.n_cells = dx * dy * dz;
    acc->grid = g;
}

__device__ __forceinline__ uint32_t cell_of(const GridParams& g, const float* mn)
{
    // +1: the padding cell; clamped so that garbage (NaN / out of range) stays inside the table
    uint32_t c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double t = floor((static_cast<double>(mn[a]) - static_cast<double>(g.origin[a])) * g.inv_cell);
        t = fmin(fmax(t, 0.0), 4.0e9);
        c[a] = static_cast<uint32_t>(t) + 1u;
    }
    const uint32_t dy = g.dim_xy / g.dim_x;
    const uint32_t dz = g.n_cells / g.dim_xy;
    c[0] = min(c[0], g.dim_x - 2u);
    c[1] = min(c[1], dy - 2u);
    c[2] = min(c[2], dz - 2u);
    return c[0] + g.dim_x * c[1] + g.dim_xy * c[2];
}

__global__ void __launch_bounds__(256) k_bp_count(uint64_t n_slots, const uint32_t* __restrict__ flags,
                                                  const float* __restrict__ aabb, Accum* acc,
                                                  uint32_t* __restrict__ cell_count, uint32_t* __restrict__ body_cell,
                                                  uint32_t* __restrict__ body_rank, uint32_t* __restrict__ large_list)
{
    const uint64_t s = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (s >= n_slots) return;
    if (!is_body(flags[s])) return;
    const GridParams g = acc->grid;
    const float* b = aabb + 6 * s;
    const float ext = fmaxf(fmaxf(b[3] - b[0], b[4] - b[1]), b[5] - b[2]);
    if (!(ext <= g.small_limit)) {
        body_cell[s] = kLargeCell;
        large_list[atomicAdd(&acc->n_large, 1u)] = static_cast<uint32_t>(s);
        return;
    }
    const uint32_t c = cell_of(g, b);
    body_cell[s] = c;
    body_rank[s] = atomicAdd(&cell_count[c], 1u);
}

// ---- exclusive scan of cell_count[0..n) into cell_start[0..n], n = padded to kScanBlock multiples by the caller
constexpr uint32_t kScanBlock = 2048; // 256 threads x 8

__global__ void __launch_bounds__(256) k_scan_blocks(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                     uint32_t* __restrict__ block_sums, uint32_t n)
{
    __shared__ uint32_t wave_tot[4];
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * 8;
    uint32_t v[8];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        sum += v[k];
    }
    // inclusive scan of per-thread sums across the wave
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if ((threadIdx.x & 63u) >= static_cast<uint32_t>(off)) incl += t;
    }
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 63u) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0;
    for (uint32_t k = 0; k < wave; ++k) wave_base += wave_tot[k];
    uint32_t run = wave_base + incl - sum;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 255) block_sums[blockIdx.x] = wave_base + incl;
}

__global__ void __launch_bounds__(256) k_scan_sums(uint32_t* __restrict__ block_sums, uint32_t n_blocks)
{
    // single workgroup: serial over chunks of 256, wave scan inside
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_blocks; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_blocks ? block_sums[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if ((threadIdx.x & 63u) >= static_cast<uint32_t>(off)) incl += t;
        }
        const uint32_t wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63u) == 63u) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t wave_base = carry;
        for (uint32_t k = 0; k < wave; ++k) wave_base += wave_tot[k];
        if (i < n_blocks) block_sums[i] = wave_base + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry = wave_base + incl;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_scan_add(uint32_t* __restrict__ out, const uint32_t* __restrict__ block_sums, uint32_t n)
{
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * 8;
    const uint32_t add = block_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (base + k < n) out[base + k] += add;
    }
}

__global__ void __launch_bounds__(256) k_bp_scatter(uint64_t n_slots, const uint32_t* __restrict__ flags,
                                                    const float* __restrict__ aabb, const uint32_t* __restrict__ cell_start,
                                                    const uint32_t* __restrict__ body_cell, const uint32_t* __restrict__ body_rank,
                                                    float4* __restrict__ sorted)
{
    const uint64_t s = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (s >= n_slots) return;
    if (!is_body(flags[s])) return;
    const uint32_t c = body_cell[s];
    if (c == kLargeCell) return;
    const uint32_t pos = cell_start[c] + body_rank[s];
    const float* b = aabb + 6 * s;
    sorted[2ull * pos] = make_float4(b[0], b[1], b[2], __uint_as_float(static_cast<uint32_t>(s)));
    sorted[2ull * pos + 1] = make_float4(b[3], b[4], b[5], __uint_as_float(c));
}

struct PairSink {
    unsigned long long* count;
    uint2* pairs;
    uint64_t cap;
};

// Wave-compacted append with per-wave staging in LDS.  Each call ballots the hits of the wave, packs them
// behind the wave's staging cursor (an SGPR-uniform count), and whenever 64 pairs are staged writes them out
// as one contiguous 512-byte burst behind ONE global atomic.  A single counter word saturates near 10^8
// atomics/s, so one atomic per hit-bearing iteration (the first version) cost 39 ms for 12.6 M pairs.
constexpr uint32_t kStage = 320;     // 255 carried + 64 new
constexpr uint32_t kFlush = 256;     // pairs written per global atomic (4 per lane, 2 KiB contiguous)

// LDS traffic between lanes of ONE wave: the hardware keeps a wave's DS operations in order; this keeps the
// compiler from moving them across the hand-over point.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct WaveStage {
    uint2* buf;     // LDS, kStage entries owned by this wave
    uint32_t fill;  // wave-uniform
};

__device__ __forceinline__ void stage_flush(const PairSink& sink, WaveStage& st, uint32_t n_out)
{
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(sink.count, static_cast<unsigned long long>(n_out));
    base = __shfl(base, 0, 64);
    for (uint32_t k = lane; k < n_out; k += 64u) {
        if (base + k < sink.cap) sink.pairs[base + k] = st.buf[k];
    }
    // carry the remainder (< 64 entries) down to the front
    const uint32_t rest = st.fill - n_out;
    uint2 carry = make_uint2(0, 0);
    if (lane < rest) carry = st.buf[n_out + lane];
    wave_sync();
    if (lane < rest) st.buf[lane] = carry;
    wave_sync();
    st.fill = rest;
}

__device__ __forceinline__ void emit_pairs(const PairSink& sink, WaveStage& st, bool hit, uint32_t ea, uint32_t eb)
{
    const unsigned long long m = __ballot(hit);
    if (m == 0) return;
    const uint32_t lane = threadIdx.x & 63u;
    if (hit) {
        const unsigned long long below = m & ((1ull << lane) - 1ull);
        st.buf[st.fill + static_cast<uint32_t>(__popcll(below))] = make_uint2(min(ea, eb), max(ea, eb));
    }
    st.fill += static_cast<uint32_t>(__popcll(m));
    wave_sync();
    if (st.fill >= kFlush) stage_flush(sink, st, kFlush);
}

__device__ __forceinline__ bool overlap(const float4& alo, const float4& ahi, const float4& blo, const float4& bhi)
{
    return alo.x <= bhi.x && ahi.x >= blo.x && alo.y <= bhi.y && ahi.y >= blo.y && alo.z <= bhi.z && ahi.z >= blo.z;
}

__device__ __forceinline__ bool filter_ok(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ group,
                                          const uint32_t* __restrict__ mask, uint32_t sa, uint32_t sb)
{
    const bool both_static = (flags[sa] & kTypeMask) == 1u && (flags[sb] & kTypeMask) == 1u;
    return !both_static && (group[sa] & mask[sb]) != 0 && (group[sb] & mask[sa]) != 0;
}

__global__ void __launch_bounds__(256) k_bp_pairs(const Accum* __restrict__ acc, const uint32_t* __restrict__ cell_start,
                                                  const float4* __restrict__ sorted, const uint32_t* __restrict__ flags,
                                                  const uint32_t* __restrict__ group, const uint32_t* __restrict__ mask,
                                                  const uint32_t* __restrict__ entity_of_slot, PairSink sink,
                                                  uint32_t n_sorted_max)
{
    __shared__ uint2 stage_lds[4][kStage];
    const GridParams g = acc->grid;
    const uint32_t n_sorted = g.n_bodies - acc->n_large;
    WaveStage st{stage_lds[threadIdx.x >> 6], 0u};
    const uint32_t stride = gridDim.x * blockDim.x;

    // grid-stride over the sorted bodies: a wave keeps its staging buffer across chunks, so the tail flush
    // (one atomic) is paid once per wave, not once per 64 bodies
    for (uint32_t i0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n_sorted; i0 += stride) {
        const uint32_t i = i0 + (threadIdx.x & 63u);
        const bool active = i < n_sorted;
        float4 lo = make_float4(0, 0, 0, 0), hi = lo;
        uint32_t cell = 0;
        if (active) {
            lo = sorted[2ull * i];
            hi = sorted[2ull * i + 1];
            cell = __float_as_uint(hi.w);
        }
        const uint32_t slot_i = __float_as_uint(lo.w);

        // 5 contiguous runs cover the own cell's later records and the 13 forward neighbour cells
#pragma unroll 1
        for (int row = 0; row < 5; ++row) {
            uint32_t j = 0, end = 0;
            if (active) {
                if (row == 0) {
                    j = i + 1;
                    end = cell_start[cell + 2];
                } else {
                    const int dy = (row == 1) ? 1 : (row - 3); // rows 2,3,4 -> dy = -1,0,1 at dz = 1
                    const int dz = (row == 1) ? 0 : 1;
                    const uint32_t c = cell + static_cast<uint32_t>(dy * static_cast<int>(g.dim_x)) + static_cast<uint32_t>(dz) * g.dim_xy;
                    j = cell_start[c - 1];
                    end = cell_start[c + 2];
                }
            }
            while (__any(j < end)) {
                // two candidates per trip: the loads of both are in flight together
                bool hit0 = false, hit1 = false;
                uint32_t s0 = 0, s1 = 0;
                if (j < end) {
                    const bool two = j + 1 < end;
                    const float4 alo = sorted[2ull * j], ahi = sorted[2ull * j + 1];
                    float4 blo = alo, bhi = ahi;
                    if (two) {
                        blo = sorted[2ull * j + 2];
                        bhi = sorted[2ull * j + 3];
                    }
                    if (overlap(lo, hi, alo, ahi)) {
                        s0 = __float_as_uint(alo.w);
                        hit0 = filter_ok(flags, group, mask, slot_i, s0);
                    }
                    if (two && overlap(lo, hi, blo, bhi)) {
                        s1 = __float_as_uint(blo.w);
                        hit1 = filter_ok(flags, group, mask, slot_i, s1);
                    }
                    j += 2;
                }
                if (__any(hit0 || hit1)) {
                    const uint32_t ei = (hit0 || hit1) ? entity_of_slot[slot_i] : 0u;
                    emit_pairs(sink, st, hit0, ei, hit0 ? entity_of_slot[s0] : 0u);
                    emit_pairs(sink, st, hit1, ei, hit1 ? entity_of_slot[s1] : 0u);
                }
            }
        }
    }
    if (st.fill) stage_flush(sink, st, st.fill);
}

__global__ void __launch_bounds__(256) k_bp_large(uint64_t n_slots, const Accum* __restrict__ acc,
                                                  const uint32_t* __restrict__ large_list, const uint32_t* __restrict__ body_cell,
                                                  const float* __restrict__ aabb, const uint32_t* __restrict__ flags,
                                                  const uint32_t* __restrict__ group, const uint32_t* __restrict__ mask,
                                                  const uint32_t* __restrict__ entity_of_slot, PairSink sink)
{
    __shared__ uint2 stage_lds[4][kStage];
    const uint32_t n_large = acc->n_large;
    if (n_large == 0) return;
    WaveStage st{stage_lds[threadIdx.x >> 6], 0u};
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    const uint64_t first = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    // uniform trip count per wave so that the ballot in emit_pairs sees every lane
    const uint64_t rounds = (n_slots + stride - 1) / stride;
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t s = first + r * stride;
        const bool body = s < n_slots && is_body(flags[s]);
        float4 lo = make_float4(0, 0, 0, 0), hi = lo;
        bool s_large = false;
        if (body) {
            const float* b = aabb + 6 * s;
            lo = make_float4(b[0], b[1], b[2], 0);
            hi = make_float4(b[3], b[4], b[5], 0);
            s_large = body_cell[s] == kLargeCell;
        }
        for (uint32_t k = 0; k < n_large; ++k) {
            const uint32_t L = large_list[k];
            const float* b = aabb + 6ull * L;
            const float4 llo = make_float4(b[0], b[1], b[2], 0), lhi = make_float4(b[3], b[4], b[5], 0);
            // large-vs-small: reported from the small body's side; large-vs-large: from the lower slot's side
            bool hit = body && s != L && (!s_large || s < L) && overlap(lo, hi, llo, lhi);
            if (hit) hit = filter_ok(flags, group, mask, static_cast<uint32_t>(s), L);
            emit_pairs(sink, st, hit, hit ? entity_of_slot[s] : 0u, hit ? entity_of_slot[L] : 0u);
        }
    }
    if (st.fill) stage_flush(sink, st, st.fill);
}

inline uint32_t blocks_for(uint64_t n, uint32_t per) { return static_cast<uint32_t>((n + per - 1) / per); }

} // namespace

// ------------------------------------------------------------------ host side
int Broadphase::fail(int code, const char* what, hipError_t e)
{
    error_ = std::string(what) + ": " + hipGetErrorString(e);
    return code;
}

void Broadphase::release()
{
    for (void** p : {&pairs_, &counters_, &cell_count_, &cell_start_, &scan_tmp_, &sorted_slot_, &sorted_aabb_, &body_cell_,
                     &large_list_}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    n_slots_ = capacity_ = 0;
    table_size_ = 0;
    ran_ = false;
}

#define BP_TRY(expr)                                                       \
    do {                                                                   \
        const hipError_t e_ = (expr);                                      \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? BGE_ERR_OOM : BGE_ERR_HIP, #expr, e_); \
    } while (0)

int Broadphase::configure(uint64_t n_slots, uint64_t pair_capacity)
{
    if (n_slots <= n_slots_ && pair_capacity <= capacity_) {
        ran_ = false;
        return BGE_OK;
    }
    release();
    n_slots_ = n_slots;
    capacity_ = pair_capacity;
    // table: at least 2 cells per possible body, multiple of the scan block
    uint64_t t = std::max<uint64_t>(2 * n_slots, 4096);
    t = (t + kScanBlock - 1) / kScanBlock * kScanBlock;
    if (t > 0x7ffff000ull) t = 0x7ffff000ull / kScanBlock * kScanBlock;
    table_size_ = static_cast<uint32_t>(t);
    // Allocation is lazy-free: a world that never runs the broadphase still pays for these buffers;
    // they total ~ (8 + 8 + 32 + 8 + 4) B per slot + 8 B per pair.
    BP_TRY(hipMalloc(&pairs_, std::max<uint64_t>(capacity_, 1) * 8));
    BP_TRY(hipMalloc(&counters_, sizeof(Accum)));
    BP_TRY(hipMalloc(&cell_count_, (static_cast<size_t>(table_size_) + kScanBlock) * 4));
    BP_TRY(hipMalloc(&cell_start_, (static_cast<size_t>(table_size_) + kScanBlock) * 4));
    BP_TRY(hipMalloc(&scan_tmp_, (static_cast<size_t>(table_size_) / kScanBlock + 2) * 4));
    BP_TRY(hipMalloc(&sorted_slot_, std::max<uint64_t>(n_slots, 1) * 4));  // body rank inside its cell
    BP_TRY(hipMalloc(&sorted_aabb_, std::max<uint64_t>(n_slots, 1) * 32));
    BP_TRY(hipMalloc(&body_cell_, std::max<uint64_t>(n_slots, 1) * 4));
    BP_TRY(hipMalloc(&large_list_, std::max<uint64_t>(n_slots, 1) * 4));
    BP_TRY(hipMemset(counters_, 0, sizeof(Accum)));
    return BGE_OK;
}

int Broadphase::run(hipStream_t stream, const WorldView& w, uint64_t n, const uint32_t* entity_of_slot)
{
    if (n > n_slots_) {
        error_ = "broadphase not configured for this many slots";
        return BGE_ERR_STATE;
    }
    Accum* acc = static_cast<Accum*>(counters_);
    uint32_t* cell_count = static_cast<uint32_t*>(cell_count_);
    uint32_t* cell_start = static_cast<uint32_t*>(cell_start_);
    uint32_t* block_sums = static_cast<uint32_t*>(scan_tmp_);
    uint32_t* body_rank = static_cast<uint32_t*>(sorted_slot_);
    uint32_t* body_cell = static_cast<uint32_t*>(body_cell_);
    uint32_t* large_list = static_cast<uint32_t*>(large_list_);
    float4* sorted = static_cast<float4*>(sorted_aabb_);
    const PairSink sink{&acc->n_pairs, static_cast<uint2*>(pairs_), capacity_};
    ran_ = true;
    if (n == 0) {
        BP_TRY(hipMemsetAsync(counters_, 0, sizeof(Accum), stream));
        return BGE_OK;
    }
    // the scan covers table_size_ + 2 entries (cell_start[c + 2] is read for the last cell)
    const uint32_t scan_n = table_size_ + 2;
    const uint32_t scan_blocks = blocks_for(scan_n, kScanBlock);
    const uint32_t slot_blocks = blocks_for(n, 256);

    hipLaunchKernelGGL(k_bp_reset, dim3(1), dim3(256), 0, stream, acc);
    hipLaunchKernelGGL(k_bp_bounds, dim3(std::min<uint32_t>(slot_blocks, 1024)), dim3(256), 0, stream, n, w.flags, w.aabb, acc);
    hipLaunchKernelGGL(k_bp_params, dim3(1), dim3(64), 0, stream, acc, table_size_);
    BP_TRY(hipMemsetAsync(cell_count, 0, (static_cast<size_t>(table_size_) + kScanBlock) * 4, stream));
    hipLaunchKernelGGL(k_bp_count, dim3(slot_blocks), dim3(256), 0, stream, n, w.flags, w.aabb, acc, cell_count, body_cell,
                       body_rank, large_list);
    hipLaunchKernelGGL(k_scan_blocks, dim3(scan_blocks), dim3(256), 0, stream, cell_count, cell_start, block_sums, scan_n);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, stream, block_sums, scan_blocks);
    hipLaunchKernelGGL(k_scan_add, dim3(scan_blocks), dim3(256), 0, stream, cell_start, block_sums, scan_n);
    hipLaunchKernelGGL(k_bp_scatter, dim3(slot_blocks), dim3(256), 0, stream, n, w.flags, w.aabb, cell_start, body_cell,
                       body_rank, sorted);
    hipLaunchKernelGGL(k_bp_pairs, dim3(std::min<uint32_t>(slot_blocks, 2048)), dim3(256), 0, stream, acc, cell_start, sorted, w.flags, w.group, w.mask,
                       entity_of_slot, sink, static_cast<uint32_t>(n));
    hipLaunchKernelGGL(k_bp_large, dim3(std::min<uint32_t>(slot_blocks, 4096)), dim3(256), 0, stream, n, acc, large_list,
                       body_cell, w.aabb, w.flags, w.group, w.mask, entity_of_slot, sink);
    BP_TRY(hipGetLastError());
    return BGE_OK;
}

int Broadphase::download(hipStream_t stream, uint32_t* pairs2, uint64_t cap, uint64_t* total)
{
    *total = 0;
    if (!ran_) return BGE_OK;
    unsigned long long n = 0;
    const Accum* acc = static_cast<const Accum*>(counters_);
    BP_TRY(hipMemcpyAsync(&n, &acc->n_pairs, sizeof n, hipMemcpyDeviceToHost, stream));
    BP_TRY(hipStreamSynchronize(stream));
    *total = n;
    const uint64_t take = std::min<uint64_t>(std::min<uint64_t>(n, cap), capacity_);
    if (take && pairs2) {
        BP_TRY(hipMemcpyAsync(pairs2, pairs_, take * 8, hipMemcpyDeviceToHost, stream));
        BP_TRY(hipStreamSynchronize(stream));
    }
    return BGE_OK;
}

} // namespace bge
