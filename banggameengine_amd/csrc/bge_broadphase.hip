// bge_broadphase.hip — AABB broadphase on gfx950: uniform grid + counting sort + wave-compacted pair list.
//
// Replaces what the reference gets from Bullet's btDbvtBroadphase (src/physics/PhysicsSystem.cpp:124;
// updateAabbs / calculateOverlappingPairs inside stepSimulation, :863).  The pair SET it produces is the
// history-free core of Bullet's pair cache (see oracle/broadphase_ref.h for the exact specification):
//   AABBs overlap (non-strict) AND (groupA & maskB) && (groupB & maskA) AND not both Static.
//
// Per tick, all on the world's stream, no host round trip (DESIGN.md 4.3 has the measurements behind every choice):
//   1. k_bp_bounds      scene bounds (wave64 __shfl reductions -> per-workgroup partials) + per extent bin, in LDS, the
//                       number of bodies and the widest one (flushed to 16 copies of the global arrays)
//   2. k_bp_params      one workgroup derives the grid: the bin edge that minimises the expected number of AABB tests, then
//                       cell = the widest body below that edge (grown until the padded grid fits the table); bodies wider
//                       than a cell are "large" (step 7).  It also re-arms the accumulators (there is no reset kernel)
//   3. k_sort_hist      two-level counting sort by cell, level 1: per chunk workgroup an LDS histogram over coarse buckets
//                       of 4096 cells -> bucket x workgroup matrix;  k_scan_lookback: single-pass scan of the matrix
//   4. k_sort_coarse_t  every record to its bucket: 8192 records per workgroup held in registers, ranked in LDS, written out
//                       through an LDS window in contiguous runs
//   5. k_sort_fine_t    one workgroup per bucket: LDS histogram over its cells -> cell_start, records in cell order, read once
//                       and written as one stream (two-sweep path for buckets beyond one register pass)
//                       [tables beyond 32 M cells, BGE_BP_SORT=atomic: k_bp_count / scan / k_bp_scatter with global atomics]
//   6. k_bp_pairs_wave  one wave owns 64 consecutive sorted bodies; for each of the 5 contiguous runs that cover the 14
//                       "forward" neighbour cells the candidates are staged in wave-private LDS; hits are ballot-compacted
//                       into an LDS staging buffer and written 128 at a time behind one atomic on one of 64 counters
//   7. k_bp_large       large bodies against everything
//   8. k_bp_compact     (on demand) the 64 shard slices of the pair list -> one compact list + totals
// Sorted records are 32 bytes (min xyz | entity, max xyz | filter class) or, for the slab search and more than 255 filter
// classes, 48 bytes (... | cell, group | mask | static | slot).
// A small body's AABB is narrower than one cell (by a 2^-20 margin that dominates the f64 rounding of the cell
// coordinates), so two overlapping small bodies sit in cells that differ by at most one per axis; scanning only forward
// neighbours (and, inside the own cell, only later records) reports each pair once.
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
constexpr uint32_t kShards = 64;
constexpr uint32_t kBoundsBlocks = 2048; // grid of k_bp_bounds: 8 workgroups per CU keep enough loads in flight
constexpr uint32_t kExtentBins = 1024;
constexpr uint32_t kHistShards = 16;
constexpr uint32_t kParamsThreads = 1024; // k_bp_params: 16 waves gather, one decides
__host__ __device__ inline uint32_t extent_bin(float e) { return (__builtin_bit_cast(uint32_t, e) >> 21) & (kExtentBins - 1u); }
__host__ __device__ inline float extent_bin_upper(uint32_t b) { return __builtin_bit_cast(float, (b + 1u) << 21); }

struct Accum {
    uint32_t min_bits[3]; // ordered-uint encoding of floats
    uint32_t max_bits[3];
    uint32_t n_bodies;
    uint32_t n_large;
    unsigned long long n_pairs;        // total found (written by k_bp_compact)
    unsigned long long n_pairs_kept;   // total present in the compact list
    GridParams grid;
    // Pair emission is sharded: one counter (on its own 64-byte line) and one slice of the staging buffer per shard.
    // A single counter word sustains ~10^8 atomics/s; 57 k flushes on one word cost ~0.6 ms for 12.6 M pairs.
    unsigned long long shard_count[kShards][8];
    // per-workgroup partial bounds of k_bp_bounds (no atomics), reduced by k_bp_params
    float part_min[kBoundsBlocks][3];
    float part_max[kBoundsBlocks][3];
    uint32_t part_count[kBoundsBlocks];
    // bodies per extent bin (bin = float exponent + 2 mantissa bits), in kHistShards copies: the bodies of a scene share a
    // few sizes, so every workgroup of k_bp_bounds ends with an atomic on the SAME bin — 2048 of them on one word cost
    // ~20 us of the kernel's 41 (one word sustains ~10^8 atomics/s); k_bp_params adds the copies up
    uint32_t extent_hist[kHistShards][kExtentBins];
    // widest extent seen in each bin (float bits; extents are positive), same copies: the cell size is the widest SMALL
    // body, not the upper edge of its bin — edges are 19-25 % apart and the number of AABB tests grows with the cube of the cell
    uint32_t extent_max[kHistShards][kExtentBins];
    // "large" weight of every bin: bodies (k_bp_bounds: each body counts itself) or waves (k_bp_reduce_partials: the tick
    // kernel reports one widest extent per wave, so a bin above the chosen edge stands for at least that many large bodies)
    uint32_t extent_large[kHistShards][kExtentBins];
    uint32_t scan_ticket;              // tile tickets of k_scan_lookback (dispatch order)
    uint32_t scan_error;               // a look-back gave up (never observed; keeps a logic error from hanging the GPU)
    uint32_t reduce_ticket;            // workgroups of k_bp_reduce_partials that are done; the last one decides the grid
};

constexpr uint32_t kLargeCell = 0xffffffffu;

__device__ __forceinline__ uint32_t f2ord(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ bool is_body(uint32_t f) { return (f & kTypeMask) != 0; } // (type bits exist only on slots that carry a body — with a Transform, or orphaned)

__global__ void __launch_bounds__(256) k_bp_bounds(uint64_t n_slots, const uint32_t* __restrict__ flags,
                                                   const float* __restrict__ aabb, Accum* acc)
{
    __shared__ uint32_t hist[kExtentBins];
    __shared__ uint32_t hmax[kExtentBins];
    __shared__ float red[4][6];
    __shared__ uint32_t red_cnt[4];
    for (uint32_t k = threadIdx.x; k < kExtentBins; k += blockDim.x) {
        hist[k] = 0;
        hmax[k] = 0;
    }
    __syncthreads();

    float mn[3] = {INFINITY, INFINITY, INFINITY};
    float mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t cnt = 0;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    // uniform trip count per wave (the aggregation below ballots across all 64 lanes)
    const uint64_t first = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    const uint64_t rounds = (n_slots + stride - 1) / stride;
    // kBoundsUnroll rounds at a time: all their loads are issued before the first histogram update, whose ballots and LDS
    // atomics the compiler will not move loads across (measured at 4 M bodies: 44 us one round at a time)
    constexpr int kBoundsUnroll = 4;
    for (uint64_t r0 = 0; r0 < rounds; r0 += kBoundsUnroll) {
        uint32_t fl[kBoundsUnroll];
        float2 b0[kBoundsUnroll], b1[kBoundsUnroll], b2[kBoundsUnroll];
        bool live[kBoundsUnroll];
#pragma unroll
        for (int u = 0; u < kBoundsUnroll; ++u) {
            const uint64_t s = first + (r0 + u) * stride;
            live[u] = r0 + u < rounds && s < n_slots;
            // flags and AABB are loaded together (the AABB of a slot without a body is just not used): one memory round
            // trip per body instead of two dependent ones
            const uint64_t sc = live[u] ? s : 0;
            fl[u] = flags[sc];
            const float2* b = reinterpret_cast<const float2*>(aabb + 6 * sc);
            b0[u] = b[0]; // min.x min.y
            b1[u] = b[1]; // min.z max.x
            b2[u] = b[2]; // max.y max.z
        }
#pragma unroll
        for (int u = 0; u < kBoundsUnroll; ++u) {
            uint32_t my_bin = 0;
            bool have_bin = false;
            if (live[u] && is_body(fl[u])) {
                mn[0] = fminf(mn[0], b0[u].x);
                mn[1] = fminf(mn[1], b0[u].y);
                mn[2] = fminf(mn[2], b1[u].x);
                mx[0] = fmaxf(mx[0], b1[u].y);
                mx[1] = fmaxf(mx[1], b2[u].x);
                mx[2] = fmaxf(mx[2], b2[u].y);
                const float e = fmaxf(fmaxf(b1[u].y - b0[u].x, b2[u].x - b0[u].y), b2[u].y - b1[u].x);
                my_bin = (e > 0.0f && e < INFINITY) ? extent_bin(e) : 0u;
                if (e > 0.0f && e < INFINITY) atomicMax(&hmax[my_bin], __float_as_uint(e)); // (lanes of a bin serialise: ~64 cycles)
                have_bin = true;
                cnt += 1;
            }
            // Histogram update, wave-aggregated: bodies of similar size share a bin, and 64 lanes hammering one LDS
            // word serialise.  Each round the first pending lane's bin is broadcast, all lanes with that bin retire
            // together and their leader adds the population count (usually one or two rounds).
            while (true) {
                const unsigned long long pending = __ballot(have_bin);
                if (pending == 0) break;
                const int leader = __ffsll(static_cast<long long>(pending)) - 1;
                const uint32_t lead_bin = __shfl(my_bin, leader, 64);
                const unsigned long long same = __ballot(have_bin && my_bin == lead_bin);
                if (static_cast<int>(threadIdx.x & 63u) == leader) atomicAdd(&hist[lead_bin], static_cast<uint32_t>(__popcll(same)));
                if (have_bin && my_bin == lead_bin) have_bin = false;
            }
        }
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
        for (int a = 0; a < 3; ++a) {
            acc->part_min[blockIdx.x][a] = fminf(fminf(red[0][a], red[1][a]), fminf(red[2][a], red[3][a]));
            acc->part_max[blockIdx.x][a] = fmaxf(fmaxf(red[0][3 + a], red[1][3 + a]), fmaxf(red[2][3 + a], red[3][3 + a]));
        }
        acc->part_count[blockIdx.x] = total;
    }
    for (uint32_t k = threadIdx.x; k < kExtentBins; k += blockDim.x) {
        const uint32_t h = hist[k];
        if (h) {
            atomicAdd(&acc->extent_hist[blockIdx.x % kHistShards][k], h);
            atomicAdd(&acc->extent_large[blockIdx.x % kHistShards][k], h);
            if (hmax[k]) atomicMax(&acc->extent_max[blockIdx.x % kHistShards][k], hmax[k]);
        }
    }
}

// The same accumulators from the per-wave partials the tick kernel wrote beside the AABBs (TickParams::bp_partial: bounds,
// body count and widest extent of each wave's 64 slots) — 32 bytes per wave instead of a pass over 28 bytes per body.  The
// extent histogram is then a histogram of WAVE maxima: bin b holds the bodies of the waves whose widest body falls into b
// (they are all at most that wide) and, as its "large" weight, the number of such waves.  Good enough to choose the cell
// size — the choice only affects speed: whatever the cell, a body wider than it is treated as large by the sort itself.
// With `decide` the workgroup that finishes last goes on to choose the grid (decide_grid below) — what a separate
// k_bp_params launch did: 5.2 + 15.6 us as two kernels at 4 M bodies (profiles/r02/cube4m_kernel_stats.csv).
constexpr uint32_t kReduceBlocks = 128;
template <uint32_t NS> __device__ void decide_grid(Accum* acc, uint32_t max_cells, uint32_t n_bounds_blocks);
constexpr uint32_t kReduceShards = 4; // histogram copies the 128 workgroups of k_bp_reduce_partials spread their atomics over
__global__ void __launch_bounds__(256) k_bp_reduce_partials(const float4* __restrict__ partials, uint32_t n_partials, Accum* acc,
                                                            uint32_t max_cells, uint32_t decide)
{
    __shared__ uint32_t hist[kExtentBins];
    __shared__ uint32_t hwaves[kExtentBins];
    __shared__ uint32_t hmax[kExtentBins];
    __shared__ float red[4][6];
    __shared__ uint32_t red_cnt[4];
    for (uint32_t k = threadIdx.x; k < kExtentBins; k += blockDim.x) {
        hist[k] = 0;
        hwaves[k] = 0;
        hmax[k] = 0;
    }
    __syncthreads();
    float mn[3] = {INFINITY, INFINITY, INFINITY};
    float mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t cnt = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounds = (n_partials + stride - 1u) / stride; // uniform trip count: the aggregation ballots across the wave
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x + r * stride;
        uint32_t my_bin = 0, my_count = 0, my_ext = 0;
        bool have_bin = false;
        if (q < n_partials) {
            const float4 a = partials[2ull * q], b = partials[2ull * q + 1ull];
            my_count = __float_as_uint(b.w);
            if (my_count) {
                mn[0] = fminf(mn[0], a.x);
                mn[1] = fminf(mn[1], a.y);
                mn[2] = fminf(mn[2], a.z);
                mx[0] = fmaxf(mx[0], b.x);
                mx[1] = fmaxf(mx[1], b.y);
                mx[2] = fmaxf(mx[2], b.z);
                cnt += my_count;
                const float e = a.w;
                const bool ok = e > 0.0f && e < INFINITY;
                my_bin = ok ? extent_bin(e) : 0u;
                my_ext = ok ? __float_as_uint(e) : 0u;
                have_bin = true;
            }
        }
        // wave-aggregated histogram update (the waves of a scene share a few bins), as in k_bp_bounds
        while (true) {
            const unsigned long long pending = __ballot(have_bin);
            if (pending == 0) break;
            const int leader = __ffsll(static_cast<long long>(pending)) - 1;
            const uint32_t lead_bin = __shfl(my_bin, leader, 64);
            const bool mine = have_bin && my_bin == lead_bin;
            uint32_t c = mine ? my_count : 0u, e = mine ? my_ext : 0u;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                c += __shfl_xor(c, off, 64);
                e = max(e, static_cast<uint32_t>(__shfl_xor(e, off, 64)));
            }
            const unsigned long long same = __ballot(mine);
            if (static_cast<int>(threadIdx.x & 63u) == leader) {
                atomicAdd(&hist[lead_bin], c);
                atomicAdd(&hwaves[lead_bin], static_cast<uint32_t>(__popcll(same)));
                atomicMax(&hmax[lead_bin], e);
            }
            if (mine) have_bin = false;
        }
    }
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
        for (int a = 0; a < 3; ++a) {
            acc->part_min[blockIdx.x][a] = fminf(fminf(red[0][a], red[1][a]), fminf(red[2][a], red[3][a]));
            acc->part_max[blockIdx.x][a] = fmaxf(fmaxf(red[0][3 + a], red[1][3 + a]), fmaxf(red[2][3 + a], red[3][3 + a]));
        }
        acc->part_count[blockIdx.x] = total;
    }
    for (uint32_t k = threadIdx.x; k < kExtentBins; k += blockDim.x) {
        const uint32_t h = hist[k];
        if (h) {
            atomicAdd(&acc->extent_hist[blockIdx.x % kReduceShards][k], h);
            atomicAdd(&acc->extent_large[blockIdx.x % kReduceShards][k], hwaves[k]);
            if (hmax[k]) atomicMax(&acc->extent_max[blockIdx.x % kReduceShards][k], hmax[k]);
        }
    }
    if (!decide) return;
    // Every thread's stores and atomics are made visible at device scope, then one ticket per workgroup; whoever draws the
    // last one sees all the others' results (acquire: this CU's vector L1 is dropped) and decides.  Every workgroup reaches
    // this point, so the ticket word is back at 0 when the kernel ends.
    __shared__ uint32_t last_one;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(&acc->reduce_ticket, 1u);
        last_one = t == gridDim.x - 1u;
        if (last_one) acc->reduce_ticket = 0;
    }
    __syncthreads();
    if (!last_one) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    decide_grid<kReduceShards>(acc, max_cells, gridDim.x);
}

// LDS traffic between lanes of ONE wave: the hardware keeps a wave's DS operations in order; this keeps the
// compiler from moving them across the hand-over point.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One wave: pick the cell size that minimises the expected number of AABB tests,
//   tests(c) = n_small(c)^2 / volume * c^3 * 13.5   (13 forward cells + half the own cell)
//            + n_large(c) * n                        (k_bp_large: every large body against everything)
// over the histogram's bin edges c (a body is "small" when its widest side is < c); then grow the cell
// until the padded grid fits the table.
// NS: how many of the kHistShards copies of the extent histogram the producer filled (the others are all zero).
template <uint32_t NS> __device__ void decide_grid(Accum* acc, uint32_t max_cells, uint32_t n_bounds_blocks)
{
    constexpr uint32_t kWaves = kParamsThreads / 64; // at most: the caller's workgroup has blockDim.x / 64 of them
    __shared__ uint32_t below[kExtentBins + 1]; // exclusive prefix: bodies in bins < b
    __shared__ uint32_t lbelow[kExtentBins + 1]; // the same prefix over the bins' "large" weights (bodies, or waves)
    __shared__ uint32_t binmax[kExtentBins];    // widest extent in the bin (float bits)
    __shared__ float red[kWaves][6];
    __shared__ uint32_t red_cnt[kWaves];
    const uint32_t lane = threadIdx.x & 63u;
    {
        // Phase 1, all 16 waves: reduce the per-workgroup partial bounds and add up the histogram's copies.  Both are chains
        // of dependent-latency loads (~1.5 us a round trip): one wave alone spent 15 us here, 1024 threads need two rounds
        // for the partials and one for the histogram (every thread one bin, its 16 copies in flight together).
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        uint32_t cnt = 0;
        for (uint32_t b = threadIdx.x; b < n_bounds_blocks; b += blockDim.x) {
            for (int a = 0; a < 3; ++a) {
                mn[a] = fminf(mn[a], acc->part_min[b][a]);
                mx[a] = fmaxf(mx[a], acc->part_max[b][a]);
            }
            cnt += acc->part_count[b];
        }
        // All of a thread's histogram words are loaded before the first of them is cleared: interleaved, every load waited
        // behind the store in front of it and the 48 round trips came one after the other (20 us of k_bp_params' 21).
        constexpr uint32_t G = kHistShards / NS; // bins per thread and sweep: 48 loads in flight either way
        for (uint32_t b0 = threadIdx.x; b0 < kExtentBins; b0 += blockDim.x * G) {
            uint32_t hv[G][NS], lv[G][NS], mv[G][NS];
#pragma unroll
            for (uint32_t g = 0; g < G; ++g) {
                const uint32_t b = min(b0 + g * blockDim.x, kExtentBins - 1u);
#pragma unroll
                for (uint32_t c = 0; c < NS; ++c) {
                    hv[g][c] = acc->extent_hist[c][b];
                    lv[g][c] = acc->extent_large[c][b];
                    mv[g][c] = acc->extent_max[c][b];
                }
            }
#pragma unroll
            for (uint32_t g = 0; g < G; ++g) {
                const uint32_t b = b0 + g * blockDim.x;
                if (b >= kExtentBins) continue;
                uint32_t t = 0, m = 0, lw = 0;
#pragma unroll
                for (uint32_t c = 0; c < NS; ++c) {
                    t += hv[g][c];
                    lw += lv[g][c];
                    m = max(m, mv[g][c]);
                    acc->extent_hist[c][b] = 0; // consumed: ready for the next run (there is no reset kernel)
                    acc->extent_large[c][b] = 0;
                    acc->extent_max[c][b] = 0;
                }
                below[b] = t; // the totals, for the moment
                lbelow[b] = lw;
                binmax[b] = m;
            }
        }
        // the counters the REST of this run accumulates into (every kernel that touches them is launched after this one)
        if (threadIdx.x < kShards) acc->shard_count[threadIdx.x][0] = 0;
        if (threadIdx.x == kShards) {
            acc->n_large = 0;
            acc->n_pairs = 0;
            acc->n_pairs_kept = 0;
            acc->scan_ticket = 0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            for (int a = 0; a < 3; ++a) {
                mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
                mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
            }
            cnt += __shfl_xor(cnt, off, 64);
        }
        if (lane == 0) {
            for (int a = 0; a < 3; ++a) {
                red[threadIdx.x >> 6][a] = mn[a];
                red[threadIdx.x >> 6][3 + a] = mx[a];
            }
            red_cnt[threadIdx.x >> 6] = cnt;
        }
        __syncthreads();
        if (threadIdx.x >= 64) return; // one wave from here on: the barriers below are wave-level
    }
    float gmn[3], gmx[3]; // every lane of the remaining wave holds the scene bounds and the body count
    uint32_t n = 0;
    for (int a = 0; a < 3; ++a) {
        gmn[a] = INFINITY;
        gmx[a] = -INFINITY;
    }
    for (uint32_t wv = 0; wv < blockDim.x / 64; ++wv) {
        for (int a = 0; a < 3; ++a) {
            gmn[a] = fminf(gmn[a], red[wv][a]);
            gmx[a] = fmaxf(gmx[a], red[wv][3 + a]);
        }
        n += red_cnt[wv];
    }
    if (lane == 0) {
        for (int a = 0; a < 3; ++a) {
            acc->min_bits[a] = f2ord(gmn[a]);
            acc->max_bits[a] = f2ord(gmx[a]);
        }
        acc->n_bodies = n;
    }
    {
        // exclusive prefix of the histogram: 16 consecutive bins per lane + a wave64 __shfl_up scan of the lane totals
        for (int which = 0; which < 2; ++which) {
            uint32_t* arr = which ? lbelow : below;
            uint32_t h[16];
            uint32_t sum = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                h[k] = arr[lane * 16 + k];
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
                arr[lane * 16 + k] = run;
                run += h[k];
            }
            if (lane == 63) arr[kExtentBins] = run;
        }
    }
    wave_sync();
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    double volume = 1.0;
    if (n) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = gmn[a];
            hi[a] = gmx[a];
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
        // bodies that would be large: every body of a bin above the edge (k_bp_bounds), or at least one per wave whose
        // widest body lies above it (k_bp_reduce_partials)
        const double large = static_cast<double>(lbelow[kExtentBins] - lbelow[b + 1]);
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
    // the widest body below the chosen edge: it, not the edge, sets the cell size
    best_bin = __shfl(best_bin, 0, 64);
    uint32_t widest = 0;
    for (uint32_t b = lane; b <= best_bin && b < kExtentBins; b += 64) widest = max(widest, binmax[b]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) widest = max(widest, static_cast<uint32_t>(__shfl_xor(widest, off, 64)));
    if (lane != 0) return;

    GridParams g;
    g.n_bodies = n;
    // small_limit = cell * (1 - 2^-20) must not fall below the widest small body: cell = widest * (1 + 2^-19)
    float cell = n ? (widest ? __uint_as_float(widest) * 1.0000019073486328f : extent_bin_upper(best_bin)) : 1.0f;
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

// decide_grid as a launch of its own: after k_bp_bounds, and after k_bp_reduce_partials under BGE_BP_PARAMS=split
__global__ void __launch_bounds__(kParamsThreads) k_bp_params(Accum* acc, uint32_t max_cells, uint32_t n_bounds_blocks)
{
    decide_grid<kHistShards>(acc, max_cells, n_bounds_blocks);
}

__device__ __forceinline__ uint32_t cell_axis(const GridParams& g, float v, int a)
{
    // +1: the padding cell; clamped so that garbage (NaN / out of range) stays inside the table
    double t = floor((static_cast<double>(v) - static_cast<double>(g.origin[a])) * g.inv_cell);
    t = fmin(fmax(t, 0.0), 4.0e9);
    return static_cast<uint32_t>(t) + 1u;
}
__device__ __forceinline__ uint32_t cell_of(const GridParams& g, float x, float y, float z)
{
    const uint32_t dy = g.dim_xy / g.dim_x;
    const uint32_t dz = g.n_cells / g.dim_xy;
    const uint32_t cx = min(cell_axis(g, x, 0), g.dim_x - 2u);
    const uint32_t cy = min(cell_axis(g, y, 1), dy - 2u);
    const uint32_t cz = min(cell_axis(g, z, 2), dz - 2u);
    return cx + g.dim_x * cy + g.dim_xy * cz;
}
__device__ __forceinline__ uint32_t cell_of(const GridParams& g, const float* mn) { return cell_of(g, mn[0], mn[1], mn[2]); }

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

// Single-pass scan: one read and one write of the table instead of the three kernels below (76 us -> see DESIGN.md 4.3).
// The table has at most ~1000 tiles of 8192 cells, so there is no look-back CHAIN: every tile publishes its own sum and
// then adds up the sums of ALL earlier tiles itself (256 threads x <= 4 status words, one round trip), O(tiles^2 / 2)
// 8-byte reads in total — 4 MB for 1024 tiles.
//   * a workgroup takes its tile number from a ticket counter, so a tile only ever waits for tiles whose workgroups
//     were dispatched before it (no deadlock whatever the dispatch order or residency);
//   * per tile ONE 64-bit status word (epoch << 32 | sum), written and read with agent-scope atomics only: coherent
//     across the 8 XCDs' L2s, and the payload travels in the flag word itself — no fence, nothing that could be stale;
//   * the epoch (one per run) makes words of earlier runs read as "not there yet": the status array is never cleared;
//   * tiles entirely beyond `limit` (= cells in use + 2, known only on the device) publish a zero and touch no memory.
constexpr uint32_t kScanTile = 8192; // 256 threads x 32
__global__ void __launch_bounds__(256) k_scan_lookback(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                       unsigned long long* __restrict__ status, Accum* acc, uint32_t n,
                                                       uint32_t epoch, uint32_t cells_limit)
{
    __shared__ uint32_t wave_tot[4], wave_back[4];
    __shared__ uint32_t s_tile;
    if (threadIdx.x == 0) s_tile = atomicAdd(&acc->scan_ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t limit = cells_limit ? min(n, acc->grid.n_cells + 2u) : n; // cells_limit: scan of the cell table
    const bool live = tile * kScanTile < limit; // the buffers are padded to whole tiles and zero beyond the cells in use
    const uint32_t base = tile * kScanTile + threadIdx.x * 32;
    uint4 v[8];
    uint32_t sum = 0;
    if (live) {
        const uint4* src = reinterpret_cast<const uint4*>(in + base);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = src[k];
            if (!cells_limit) { // generic input: nothing is promised about the padding of the last tile
                const uint32_t e = base + 4u * k;
                if (e >= limit) v[k].x = 0;
                if (e + 1u >= limit) v[k].y = 0;
                if (e + 2u >= limit) v[k].z = 0;
                if (e + 3u >= limit) v[k].w = 0;
            }
            sum += v[k].x + v[k].y + v[k].z + v[k].w;
        }
    }
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if ((threadIdx.x & 63u) >= static_cast<uint32_t>(off)) incl += t;
    }
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (lane == 63u) wave_tot[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __hip_atomic_store(&status[tile], (static_cast<unsigned long long>(epoch) << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // sum of all earlier tiles
    uint32_t back = 0;
    for (uint32_t j = threadIdx.x; j < tile; j += 256u) {
        unsigned long long w = 0;
        uint32_t spins = 0;
        while (true) {
            w = __hip_atomic_load(&status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((w >> 32) == epoch) break;
            if (++spins > (1u << 24)) { // ~seconds: a logic error must not hang the GPU
                acc->scan_error = 1u;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        back += static_cast<uint32_t>(w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) back += __shfl_xor(back, off, 64);
    if (lane == 0) wave_back[wave] = back;
    __syncthreads();
    if (!live) return;
    uint32_t run = wave_back[0] + wave_back[1] + wave_back[2] + wave_back[3];
    for (uint32_t k = 0; k < wave; ++k) run += wave_tot[k];
    run += incl - sum;
    uint4* dst = reinterpret_cast<uint4*>(out + base);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint4 o;
        o.x = run;
        o.y = o.x + v[k].x;
        o.z = o.y + v[k].y;
        o.w = o.z + v[k].z;
        run = o.w + v[k].w;
        dst[k] = o;
    }
}

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

// ---- Counting sort by cell WITHOUT global atomics: two levels, all counting in LDS -------------------------------
// k_bp_count + scan + k_bp_scatter issue one returning global atomic and one 32/48-byte scattered write per body into
// tables far larger than the L2s; random device-scope atomics run at ~25-30 G/s chip-wide on MI355X (4 M bodies:
// 160 us for the atomics alone, measured), whatever the occupancy.  The LDS sort replaces them:
//   A0 k_sort_hist    a workgroup owns one contiguous chunk of slots; LDS histogram of COARSE buckets
//                     (bucket = cell >> shift, 4096 or 8192 cells); one row of the (bucket x workgroup) count matrix
//   -- exclusive scan of the matrix in bucket-major order (k_scan_lookback) = where each workgroup's share of each
//      bucket starts
//   A1 k_sort_coarse  same chunks, same order: every body's record goes to its bucket at an LDS-atomic cursor
//   B  k_sort_fine    one workgroup per bucket: LDS histogram over the bucket's cells, LDS scan -> cell_start for those
//                     cells, second sweep (L2 hits) places the records in cell order
// Every global access is coalesced or confined to one bucket's few hundred KB; the cell table is written once
// (no memset, no 8 M-entry scan).  A bucket that holds most of the scene (everything in one spot) is still handled,
// by one workgroup sweeping it.
constexpr uint32_t kSortThreads = 1024;
constexpr uint32_t kSortMaxBuckets = 4100; // (table >> shift) + 2
#ifndef BGE_SORT_GROUPS
#define BGE_SORT_GROUPS 512
#endif
#ifndef BGE_FINE_THREADS
#define BGE_FINE_THREADS 512 /* measured at 4 M bodies: 256 -> 0.747 ms per tick, 512 -> 0.727, 1024 -> 0.726 */
#endif
constexpr uint32_t kSortGroups = BGE_SORT_GROUPS; // chunk workgroups of passes A0 / A1 (2 per CU)
constexpr uint32_t kFineThreads = BGE_FINE_THREADS;

__device__ __forceinline__ bool body_is_large(const GridParams& g, const float* b)
{
    const float ext = fmaxf(fmaxf(b[3] - b[0], b[4] - b[1]), b[5] - b[2]);
    return !(ext <= g.small_limit);
}

__global__ void __launch_bounds__(kSortThreads) k_sort_hist(uint64_t n_slots, uint64_t chunk, const uint32_t* __restrict__ flags,
                                                            const float* __restrict__ aabb, Accum* acc, uint32_t shift,
                                                            uint32_t n_buckets, uint32_t* __restrict__ matrix,
                                                            uint32_t* __restrict__ large_list)
{
    __shared__ uint32_t hist[kSortMaxBuckets];
    for (uint32_t k = threadIdx.x; k < n_buckets; k += kSortThreads) hist[k] = 0;
    __syncthreads();
    const GridParams g = acc->grid;
    const uint64_t begin = blockIdx.x * chunk, end = min(begin + chunk, n_slots);
#pragma unroll 2
    for (uint64_t s = begin + threadIdx.x; s < end; s += kSortThreads) {
        // (as written the compiler sinks the AABB loads below the is_body test — flags, wait, AABB, wait.  Pinning all loads of
        //  1 / 2 / 4 slots in front of the first use, as k_sort_coarse_t and the pair search now do, was measured here at 4 M
        //  bodies: 29.3 / 32.1 / 32.8 us against 29.4 — this pass does not wait on its loads)
        const uint32_t fl = flags[s];
        const float2* bp = reinterpret_cast<const float2*>(aabb + 6 * s);
        const float2 b0 = bp[0], b1 = bp[1], b2 = bp[2];
        if (!is_body(fl)) continue;
        const float b[6] = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y};
        if (body_is_large(g, b)) {
            large_list[atomicAdd(&acc->n_large, 1u)] = static_cast<uint32_t>(s);
            continue;
        }
        atomicAdd(&hist[cell_of(g, b[0], b[1], b[2]) >> shift], 1u);
    }
    __syncthreads();
    // bucket-major matrix: entry (bucket, workgroup)
    for (uint32_t k = threadIdx.x; k < n_buckets; k += kSortThreads) matrix[static_cast<uint64_t>(k) * gridDim.x + blockIdx.x] = hist[k];
}

template <bool COMPACT>
__global__ void __launch_bounds__(kSortThreads) k_sort_coarse(uint64_t n_slots, uint64_t chunk, const uint32_t* __restrict__ flags,
                                                              const float* __restrict__ aabb, const Accum* __restrict__ acc,
                                                              uint32_t shift, uint32_t n_buckets, const uint32_t* __restrict__ offsets,
                                                              const uint32_t* __restrict__ group, const uint32_t* __restrict__ mask,
                                                              const uint32_t* __restrict__ class_of_slot,
                                                              const uint32_t* __restrict__ entity_of_slot, float4* __restrict__ coarse)
{
    __shared__ uint32_t cursor[kSortMaxBuckets];
    for (uint32_t k = threadIdx.x; k < n_buckets; k += kSortThreads) cursor[k] = offsets[static_cast<uint64_t>(k) * gridDim.x + blockIdx.x];
    __syncthreads();
    const GridParams g = acc->grid;
    const uint64_t begin = blockIdx.x * chunk, end = min(begin + chunk, n_slots);
#pragma unroll 2
    for (uint64_t s = begin + threadIdx.x; s < end; s += kSortThreads) {
        const uint32_t f = flags[s];
        const float2* bp = reinterpret_cast<const float2*>(aabb + 6 * s);
        const float2 b0 = bp[0], b1 = bp[1], b2 = bp[2];
        const uint32_t ent = entity_of_slot[s];
        if (!is_body(f)) continue;
        const float b[6] = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y};
        if (body_is_large(g, b)) continue;
        const uint32_t c = cell_of(g, b[0], b[1], b[2]);
        const uint64_t pos = atomicAdd(&cursor[c >> shift], 1u);
        if (COMPACT) {
            coarse[2ull * pos] = make_float4(b[0], b[1], b[2], __uint_as_float(ent));
            coarse[2ull * pos + 1] = make_float4(b[3], b[4], b[5], __uint_as_float(class_of_slot[s]));
        } else {
            coarse[3ull * pos] = make_float4(b[0], b[1], b[2], __uint_as_float(ent));
            coarse[3ull * pos + 1] = make_float4(b[3], b[4], b[5], __uint_as_float(c));
            coarse[3ull * pos + 2] = make_float4(__uint_as_float(group[s]), __uint_as_float(mask[s]),
                                                 __uint_as_float((f & kTypeMask) == 1u ? 1u : 0u), __uint_as_float(static_cast<uint32_t>(s)));
        }
    }
}

// Pass A1, transposing variant (32-byte records): the same result as k_sort_coarse<true> — every record in its bucket, at a
// position inside the range the matrix scan reserved for this workgroup — but written out in BUCKET ORDER.  k_sort_coarse
// stores each record straight from the thread that loaded it: 4 M scattered 32-byte writes, which cost 65 of its ~110 us
// (measured against linear writes).  Here a workgroup keeps a pass of 8192 slots in REGISTERS (8 records per
// thread), counts and ranks them by bucket in LDS, passes them through an LDS window ordered by rank and writes the window
// out with consecutive lanes on consecutive float4s: a (workgroup, bucket) run of ~15 records becomes one ~480-byte write.
template <bool COMPACT>
struct SortT {
    static constexpr uint32_t RS = COMPACT ? 2u : 3u;          // float4 per record
    static constexpr uint32_t PT = COMPACT ? 8u : 4u;          // records a thread holds in registers
    static constexpr uint32_t kPass = PT * kSortThreads;       // records per pass of a workgroup
    static constexpr uint32_t kWindow = COMPACT ? 4096u : 2048u; // records in the coarse pass's LDS window (128 / 96 KiB)
};
template <bool COMPACT>
__global__ void __launch_bounds__(kSortThreads, 1)
    k_sort_coarse_t(uint64_t n_slots, uint64_t chunk, const uint32_t* __restrict__ flags, const float* __restrict__ aabb,
                    const Accum* __restrict__ acc, uint32_t shift, uint32_t n_buckets, const uint32_t* __restrict__ offsets,
                    const uint32_t* __restrict__ group, const uint32_t* __restrict__ mask,
                    const uint32_t* __restrict__ class_of_slot, const uint32_t* __restrict__ entity_of_slot,
                    float4* __restrict__ coarse)
{
    using T = SortT<COMPACT>;
    constexpr uint32_t RS = T::RS, PT = T::PT;
    extern __shared__ uint32_t lds_u32[];
    uint32_t* lcur = lds_u32;                       // [n_buckets] counts -> exclusive prefix -> running cursor
    uint32_t* goff = lds_u32 + n_buckets;           // [n_buckets] this workgroup's next free position in each bucket
    uint32_t* wave_tot = goff + n_buckets;          // [16]
    float4* window = reinterpret_cast<float4*>(wave_tot + 16 + ((16u - ((2u * n_buckets + 16u) & 3u)) & 3u)); // 16-byte aligned
    __shared__ uint32_t s_total;

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t k = tid; k < n_buckets; k += kSortThreads) goff[k] = offsets[static_cast<uint64_t>(k) * gridDim.x + blockIdx.x];
    const GridParams g = acc->grid;
    const uint64_t begin = blockIdx.x * chunk, end = min(begin + chunk, n_slots);
    // entries of lcur / goff this thread owns in the scan: `per` consecutive ones
    const uint32_t per = (n_buckets + kSortThreads - 1u) / kSortThreads; // <= 5

    for (uint64_t pass = begin; pass < end; pass += T::kPass) {
        // 1. this thread's records, in registers
        float4 rec[PT][RS];
        uint32_t bkt[PT];
        // (every load of the pass issued before the first use: left to itself the compiler sinks each slot's AABB loads below its
        //  flag test and the entity / class loads below the size test — three dependent round trips per slot, slot after slot)
        uint32_t fl_[PT], ent_[PT], extra_[PT], msk_[COMPACT ? 1 : PT];
        float2 b0_[PT], b1_[PT], b2_[PT];
#pragma unroll
        for (uint32_t k = 0; k < PT; ++k) {
            const uint64_t s = min(pass + static_cast<uint64_t>(k) * kSortThreads + tid, end - 1u);
            fl_[k] = flags[s];
            const float2* bp = reinterpret_cast<const float2*>(aabb + 6 * s);
            b0_[k] = bp[0];
            b1_[k] = bp[1];
            b2_[k] = bp[2];
            ent_[k] = entity_of_slot[s];
            extra_[k] = COMPACT ? class_of_slot[s] : group[s];
            if (!COMPACT) msk_[k] = mask[s];
        }
#pragma unroll
        for (uint32_t k = 0; k < PT; ++k) {
            asm volatile("" : "+v"(fl_[k]), "+v"(ent_[k]), "+v"(extra_[k]), "+v"(b0_[k].x), "+v"(b0_[k].y), "+v"(b1_[k].x), "+v"(b1_[k].y),
                         "+v"(b2_[k].x), "+v"(b2_[k].y));
            if (!COMPACT) asm volatile("" : "+v"(msk_[k]));
        }
#pragma unroll
        for (uint32_t k = 0; k < PT; ++k) {
            const uint64_t s = pass + static_cast<uint64_t>(k) * kSortThreads + tid;
            bkt[k] = 0xffffffffu;
#pragma unroll
            for (uint32_t q = 0; q < RS; ++q) rec[k][q] = make_float4(0, 0, 0, 0);
            if (s < end) {
                const uint32_t f = fl_[k];
                const float2 b0 = b0_[k], b1 = b1_[k], b2 = b2_[k];
                const uint32_t ent = ent_[k];
                const uint32_t extra = extra_[k];
                const uint32_t msk = COMPACT ? 0u : msk_[k];
                const float b[6] = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y};
                if (is_body(f) && !body_is_large(g, b)) {
                    const uint32_t c = cell_of(g, b[0], b[1], b[2]);
                    bkt[k] = c >> shift;
                    rec[k][0] = make_float4(b[0], b[1], b[2], __uint_as_float(ent));
                    if (COMPACT) {
                        rec[k][1] = make_float4(b[3], b[4], b[5], __uint_as_float((extra & 255u) | (bkt[k] << 8))); // the bucket rides along
                    } else {
                        rec[k][1] = make_float4(b[3], b[4], b[5], __uint_as_float(c)); // the bucket is c >> shift
                        rec[k][RS - 1u] = make_float4(__uint_as_float(extra), __uint_as_float(msk), __uint_as_float((f & kTypeMask) == 1u ? 1u : 0u),
                                                      __uint_as_float(static_cast<uint32_t>(s)));
                    }
                }
            }
        }
        // 2. count per bucket
        for (uint32_t k = tid; k < n_buckets; k += kSortThreads) lcur[k] = 0;
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < PT; ++k)
            if (bkt[k] != 0xffffffffu) atomicAdd(&lcur[bkt[k]], 1u);
        __syncthreads();
        // 3. exclusive prefix over the buckets (consecutive entries per thread, wave scan, wave totals)
        uint32_t cnt[5];
        uint32_t sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < 5; ++k) {
            const uint32_t e = tid * per + k;
            cnt[k] = (k < per && e < n_buckets) ? lcur[e] : 0u;
            sum += cnt[k];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= static_cast<uint32_t>(off)) incl += t;
        }
        if (lane == 63u) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t base = 0;
        for (uint32_t wv = 0; wv < wave; ++wv) base += wave_tot[wv];
        if (tid == kSortThreads - 1u) s_total = base + incl;
        uint32_t run = base + incl - sum;
#pragma unroll
        for (uint32_t k = 0; k < 5; ++k) {
            const uint32_t e = tid * per + k;
            if (k < per && e < n_buckets) lcur[e] = run;
            run += cnt[k];
        }
        __syncthreads();
        const uint32_t n_local = s_total;
        // 4. rank of every record inside the pass (bucket-major); afterwards lcur[b] is the END of bucket b's local range
        uint32_t rank[PT];
#pragma unroll
        for (uint32_t k = 0; k < PT; ++k) rank[k] = bkt[k] != 0xffffffffu ? atomicAdd(&lcur[bkt[k]], 1u) : 0xffffffffu;
        __syncthreads();
        // 5. through the window, in rank order
        for (uint32_t w0 = 0; w0 < n_local; w0 += T::kWindow) {
#pragma unroll
            for (uint32_t k = 0; k < PT; ++k) {
                const uint32_t r = rank[k] - w0; // wraps for ranks below the window and for 0xffffffff
                if (r < T::kWindow && rank[k] != 0xffffffffu) {
#pragma unroll
                    for (uint32_t q = 0; q < RS; ++q) window[RS * r + q] = rec[k][q];
                }
            }
            __syncthreads();
            const uint32_t n_win = min(T::kWindow, n_local - w0);
            for (uint32_t u = tid; u < RS * n_win; u += kSortThreads) {
                const uint32_t t = u / RS, part = u - t * RS;
                float4 v = window[u];
                const uint32_t tag = __float_as_uint(window[RS * t + 1u].w);
                const uint32_t b = COMPACT ? tag >> 8 : tag >> shift;
                const uint32_t lstart = b ? lcur[b - 1u] : 0u;
                const uint64_t gpos = static_cast<uint64_t>(goff[b]) + (w0 + t - lstart);
                if (COMPACT && part == 1u) v.w = __uint_as_float(tag & 255u);
                coarse[static_cast<uint64_t>(RS) * gpos + part] = v;
            }
            __syncthreads();
        }
        // 6. the next pass of this workgroup continues behind these records
#pragma unroll
        for (uint32_t k = 0; k < 5; ++k) {
            const uint32_t e = tid * per + k;
            if (k < per && e < n_buckets) goff[e] += cnt[k];
        }
        __syncthreads();
    }
}

// Pass B of the LDS sort for ONE bucket, two sweeps over its coarse records: histogram over the bucket's cells, scan ->
// cell_start, then every record to its cell's cursor.  `hist` holds 1 << shift words of LDS.
template <uint32_t THREADS, bool COMPACT>
__device__ __forceinline__ void fine_two_sweeps(uint32_t* hist, uint32_t* wave_tot, const GridParams& g, uint32_t cpb, uint32_t cell0,
                                                uint32_t begin, uint32_t end, const float4* __restrict__ coarse,
                                                float4* __restrict__ sorted, uint32_t* __restrict__ cell_start)
{
    constexpr uint32_t RS = COMPACT ? 2u : 3u;
    for (uint32_t k = threadIdx.x; k < cpb; k += THREADS) hist[k] = 0;
    __syncthreads();
#pragma unroll 4
    for (uint32_t r = begin + threadIdx.x; r < end; r += THREADS) {
        const float4 lo = coarse[static_cast<uint64_t>(RS) * r];
        atomicAdd(&hist[cell_of(g, lo.x, lo.y, lo.z) - cell0], 1u);
    }
    __syncthreads();
    // exclusive scan of the cell counts: cpb / THREADS consecutive cells per thread
    const uint32_t per = cpb / THREADS;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < per; ++k) sum += hist[threadIdx.x * per + k];
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if ((threadIdx.x & 63u) >= static_cast<uint32_t>(off)) incl += t;
    }
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 63u) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (uint32_t k = 0; k < wave; ++k) run += wave_tot[k];
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t cnt = hist[threadIdx.x * per + k];
        hist[threadIdx.x * per + k] = run; // becomes the cell's cursor
        run += cnt;
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < cpb; k += THREADS) cell_start[cell0 + k] = begin + hist[k];
    __syncthreads(); // (cursors are read by the atomics below; keep the plain reads above ahead of them)
#pragma unroll 2
    for (uint32_t r = begin + threadIdx.x; r < end; r += THREADS) {
        const float4 lo = coarse[static_cast<uint64_t>(RS) * r];
        const float4 hi = coarse[static_cast<uint64_t>(RS) * r + 1];
        const uint32_t p = begin + atomicAdd(&hist[cell_of(g, lo.x, lo.y, lo.z) - cell0], 1u);
        sorted[static_cast<uint64_t>(RS) * p] = lo;
        sorted[static_cast<uint64_t>(RS) * p + 1] = hi;
        if (!COMPACT) sorted[3ull * p + 2] = coarse[3ull * r + 2];
    }
}

template <bool COMPACT>
__global__ void __launch_bounds__(kFineThreads) k_sort_fine(const Accum* __restrict__ acc, uint32_t shift, uint32_t n_buckets, uint32_t n_groups,
                                                   const uint32_t* __restrict__ offsets, const float4* __restrict__ coarse,
                                                   float4* __restrict__ sorted, uint32_t* __restrict__ cell_start)
{
    extern __shared__ uint32_t hist[]; // 1 << shift counters, then cursors
    __shared__ uint32_t wave_tot[kFineThreads / 64];
    const uint32_t cpb = 1u << shift;
    const uint32_t bucket = blockIdx.x;
    const GridParams g = acc->grid;
    const uint32_t n_sorted = g.n_bodies - acc->n_large;
    const uint32_t begin = offsets[static_cast<uint64_t>(bucket) * n_groups];
    const uint32_t end = bucket + 1u < n_buckets ? offsets[static_cast<uint64_t>(bucket + 1u) * n_groups] : n_sorted;
    const uint32_t cell0 = bucket << shift;
    if (cell0 > g.n_cells + 2u) return; // beyond the cells in use: no records, and nobody reads cell_start there
    fine_two_sweeps<kFineThreads, COMPACT>(hist, wave_tot, g, cpb, cell0, begin, end, coarse, sorted, cell_start);
}

// Pass B, one-sweep variant (32-byte records): a bucket of up to 8192 records — the usual case, a bucket is 4096 cells — is
// read ONCE into registers (1024 threads x 8 records), counted and ranked by cell in LDS, passed through an LDS window in
// rank order and written out as one contiguous stream.  k_sort_fine reads every record twice (the second sweep does not
// hit L2: profiles/README.md) and stores each one from the thread that happens to hold it.  Bigger buckets (everything in
// one spot) take the two-sweep path, in the same launch.
template <bool COMPACT>
__global__ void __launch_bounds__(kSortThreads, 1)
    k_sort_fine_t(const Accum* __restrict__ acc, uint32_t shift, uint32_t n_buckets, uint32_t n_groups,
                  const uint32_t* __restrict__ offsets, const float4* __restrict__ coarse, float4* __restrict__ sorted,
                  uint32_t* __restrict__ cell_start, uint32_t window_records)
{
    using T = SortT<COMPACT>;
    constexpr uint32_t RS = T::RS, PT = T::PT;
    extern __shared__ uint32_t hist[]; // 1 << shift counters / cursors, then the record window (16-byte aligned: 4 << shift bytes)
    __shared__ uint32_t wave_tot[kSortThreads / 64];
    const uint32_t cpb = 1u << shift;
    float4* window = reinterpret_cast<float4*>(hist + cpb);
    const uint32_t bucket = blockIdx.x;
    const GridParams g = acc->grid;
    const uint32_t n_sorted = g.n_bodies - acc->n_large;
    const uint32_t begin = offsets[static_cast<uint64_t>(bucket) * n_groups];
    const uint32_t end = bucket + 1u < n_buckets ? offsets[static_cast<uint64_t>(bucket + 1u) * n_groups] : n_sorted;
    const uint32_t cell0 = bucket << shift;
    if (cell0 > g.n_cells + 2u) return;
    const uint32_t n_rec = end - begin;
    if (n_rec > T::kPass) { // workgroup-uniform
        fine_two_sweeps<kSortThreads, COMPACT>(hist, wave_tot, g, cpb, cell0, begin, end, coarse, sorted, cell_start);
        return;
    }
    const uint32_t tid = threadIdx.x;
    float4 rec[PT][RS];
    uint32_t cell[PT];
    // (the compiler waits for each record before it asks for the next: every load sits in its own `r < n_rec` block.  Requesting all
    //  eight up front, as k_sort_coarse_t does, made THIS pass slower at 4 M bodies — 60.9 -> 67.4 us: sixteen 16-byte loads per
    //  thread from eight strided streams — so it stays as it is)
#pragma unroll
    for (uint32_t k = 0; k < PT; ++k) {
        const uint32_t r = k * kSortThreads + tid;
        cell[k] = 0xffffffffu;
#pragma unroll
        for (uint32_t q = 0; q < RS; ++q) rec[k][q] = make_float4(0, 0, 0, 0);
        if (r < n_rec) {
#pragma unroll
            for (uint32_t q = 0; q < RS; ++q) rec[k][q] = coarse[static_cast<uint64_t>(RS) * (begin + r) + q];
            cell[k] = cell_of(g, rec[k][0].x, rec[k][0].y, rec[k][0].z) - cell0;
        }
    }
    for (uint32_t k = tid; k < cpb; k += kSortThreads) hist[k] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < PT; ++k)
        if (cell[k] != 0xffffffffu) atomicAdd(&hist[cell[k]], 1u);
    __syncthreads();
    const uint32_t per = cpb / kSortThreads;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < per; ++k) sum += hist[tid * per + k];
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if ((tid & 63u) >= static_cast<uint32_t>(off)) incl += t;
    }
    const uint32_t wave = tid >> 6;
    if ((tid & 63u) == 63u) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (uint32_t k = 0; k < wave; ++k) run += wave_tot[k];
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t cnt = hist[tid * per + k];
        hist[tid * per + k] = run;
        run += cnt;
    }
    __syncthreads();
    for (uint32_t k = tid; k < cpb; k += kSortThreads) cell_start[cell0 + k] = begin + hist[k];
    __syncthreads();
    uint32_t rank[PT];
#pragma unroll
    for (uint32_t k = 0; k < PT; ++k) rank[k] = cell[k] != 0xffffffffu ? atomicAdd(&hist[cell[k]], 1u) : 0xffffffffu;
    for (uint32_t w0 = 0; w0 < n_rec; w0 += window_records) {
#pragma unroll
        for (uint32_t k = 0; k < PT; ++k) {
            const uint32_t r = rank[k] - w0;
            if (r < window_records && rank[k] != 0xffffffffu) {
#pragma unroll
                for (uint32_t q = 0; q < RS; ++q) window[RS * r + q] = rec[k][q];
            }
        }
        __syncthreads();
        const uint32_t n_win = min(window_records, n_rec - w0);
        float4* dst = sorted + static_cast<uint64_t>(RS) * (begin + w0);
        for (uint32_t u = tid; u < RS * n_win; u += kSortThreads) dst[u] = window[u];
        __syncthreads();
    }
}

// Sorted record: everything the pair test and the emission need, so that a hit costs no further (uncoalesced,
// dependent) global loads.
//   full, 48 bytes:     r0 = min.xyz | entity   r1 = max.xyz | cell    r2 = group | mask | static | slot
//   COMPACT, 32 bytes:  r0 = min.xyz | entity   r1 = max.xyz | class   (class -> group, mask, static through the world's
//                       filter palette; the cell is recomputed from min.xyz).  One aligned 32-byte sector per record:
//                       the scatter's random writes and the pair search's L2 -> LDS staging both move 1/3 fewer bytes.
template <bool COMPACT>
__global__ void __launch_bounds__(256) k_bp_scatter(uint64_t n_slots, const uint32_t* __restrict__ flags,
                                                    const float* __restrict__ aabb, const uint32_t* __restrict__ cell_start,
                                                    const uint32_t* __restrict__ body_cell, const uint32_t* __restrict__ body_rank,
                                                    const uint32_t* __restrict__ group, const uint32_t* __restrict__ mask,
                                                    const uint32_t* __restrict__ class_of_slot,
                                                    const uint32_t* __restrict__ entity_of_slot, float4* __restrict__ sorted)
{
    const uint64_t s = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (s >= n_slots) return;
    const uint32_t f = flags[s];
    if (!is_body(f)) return;
    const uint32_t c = body_cell[s];
    if (c == kLargeCell) return;
    const uint32_t pos = cell_start[c] + body_rank[s];
    const float* b = aabb + 6 * s;
    if (COMPACT) {
        sorted[2ull * pos] = make_float4(b[0], b[1], b[2], __uint_as_float(entity_of_slot[s]));
        sorted[2ull * pos + 1] = make_float4(b[3], b[4], b[5], __uint_as_float(class_of_slot[s]));
    } else {
        sorted[3ull * pos] = make_float4(b[0], b[1], b[2], __uint_as_float(entity_of_slot[s]));
        sorted[3ull * pos + 1] = make_float4(b[3], b[4], b[5], __uint_as_float(c));
        sorted[3ull * pos + 2] = make_float4(__uint_as_float(group[s]), __uint_as_float(mask[s]),
                                             __uint_as_float((f & kTypeMask) == 1u ? 1u : 0u), __uint_as_float(static_cast<uint32_t>(s)));
    }
}

struct PairSink {
    unsigned long long* shard_count; // [kShards][8]
    uint2* pairs;                    // staging: kShards slices of shard_cap pairs
    uint64_t shard_cap;
    // Slab window of the sharded broadphase (PairWindow): a pair is reported only where the lower end of its overlap
    // interval along `axis`, max(min_a, min_b), falls into [win_lo, win_hi).  (-inf, +inf) = report everything.
    uint32_t axis;
    float win_lo, win_hi;
};

__device__ __forceinline__ float axis_of(const float4& v, uint32_t axis) { return axis == 0u ? v.x : (axis == 1u ? v.y : v.z); }
__device__ __forceinline__ bool in_window(const PairSink& s, const float4& alo, const float4& blo)
{
    const float m = fmaxf(axis_of(alo, s.axis), axis_of(blo, s.axis));
    return m >= s.win_lo && m < s.win_hi;
}

// Wave-compacted append with per-wave staging in LDS.  Each call ballots the hits of the wave, packs them
// behind the wave's staging cursor (an SGPR-uniform count), and whenever 64 pairs are staged writes them out
// as one contiguous 512-byte burst behind ONE global atomic.  A single counter word saturates near 10^8
// atomics/s, so one atomic per hit-bearing iteration (the first version) cost 39 ms for 12.6 M pairs.
constexpr uint32_t kStage = 192;     // 127 carried + 64 new
constexpr uint32_t kFlush = 128;     // pairs written per global atomic (2 per lane, 1 KiB contiguous)

struct WaveStage {
    uint2* buf;     // LDS, kStage entries owned by this wave
    uint32_t fill;  // wave-uniform
};

__device__ __forceinline__ void stage_flush(const PairSink& sink, WaveStage& st, uint32_t n_out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t shard = blockIdx.x % kShards;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&sink.shard_count[shard * 8u], static_cast<unsigned long long>(n_out));
    base = __shfl(base, 0, 64);
    uint2* dst = sink.pairs + static_cast<uint64_t>(shard) * sink.shard_cap;
    for (uint32_t k = lane; k < n_out; k += 64u) {
        if (base + k < sink.shard_cap) dst[base + k] = st.buf[k];
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

// Lane-private pair lists (wave search).  emit_pairs costs a ballot, a rank, an LDS write, a cursor update and a flush test
// per candidate, hit or not — in a loop whose useful part is six comparisons.  Here a hit is one exec-masked ds_write into
// the lane's own column of a [kLaneCap][64] LDS array (no lane ever reads another's entries) and a counter increment; the
// lists of a 64-body block are written out together — one scan of the counts, ONE global atomic, every lane storing its own
// entries — at the end of the block, or earlier if a lane could overflow in the next trip.
#ifndef BGE_LANE_CAP
#define BGE_LANE_CAP 8
#endif
constexpr uint32_t kLaneCap = BGE_LANE_CAP;
struct LaneList {
    uint32_t* buf; // LDS, [kLaneCap][64], this wave's
    uint32_t cnt;  // this lane's entries
};
__device__ __forceinline__ void lane_push(LaneList& l, bool hit, uint32_t partner)
{
    if (hit) {
        l.buf[(threadIdx.x & 63u) + 64u * l.cnt] = partner;
        ++l.cnt;
    }
}
__device__ __forceinline__ void lane_flush(const PairSink& sink, LaneList& l, uint32_t own_entity)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t incl = l.cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= static_cast<uint32_t>(off)) incl += t;
    }
    const uint32_t total = __shfl(incl, 63, 64);
    if (total == 0) return;
    const uint32_t shard = blockIdx.x % kShards;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&sink.shard_count[shard * 8u], static_cast<unsigned long long>(total));
    base = __shfl(base, 0, 64) + (incl - l.cnt);
    uint2* dst = sink.pairs + static_cast<uint64_t>(shard) * sink.shard_cap;
    for (uint32_t k = 0; __any(k < l.cnt); ++k) {
        if (k < l.cnt) {
            const uint32_t partner = l.buf[lane + 64u * k];
            if (base + k < sink.shard_cap) dst[base + k] = make_uint2(min(own_entity, partner), max(own_entity, partner));
        }
    }
    l.cnt = 0;
}

// The flush at the END of a block in two halves, so that the atomic's round trip runs beside the next block's first loads:
// lane_flush_begin scans the counts and issues the atomic, lane_flush_end (after the next block's loads have been issued)
// picks the result up and stores the entries.
struct LanePending {
    unsigned long long base; // lane 0: the atomic's result (in flight until lane_flush_end reads it)
    uint32_t excl;           // entries of the lanes below this one
    uint32_t entity;         // the entity this lane's entries belong to
    uint32_t total;          // wave-uniform
};
__device__ __forceinline__ LanePending lane_flush_begin(const PairSink& sink, const LaneList& l, uint32_t own_entity)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t incl = l.cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= static_cast<uint32_t>(off)) incl += t;
    }
    LanePending p{0ull, incl - l.cnt, own_entity, static_cast<uint32_t>(__shfl(incl, 63, 64))};
    if (p.total != 0u && lane == 0u) p.base = atomicAdd(&sink.shard_count[(blockIdx.x % kShards) * 8u], static_cast<unsigned long long>(p.total));
    return p;
}
__device__ __forceinline__ void lane_flush_end(const PairSink& sink, LaneList& l, const LanePending& p)
{
    if (p.total == 0u) return;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long base = __shfl(p.base, 0, 64) + p.excl;
    uint2* dst = sink.pairs + static_cast<uint64_t>(blockIdx.x % kShards) * sink.shard_cap;
    for (uint32_t k = 0; __any(k < l.cnt); ++k) {
        if (k < l.cnt) {
            const uint32_t partner = l.buf[lane + 64u * k];
            if (base + k < sink.shard_cap) dst[base + k] = make_uint2(min(p.entity, partner), max(p.entity, partner));
        }
    }
    l.cnt = 0;
}

__device__ __forceinline__ bool overlap(const float4& alo, const float4& ahi, const float4& blo, const float4& bhi)
{
    // All six comparisons, no short circuit: for `&&` the compiler builds a cascade of exec-mask branches (one per axis, the
    // far corner fetched lazily) that costs more issue slots than it saves — measured at 4 M bodies, step 576 -> 566 us.
    const int x = static_cast<int>(alo.x <= bhi.x) & static_cast<int>(ahi.x >= blo.x);
    const int y = static_cast<int>(alo.y <= bhi.y) & static_cast<int>(ahi.y >= blo.y);
    const int z = static_cast<int>(alo.z <= bhi.z) & static_cast<int>(ahi.z >= blo.z);
    return (x & y & z) != 0;
}

__device__ __forceinline__ bool filter_ok(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ group,
                                          const uint32_t* __restrict__ mask, uint32_t sa, uint32_t sb)
{
    const bool both_static = (flags[sa] & kTypeMask) == 1u && (flags[sb] & kTypeMask) == 1u;
    return !both_static && (group[sa] & mask[sb]) != 0 && (group[sb] & mask[sa]) != 0;
}

// Pair search, LDS-tiled.  A workgroup owns 256 consecutive sorted bodies.  For each of the 5 neighbour runs the
// union of its bodies' candidate ranges is one contiguous range of sorted records (records are sorted by cell, x
// fastest); it is staged through LDS in coalesced chunks of kChunk records, and every lane tests only the part of the
// chunk that belongs to ITS OWN candidate range.  The first version walked the ranges with dependent global loads
// (one or two candidates in flight per lane) and was latency-bound at 1.45 ms for 4 M bodies.
constexpr uint32_t kChunk = 256; // records per staged chunk: 12 KiB of LDS (with 6 KiB of pair staging: 8 workgroups per CU)

__device__ __forceinline__ bool filter_rec(const float4& a2, const float4& b2)
{
    const bool both_static = __float_as_uint(a2.z) != 0u && __float_as_uint(b2.z) != 0u;
    return !both_static && (__float_as_uint(a2.x) & __float_as_uint(b2.y)) != 0u && (__float_as_uint(b2.x) & __float_as_uint(a2.y)) != 0u;
}

// WINDOW: apply the slab window of the sharded broadphase (a second instantiation keeps the single-GPU search free of it:
// the extra sink fields cost ~5 % there, measured)
__device__ __forceinline__ bool filter_tab(const uint4& a, const uint4& b)
{
    return !(a.z != 0u && b.z != 0u) && (a.x & b.y) != 0u && (b.x & a.y) != 0u;
}

template <bool WINDOW, bool COMPACT>
__global__ void __launch_bounds__(256) k_bp_pairs(const Accum* __restrict__ acc, const uint32_t* __restrict__ cell_start,
                                                  const float4* __restrict__ sorted, const uint4* __restrict__ filter_table,
                                                  PairSink sink)
{
    constexpr uint32_t RS = COMPACT ? 2u : 3u; // float4 per record
    __shared__ float4 cand[RS * kChunk];
    __shared__ uint2 stage_lds[4][kStage];
    __shared__ uint4 s_tab[COMPACT ? 256 : 1]; // (group, mask, static, 0) per filter class
    __shared__ uint32_t s_cell_first, s_cell_last;

    const GridParams g = acc->grid;
    if (COMPACT) s_tab[threadIdx.x] = filter_table[threadIdx.x]; // visible after the first barrier below
    const uint32_t n_sorted = g.n_bodies - acc->n_large;
    const uint32_t n_blocks = (n_sorted + 255u) / 256u;
    WaveStage st{stage_lds[threadIdx.x >> 6], 0u};
    const uint32_t tid = threadIdx.x;

    for (uint32_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const uint32_t i = blk * 256u + tid;
        const bool active = i < n_sorted;
        float4 lo = make_float4(0, 0, 0, 0), hi = lo, fi = lo;
        uint32_t cell = 0;
        if (active) {
            lo = sorted[static_cast<uint64_t>(RS) * i];
            hi = sorted[static_cast<uint64_t>(RS) * i + 1];
            if (COMPACT) {
                cell = cell_of(g, lo.x, lo.y, lo.z); // the same function of the same bits as in k_bp_count
            } else {
                fi = sorted[3ull * i + 2];
                cell = __float_as_uint(hi.w);
            }
        }
        const uint32_t entity_i = __float_as_uint(lo.w);
        const uint32_t last_tid = min(255u, n_sorted - 1u - blk * 256u);
        __syncthreads(); // previous block iteration is done with s_cell_*
        if (tid == 0) s_cell_first = cell;
        if (tid == last_tid) s_cell_last = cell;
        __syncthreads();
        const uint32_t cell_first = s_cell_first, cell_last = s_cell_last;
        uint4 own = make_uint4(0, 0, 0, 0);
        if (COMPACT && active) own = s_tab[__float_as_uint(hi.w) & 255u];

#pragma unroll 1
        for (int row = 0; row < 5; ++row) {
            // this lane's candidate range [j, end) and the workgroup's union [r_lo, r_hi)
            uint32_t j = 0, end = 0, r_lo, r_hi;
            if (row == 0) {
                // own cell's later records + the cell to the right
                if (active) {
                    j = i + 1;
                    end = cell_start[cell + 2];
                }
                r_lo = blk * 256u + 1u;
                r_hi = cell_start[cell_last + 2];
            } else {
                const int dy = (row == 1) ? 1 : (row - 3); // rows 2,3,4 -> dy = -1,0,1 at dz = 1
                const int dz = (row == 1) ? 0 : 1;
                const uint32_t off = static_cast<uint32_t>(dy * static_cast<int>(g.dim_x)) + static_cast<uint32_t>(dz) * g.dim_xy;
                if (active) {
                    j = cell_start[cell + off - 1];
                    end = cell_start[cell + off + 2];
                }
                r_lo = cell_start[cell_first + off - 1];
                r_hi = cell_start[cell_last + off + 2];
            }
            for (uint32_t base = r_lo; base < r_hi; base += kChunk) {
                const uint32_t top = min(base + kChunk, r_hi);
                // skip chunks no lane needs (sparse worlds: far-apart cells with crowded cells in between)
                const uint32_t jj0 = max(j, base);
                const uint32_t e0 = min(end, top);
                if (!__syncthreads_or(jj0 < e0)) continue;
                for (uint32_t k = tid; k < RS * (top - base); k += 256u) cand[k] = sorted[static_cast<uint64_t>(RS) * base + k];
                __syncthreads();
                uint32_t jj = jj0;
                while (__any(jj < e0)) {
                    // two candidates per trip: both LDS fetches are in flight together
                    bool hit0 = false, hit1 = false;
                    uint32_t e0j = 0, e1j = 0;
                    if (jj < e0) {
                        const uint32_t k0 = RS * (jj - base);
                        const bool two = jj + 1u < e0;
                        const uint32_t k1 = two ? k0 + RS : k0;
                        const float4 alo = cand[k0], ahi = cand[k0 + 1u];
                        const float4 blo = cand[k1], bhi = cand[k1 + 1u];
                        if (overlap(lo, hi, alo, ahi)) {
                            hit0 = (COMPACT ? filter_tab(own, s_tab[__float_as_uint(ahi.w) & 255u]) : filter_rec(fi, cand[k0 + RS - 1u])) &&
                                   (!WINDOW || in_window(sink, lo, alo));
                            e0j = __float_as_uint(alo.w);
                        }
                        if (two && overlap(lo, hi, blo, bhi)) {
                            hit1 = (COMPACT ? filter_tab(own, s_tab[__float_as_uint(bhi.w) & 255u]) : filter_rec(fi, cand[k1 + RS - 1u])) &&
                                   (!WINDOW || in_window(sink, lo, blo));
                            e1j = __float_as_uint(blo.w);
                        }
                        jj += 2u;
                    }
                    if (__any(hit0 || hit1)) {
                        emit_pairs(sink, st, hit0, entity_i, e0j);
                        emit_pairs(sink, st, hit1, entity_i, e1j);
                    }
                }
                // (the __syncthreads_or at the top of the next iteration protects cand before it is overwritten)
            }
        }
    }
    if (st.fill) stage_flush(sink, st, st.fill);
}

// Pair search, wave-granular.  Same algorithm as k_bp_pairs, but the unit of work is ONE WAVE owning 64 consecutive
// sorted bodies: the candidate chunks are staged in a wave-private LDS region and handed between the lanes of the wave
// with wave_sync() (a wave's DS operations execute in order), so the kernel has no workgroup barrier at all.  The
// workgroup version above spent its time waiting — two __syncthreads per staged chunk behind dependent cell_start and
// record loads — not moving bytes: shrinking the records from 48 to 32 bytes left its 355 us untouched.
#ifndef BGE_WAVE_CHUNK
#define BGE_WAVE_CHUNK 88 /* 8 workgroups per CU (64 VGPRs, 12 B scratch) with 32-byte records; measured at 4 M bodies, step time:
                             96 -> 583 us (7 per CU), 88 -> 577, 80 -> 586, 72 -> 604 (more rows need a second chunk) */
#endif
constexpr uint32_t kWaveChunk = BGE_WAVE_CHUNK; // records per staged chunk and wave (a row's union range is ~70 records at 0.7 bodies per cell)
constexpr uint32_t kListBytes = 4u * kLaneCap * 64u * 4u; // the four waves' pair lists
// workgroups per CU the LDS footprint allows (wave chunk regions + the pair lists + 2 KiB filter table)
constexpr uint32_t kWaveResidentCompact = (160u * 1024u) / (4u * 2u * kWaveChunk * 16u + kListBytes + 2048u + 96u) > 8u ? 8u : (160u * 1024u) / (4u * 2u * kWaveChunk * 16u + kListBytes + 2048u + 96u);
constexpr uint32_t kWaveResidentFull = (160u * 1024u) / (4u * 3u * kWaveChunk * 16u + kListBytes + 64u) > 8u ? 8u : (160u * 1024u) / (4u * 3u * kWaveChunk * 16u + kListBytes + 64u);
// SMALL (with COMPACT): at most 32 filter classes in the scene — the usual case.  Instead of the (group, mask, static) table
// the workgroup keeps one 32-bit word per class, bit c = "may pair with class c"; a lane holds its own class's word and the
// filter of a candidate is one shift instead of two dependent LDS reads and a dozen bit operations.
// (the SMALL variant has no 2 KiB filter table)
constexpr uint32_t kWaveResidentSmall = (160u * 1024u) / (4u * 2u * kWaveChunk * 16u + kListBytes + 128u + 96u) > 8u ? 8u : (160u * 1024u) / (4u * 2u * kWaveChunk * 16u + kListBytes + 128u + 96u);
template <bool WINDOW, bool COMPACT, bool SMALL = false>
__global__ void __launch_bounds__(256, SMALL ? kWaveResidentSmall : (COMPACT ? kWaveResidentCompact : kWaveResidentFull)) k_bp_pairs_wave(const Accum* __restrict__ acc, const uint32_t* __restrict__ cell_start,
                                                       const float4* __restrict__ sorted, const uint4* __restrict__ filter_table,
                                                       PairSink sink)
{
    constexpr uint32_t RS = COMPACT ? 2u : 3u; // float4 per record
    __shared__ float4 cand_all[4][RS * kWaveChunk];
    __shared__ uint32_t lists_lds[4][kLaneCap * 64];
    __shared__ uint2 s_tab[(COMPACT && !SMALL) ? 256 : 1];      // (group, mask) per filter class
    __shared__ uint32_t s_static[(COMPACT && !SMALL) ? 8 : 1];  // static bit per filter class
    __shared__ uint32_t s_compat[SMALL ? 32 : 1];               // SMALL: classes a class may pair with

    const GridParams g = acc->grid;
    if (SMALL) {
        if (threadIdx.x < 64u) {
            // one load per lane, the other classes' entries by readlane (a loop over filter_table[c] was 96 dependent scalar loads)
            const uint4 a = filter_table[threadIdx.x & 31u];
            uint32_t m = 0;
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                const uint4 o = make_uint4(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(a.x), c)),
                                           static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(a.y), c)),
                                           static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(a.z), c)), 0u);
                m |= filter_tab(a, o) ? 1u << c : 0u; // (unused entries are zero: never pair)
            }
            if (threadIdx.x < 32u) s_compat[threadIdx.x] = m;
        }
        __syncthreads();
    } else if (COMPACT) {
        const uint4 e = filter_table[threadIdx.x];
        s_tab[threadIdx.x] = make_uint2(e.x, e.y);
        const unsigned long long sm = __ballot(e.z != 0u);
        if ((threadIdx.x & 63u) == 0u) {
            s_static[2u * (threadIdx.x >> 6)] = static_cast<uint32_t>(sm);
            s_static[2u * (threadIdx.x >> 6) + 1u] = static_cast<uint32_t>(sm >> 32);
        }
        __syncthreads(); // the only workgroup barrier: the filter table is shared
    }
    const uint32_t n_sorted = g.n_bodies - acc->n_large;
    const uint32_t n_blocks = (n_sorted + 63u) / 64u;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    float4* cand = cand_all[wave];
    LaneList found{lists_lds[wave], 0u};
    LanePending pending{0ull, 0u, 0u, 0u};

    for (uint32_t blk = blockIdx.x * 4u + wave; blk < n_blocks; blk += gridDim.x * 4u) {
        const uint32_t i = blk * 64u + lane;
        const bool active = i < n_sorted;
        float4 lo = make_float4(0, 0, 0, 0), hi = lo, fi = lo;
        uint32_t cell = 0;
        if (active) {
            lo = sorted[static_cast<uint64_t>(RS) * i];
            hi = sorted[static_cast<uint64_t>(RS) * i + 1];
            if (!COMPACT) fi = sorted[3ull * i + 2];
        }
        // the previous block's pairs go out while this block's own record is on its way
        lane_flush_end(sink, found, pending);
        if (active) cell = COMPACT ? cell_of(g, lo.x, lo.y, lo.z) : __float_as_uint(hi.w);
        const uint32_t entity_i = __float_as_uint(lo.w);
        const uint32_t last_lane = min(63u, n_sorted - 1u - blk * 64u);
        uint4 own = make_uint4(0, 0, 0, 0);
        auto class_entry = [&](uint32_t cls) {
            const uint2 gm = s_tab[cls & 255u];
            return make_uint4(gm.x, gm.y, (s_static[(cls & 255u) >> 5] >> (cls & 31u)) & 1u, 0u);
        };
        uint32_t own_compat = 0;
        if (SMALL) {
            if (active) own_compat = s_compat[__float_as_uint(hi.w) & 31u];
        } else if (COMPACT && active) {
            own = class_entry(__float_as_uint(hi.w));
        }

        // a row's candidate range of this lane, [j, end): row 0 = own cell's later records + the cell to the right; rows 1..4 =
        // three cells of the row at dy = 1 (dz = 0) and at dy = -1, 0, 1 (dz = 1)
        auto row_range = [&](int row, uint32_t& j, uint32_t& end) {
            j = 0;
            end = 0;
            if (!active) return;
            if (row == 0) {
                j = i + 1;
                end = cell_start[cell + 2];
            } else {
                const int dy = (row == 1) ? 1 : (row - 3);
                const int dz = (row == 1) ? 0 : 1;
                const uint32_t off = static_cast<uint32_t>(dy * static_cast<int>(g.dim_x)) + static_cast<uint32_t>(dz) * g.dim_xy;
                j = cell_start[cell + off - 1];
                end = cell_start[cell + off + 2];
            }
        };
        uint32_t next_j, next_end;
        row_range(0, next_j, next_end);
#pragma unroll 1
        for (int row = 0; row < 5; ++row) {
            // the next row's range is in flight while this row is searched; the wave's union [r_lo, r_hi) is the first lane's
            // start and the last lane's end (cells ascend along the sorted records) — no loads of its own
            const uint32_t j = next_j, end = next_end;
            if (row < 4) row_range(row + 1, next_j, next_end);
            const uint32_t r_lo = __shfl(j, 0, 64), r_hi = __shfl(end, static_cast<int>(last_lane), 64);
            for (uint32_t base = r_lo; base < r_hi; base += kWaveChunk) {
                const uint32_t top = min(base + kWaveChunk, r_hi);
                const uint32_t jj0 = max(j, base);
                const uint32_t e0 = min(end, top);
                if (!__any(jj0 < e0)) continue;
                wave_sync(); // the previous chunk has been consumed
                // (staging the corners as separate arrays — all min corners, then all max corners — so that the lanes of a
                //  ds_read_b128 pass hit consecutive 16-byte bank groups was measured: SQ_LDS_BANK_CONFLICT 11.3 M -> 10.6 M, step
                //  0.504 -> 0.520 ms; rejected)
                {
                    // all loads of the chunk issued before the first LDS store (as a loop: one global round trip after the other)
                    constexpr uint32_t kLoads = (RS * kWaveChunk + 63u) / 64u;
                    const uint32_t cnt = RS * (top - base);
                    const float4* src = sorted + static_cast<uint64_t>(RS) * base;
                    float4 t[kLoads];
#pragma unroll
                    for (uint32_t q = 0; q < kLoads; ++q) t[q] = src[min(lane + 64u * q, cnt - 1u)];
#pragma unroll
                    for (uint32_t q = 0; q < kLoads; ++q) { // (keeps the compiler from sinking each load into its guarded store)
                        asm volatile("" : "+v"(t[q].x), "+v"(t[q].y), "+v"(t[q].z), "+v"(t[q].w));
                    }
#pragma unroll
                    for (uint32_t q = 0; q < kLoads; ++q) {
                        if (lane + 64u * q < cnt) cand[lane + 64u * q] = t[q];
                    }
                }
                wave_sync();
                uint32_t jj = jj0;
                while (__any(jj < e0)) {
                    // two candidates per trip: both LDS fetches are in flight together
                    bool hit0 = false, hit1 = false;
                    uint32_t e0j = 0, e1j = 0;
                    if (jj < e0) {
                        const uint32_t k0 = RS * (jj - base);
                        const bool two = jj + 1u < e0;
                        const uint32_t k1 = two ? k0 + RS : k0;
                        const float4 alo = cand[k0], ahi = cand[k0 + 1u];
                        const float4 blo = cand[k1], bhi = cand[k1 + 1u];
                        if (overlap(lo, hi, alo, ahi)) {
                            hit0 = (SMALL ? ((own_compat >> (__float_as_uint(ahi.w) & 31u)) & 1u) != 0u
                                          : (COMPACT ? filter_tab(own, class_entry(__float_as_uint(ahi.w))) : filter_rec(fi, cand[k0 + RS - 1u]))) &&
                                   (!WINDOW || in_window(sink, lo, alo));
                            e0j = __float_as_uint(alo.w);
                        }
                        if (two && overlap(lo, hi, blo, bhi)) {
                            hit1 = (SMALL ? ((own_compat >> (__float_as_uint(bhi.w) & 31u)) & 1u) != 0u
                                          : (COMPACT ? filter_tab(own, class_entry(__float_as_uint(bhi.w))) : filter_rec(fi, cand[k1 + RS - 1u]))) &&
                                   (!WINDOW || in_window(sink, lo, blo));
                            e1j = __float_as_uint(blo.w);
                        }
                        jj += 2u;
                    }
                    lane_push(found, hit0, e0j);
                    lane_push(found, hit1, e1j);
                    if (__any(found.cnt + 2u > kLaneCap)) lane_flush(sink, found, entity_i); // (a trip adds at most two)
                }
            }
        }
        pending = lane_flush_begin(sink, found, entity_i);
    }
    lane_flush_end(sink, found, pending);
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
            s_large = body_is_large(acc->grid, b);
        }
        for (uint32_t k = 0; k < n_large; ++k) {
            const uint32_t L = large_list[k];
            const float* b = aabb + 6ull * L;
            const float4 llo = make_float4(b[0], b[1], b[2], 0), lhi = make_float4(b[3], b[4], b[5], 0);
            // large-vs-small: reported from the small body's side; large-vs-large: from the lower slot's side
            bool hit = body && s != L && (!s_large || s < L) && overlap(lo, hi, llo, lhi);
            if (hit) hit = filter_ok(flags, group, mask, static_cast<uint32_t>(s), L) && in_window(sink, lo, llo);
            emit_pairs(sink, st, hit, hit ? entity_of_slot[s] : 0u, hit ? entity_of_slot[L] : 0u);
        }
    }
    if (st.fill) stage_flush(sink, st, st.fill);
}

// ---- box queries against the sorted bodies of this run (trigger ghosts when a scene carries many of them)
// A body sits in the cell of its min corner and a small body is narrower than one cell, so the bodies that can overlap a box
// [mn, mx] have their min corner in the cells cell(mn) - 1 .. cell(mx) of every axis (clamped like the bodies' own cells).
// Boxes covering more than kQueryMaxCells cells are handed back to the caller's all-bodies pass: one wave walking a large
// part of the table would be slower than that pass.
constexpr uint32_t kQueryMaxCells = 8192;
struct BoxCells {
    uint32_t lo[3], n[3];
};
__device__ __forceinline__ bool box_cells(const GridParams& g, const float* b, BoxCells& c)
{
    if (!(b[0] <= b[3] && b[1] <= b[4] && b[2] <= b[5]) || g.n_cells == 0u) return false; // an empty box (inactive ghost), or no grid
    const uint32_t dim[3] = {g.dim_x, g.dim_xy / g.dim_x, g.n_cells / g.dim_xy};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const uint32_t lo = min(max(cell_axis(g, b[a], a), 2u) - 1u, dim[a] - 2u);
        const uint32_t hi = min(cell_axis(g, b[3 + a], a), dim[a] - 2u);
        c.lo[a] = lo;
        c.n[a] = hi >= lo ? hi - lo + 1u : 1u;
    }
    return true;
}

// counters: [0] hits, [1] boxes in big_list, [2] boxes in grid_list
__global__ void __launch_bounds__(256) k_bp_classify_boxes(const Accum* __restrict__ acc, uint32_t n_boxes, const float* __restrict__ box,
                                                           uint32_t* __restrict__ counters, uint32_t* __restrict__ big_list,
                                                           uint32_t* __restrict__ grid_list)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_boxes) return;
    BoxCells c;
    if (!box_cells(acc->grid, box + 6ull * i, c)) return;
    const unsigned long long cells = static_cast<unsigned long long>(c.n[0]) * c.n[1] * c.n[2];
    if (cells <= kQueryMaxCells) grid_list[atomicAdd(&counters[2], 1u)] = i;
    else big_list[atomicAdd(&counters[1], 1u)] = i;
}

// One wave per box; each lane walks one row (consecutive cells in x = consecutive sorted records) of the box's cell range.
// A hit is (box index, body entity), appended behind one atomic per ballot — the format of the all-bodies pass.
template <bool COMPACT>
__global__ void __launch_bounds__(256) k_bp_query_boxes(const Accum* __restrict__ acc, const uint32_t* __restrict__ cell_start,
                                                        const float4* __restrict__ sorted, const uint4* __restrict__ filter_table,
                                                        const float* __restrict__ box, const uint32_t* __restrict__ box_group,
                                                        const uint32_t* __restrict__ box_mask, const uint32_t* __restrict__ box_entity,
                                                        const uint32_t* __restrict__ grid_list, uint32_t* __restrict__ counters,
                                                        const uint32_t* __restrict__ large_list, const float* __restrict__ aabb,
                                                        const uint32_t* __restrict__ flags, const uint32_t* __restrict__ group,
                                                        const uint32_t* __restrict__ mask, const uint32_t* __restrict__ entity_of_slot,
                                                        uint2* __restrict__ out, uint32_t cap)
{
    constexpr uint32_t RS = COMPACT ? 2u : 3u;
    const GridParams g = acc->grid;
    const uint32_t n_grid = counters[2];
    const uint32_t n_large = acc->n_large;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    auto emit = [&](bool hit, uint32_t i, uint32_t ent) {
        const unsigned long long m = __ballot(hit);
        if (m == 0) return;
        const uint32_t leader = static_cast<uint32_t>(__ffsll(static_cast<long long>(m))) - 1u;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&counters[0], static_cast<uint32_t>(__popcll(m)));
        base = __shfl(base, static_cast<int>(leader), 64);
        if (hit) {
            const uint32_t at = base + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
            if (at < cap) out[at] = make_uint2(i, ent);
        }
    };
    for (uint32_t k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); k < n_grid; k += waves) {
        const uint32_t i = grid_list[k];
        float b[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) b[a] = box[6ull * i + a];
        BoxCells c;
        if (!box_cells(g, b, c)) continue; // (classified with the same function: not taken)
        const float4 blo = make_float4(b[0], b[1], b[2], 0), bhi = make_float4(b[3], b[4], b[5], 0);
        const uint32_t tg = box_group[i], tm = box_mask[i], te = box_entity[i];
        const uint32_t rows = c.n[1] * c.n[2];
        for (uint32_t r0 = 0; r0 < rows; r0 += 64u) {
            const uint32_t r = r0 + lane;
            uint32_t j = 0, end = 0;
            if (r < rows) {
                const uint32_t c0 = c.lo[0] + g.dim_x * (c.lo[1] + r % c.n[1]) + g.dim_xy * (c.lo[2] + r / c.n[1]);
                j = cell_start[c0];
                end = cell_start[c0 + c.n[0]];
            }
            while (__any(j < end)) {
                bool hit = false;
                uint32_t ent = 0;
                if (j < end) {
                    const float4 lo = sorted[static_cast<uint64_t>(RS) * j], hi = sorted[static_cast<uint64_t>(RS) * j + 1];
                    if (overlap(blo, bhi, lo, hi)) {
                        uint4 e; // (group, mask, static)
                        if (COMPACT) {
                            e = filter_table[__float_as_uint(hi.w) & 255u];
                        } else {
                            const float4 f = sorted[3ull * j + 2];
                            e = make_uint4(__float_as_uint(f.x), __float_as_uint(f.y), __float_as_uint(f.z), 0u);
                        }
                        ent = __float_as_uint(lo.w);
                        // (Static bodies included: the pair cache pairs a ghost with every registered object whose filter passes)
                        hit = ent != te && (tg & e.y) != 0u && (e.x & tm) != 0u;
                    }
                    ++j;
                }
                emit(hit, i, ent);
            }
        }
        // the bodies too wide for the grid
        for (uint32_t l0 = 0; l0 < n_large; l0 += 64u) {
            bool hit = false;
            uint32_t ent = 0;
            if (l0 + lane < n_large) {
                const uint32_t L = large_list[l0 + lane];
                const float* q = aabb + 6ull * L;
                if (overlap(blo, bhi, make_float4(q[0], q[1], q[2], 0), make_float4(q[3], q[4], q[5], 0))) {
                    ent = entity_of_slot[L];
                    hit = (flags[L] & kTypeMask) != 0u && ent != te && (tg & mask[L]) != 0u && (group[L] & tm) != 0u;
                }
            }
            emit(hit, i, ent);
        }
    }
}

// Shard slices -> one compact list.  Block (shard, part): offset of the shard = sum of the kept counts before it.
constexpr uint32_t kCompactParts = 32;
__global__ void __launch_bounds__(256) k_bp_compact(Accum* acc, const uint2* __restrict__ staged, uint64_t shard_cap,
                                                    uint2* __restrict__ out, uint64_t out_cap)
{
    const uint32_t shard = blockIdx.x / kCompactParts;
    const uint32_t part = blockIdx.x % kCompactParts;
    unsigned long long offset = 0, total = 0, kept_total = 0;
    for (uint32_t t = 0; t < kShards; ++t) {
        const unsigned long long c = acc->shard_count[t][0];
        const unsigned long long k = c < shard_cap ? c : shard_cap;
        if (t < shard) offset += k;
        total += c;
        kept_total += k;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        acc->n_pairs = total;
        acc->n_pairs_kept = kept_total < out_cap ? kept_total : out_cap;
    }
    const unsigned long long c = acc->shard_count[shard][0];
    const unsigned long long kept = c < shard_cap ? c : shard_cap;
    const unsigned long long per = (kept + kCompactParts - 1) / kCompactParts;
    const unsigned long long lo = per * part;
    const unsigned long long hi = lo + per < kept ? lo + per : kept;
    const uint2* src = staged + static_cast<uint64_t>(shard) * shard_cap;
    for (unsigned long long k = lo + threadIdx.x; k < hi; k += 256) {
        if (offset + k < out_cap) out[offset + k] = src[k];
    }
}

// Each shard can hold far more than its even share, so that a skewed or tiny scene (few workgroups -> few shards in
// use) does not drop pairs: min(capacity, max(capacity / 16, 65536)).
inline uint64_t shard_capacity(uint64_t capacity)
{
    return std::min<uint64_t>(std::max<uint64_t>(capacity, 1), std::max<uint64_t>(capacity / 16, 65536));
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
    for (void** p : {&pairs_, &scan_stage_, &counters_, &cell_count_, &cell_start_, &scan_tmp_, &scan_status_, &sort_matrix_,
                     &sort_offsets_, &sort_status_, &coarse_, &sorted_slot_, &sorted_aabb_, &body_cell_,
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
    BP_TRY(hipMalloc(&scan_stage_, shard_capacity(capacity_) * kShards * 8)); // sharded staging of the pair list
    BP_TRY(hipMalloc(&counters_, sizeof(Accum)));
    BP_TRY(hipMalloc(&cell_count_, (static_cast<size_t>(table_size_) + 2 * kScanTile) * 4)); // whole scan tiles, zero-padded
    BP_TRY(hipMalloc(&cell_start_, (static_cast<size_t>(table_size_) + 2 * kScanTile) * 4));
    BP_TRY(hipMalloc(&scan_tmp_, (static_cast<size_t>(table_size_) / kScanBlock + 2) * 4));
    BP_TRY(hipMalloc(&scan_status_, (static_cast<size_t>(table_size_) / kScanBlock + 2) * 8));
    BP_TRY(hipMemset(scan_status_, 0, (static_cast<size_t>(table_size_) / kScanBlock + 2) * 8));
    scan_epoch_ = 0;
    if (const char* e = std::getenv("BGE_BP_SCAN")) three_kernel_scan_ = std::atoi(e) == 3; // A/B: BGE_BP_SCAN=3 keeps the three-kernel scan
    if (const char* e = std::getenv("BGE_BP_PAIRS")) block_pairs_ = std::string(e) == "block";  // A/B: workgroup-granular pair search
    sort_groups_ = kSortGroups;
    if (const char* e = std::getenv("BGE_BP_SORT_GROUPS")) sort_groups_ = std::min<uint32_t>(kSortGroups, std::max(1, std::atoi(e))); // tests: several passes per workgroup at small n
    if (const char* e = std::getenv("BGE_BP_BOUNDS")) fused_bounds_ = std::string(e) != "pass"; // A/B and tests: k_bp_bounds' own pass over the AABBs
    if (const char* e = std::getenv("BGE_BP_PARAMS")) fused_params_ = std::string(e) != "split"; // A/B: k_bp_params as its own launch
    if (const char* e = std::getenv("BGE_BP_FILTER")) small_palette_ = std::string(e) != "table"; // A/B and tests: keep the (group, mask) table
    if (const char* e = std::getenv("BGE_BP_COARSE")) transposed_coarse_ = std::string(e) != "scatter"; // A/B: per-thread scattered record writes
    if (const char* e = std::getenv("BGE_BP_RECORDS")) full_records_ = std::atoi(e) == 48;  // A/B: BGE_BP_RECORDS=48 keeps full records
    BP_TRY(hipMalloc(&sorted_slot_, std::max<uint64_t>(n_slots, 1) * 4));  // body rank inside its cell
    BP_TRY(hipMalloc(&sorted_aabb_, std::max<uint64_t>(n_slots, 1) * 48));
    BP_TRY(hipMalloc(&body_cell_, std::max<uint64_t>(n_slots, 1) * 4));
    BP_TRY(hipMalloc(&large_list_, std::max<uint64_t>(n_slots, 1) * 4));
    BP_TRY(hipMemset(counters_, 0, sizeof(Accum)));
    compacted_ = false;
    // LDS sort (k_sort_*): coarse buckets of 4096 cells (8192 for tables beyond 16 M cells); larger tables keep the
    // atomic counting sort
    sort_shift_ = 12;
    while (sort_shift_ < 13 && (table_size_ >> sort_shift_) + 2 > kSortMaxBuckets) ++sort_shift_;
    lds_sort_ = (table_size_ >> sort_shift_) + 2 <= kSortMaxBuckets;
    if (const char* e = std::getenv("BGE_BP_SORT")) lds_sort_ = lds_sort_ && std::string(e) != "atomic"; // A/B
    if (lds_sort_) {
        sort_buckets_ = (table_size_ >> sort_shift_) + 2;
        const size_t entries = static_cast<size_t>(sort_buckets_) * kSortGroups + 2 * kScanTile;
        BP_TRY(hipMalloc(&sort_matrix_, entries * 4));
        BP_TRY(hipMalloc(&sort_offsets_, entries * 4));
        BP_TRY(hipMalloc(&sort_status_, (entries / kScanTile + 2) * 8));
        BP_TRY(hipMemset(sort_status_, 0, (entries / kScanTile + 2) * 8));
        BP_TRY(hipMalloc(&coarse_, std::max<uint64_t>(n_slots, 1) * 48));
        // the transposing coarse pass wants 2 x buckets words + a 128 KiB window of LDS: more than the 64 KiB a kernel gets
        // without asking, and more than some devices have
        const size_t base_lds = (2 * static_cast<size_t>(sort_buckets_) + 16 + 4) * 4;
        const size_t lds_c[2] = {base_lds + SortT<false>::kWindow * 48, base_lds + SortT<true>::kWindow * 32};
        int dev = 0, lds_max = 0;
        BP_TRY(hipGetDevice(&dev));
        BP_TRY(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
        // k_sort_fine_t: 4 << shift bytes of counters + as large a window as fits (multiples of 1024 records, at most one pass)
        const size_t fine_counters = (static_cast<size_t>(1) << sort_shift_) * 4;
        bool fits = lds_c[0] <= static_cast<size_t>(lds_max) && lds_c[1] <= static_cast<size_t>(lds_max);
        for (int c = 0; c < 2; ++c) {
            const size_t rec_bytes = c ? 32 : 48, pass = c ? SortT<true>::kPass : SortT<false>::kPass;
            fine_window_[c] = 0;
            if (static_cast<size_t>(lds_max) > fine_counters + 1024) {
                fine_window_[c] = static_cast<uint32_t>(std::min<size_t>(pass, ((static_cast<size_t>(lds_max) - fine_counters - 1024) / rec_bytes) / 1024 * 1024));
            }
            fits = fits && fine_window_[c] != 0;
        }
        auto ask = [](const void* fn, size_t bytes) { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)) == hipSuccess; };
        if (!fits || !ask(reinterpret_cast<const void*>(&k_sort_coarse_t<true>), lds_c[1]) ||
            !ask(reinterpret_cast<const void*>(&k_sort_coarse_t<false>), lds_c[0]) ||
            !ask(reinterpret_cast<const void*>(&k_sort_fine_t<true>), fine_counters + static_cast<size_t>(fine_window_[1]) * 32) ||
            !ask(reinterpret_cast<const void*>(&k_sort_fine_t<false>), fine_counters + static_cast<size_t>(fine_window_[0]) * 48)) {
            (void)hipGetLastError();
            transposed_coarse_ = false;
        }
    }
    return BGE_OK;
}

int Broadphase::run(hipStream_t stream, const WorldView& w, uint64_t n, const uint32_t* entity_of_slot, const PairWindow* window,
                    const FilterPalette* palette, const float4* wave_partials)
{
    const bool compact_records = palette && palette->class_of_slot && palette->table && !window && !full_records_;
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
    const uint64_t shard_cap = shard_capacity(capacity_);
    const PairSink sink{&acc->shard_count[0][0], static_cast<uint2*>(scan_stage_), shard_cap,
                        window ? window->axis : 0u, window ? window->lo : -INFINITY, window ? window->hi : INFINITY};
    ran_ = true;
    last_compact_ = compact_records;
    if (n == 0) {
        BP_TRY(hipMemsetAsync(counters_, 0, sizeof(Accum), stream));
        return BGE_OK;
    }
    // the scan covers table_size_ + 2 entries (cell_start[c + 2] is read for the last cell)
    const uint32_t scan_n = table_size_ + 2;
    const uint32_t scan_blocks = blocks_for(scan_n, kScanBlock);
    const uint32_t slot_blocks = blocks_for(n, 256);

    uint32_t bounds_blocks = std::min<uint32_t>(slot_blocks, kBoundsBlocks);
    if (wave_partials && fused_bounds_) {
        // the tick kernel left one partial per wave beside the AABBs: 32 bytes per 64 slots instead of 28 per slot
        const uint32_t n_partials = static_cast<uint32_t>(n / 64);
        bounds_blocks = std::min<uint32_t>(kReduceBlocks, blocks_for(n_partials, 256));
        hipLaunchKernelGGL(k_bp_reduce_partials, dim3(bounds_blocks), dim3(256), 0, stream, wave_partials, n_partials, acc, table_size_,
                           fused_params_ ? 1u : 0u);
        if (!fused_params_) hipLaunchKernelGGL(k_bp_params, dim3(1), dim3(kParamsThreads), 0, stream, acc, table_size_, bounds_blocks);
    } else {
        hipLaunchKernelGGL(k_bp_bounds, dim3(bounds_blocks), dim3(256), 0, stream, n, w.flags, w.aabb, acc);
        hipLaunchKernelGGL(k_bp_params, dim3(1), dim3(kParamsThreads), 0, stream, acc, table_size_, bounds_blocks);
    }
    if (lds_sort_) {
        const uint32_t groups = std::min<uint32_t>(sort_groups_, blocks_for(n, kSortThreads));
        const uint64_t chunk = (n + groups - 1) / groups;
        uint32_t* matrix = static_cast<uint32_t*>(sort_matrix_);
        uint32_t* offsets = static_cast<uint32_t*>(sort_offsets_);
        float4* coarse = static_cast<float4*>(coarse_);
        const uint32_t n_scan = sort_buckets_ * groups;
        hipLaunchKernelGGL(k_sort_hist, dim3(groups), dim3(kSortThreads), 0, stream, n, chunk, w.flags, w.aabb, acc, sort_shift_,
                           sort_buckets_, matrix, large_list);
        scan_epoch_ = (scan_epoch_ + 1u) & 0x3fffffffu;
        if (scan_epoch_ == 0u) scan_epoch_ = 1u;
        hipLaunchKernelGGL(k_scan_lookback, dim3(blocks_for(n_scan, kScanTile)), dim3(256), 0, stream, matrix, offsets,
                           static_cast<unsigned long long*>(sort_status_), acc, n_scan, scan_epoch_, 0u);
        const size_t fine_lds = (static_cast<size_t>(1) << sort_shift_) * 4;
        if (transposed_coarse_) {
            const uint32_t* no_u32 = nullptr;
            const size_t base_lds = (2 * static_cast<size_t>(sort_buckets_) + 16 + 4) * 4;
            if (compact_records) {
                hipLaunchKernelGGL(k_sort_coarse_t<true>, dim3(groups), dim3(kSortThreads), base_lds + SortT<true>::kWindow * 32, stream, n, chunk,
                                   w.flags, w.aabb, acc, sort_shift_, sort_buckets_, offsets, no_u32, no_u32, palette->class_of_slot,
                                   entity_of_slot, coarse);
                hipLaunchKernelGGL(k_sort_fine_t<true>, dim3(sort_buckets_), dim3(kSortThreads), fine_lds + static_cast<size_t>(fine_window_[1]) * 32,
                                   stream, acc, sort_shift_, sort_buckets_, groups, offsets, coarse, sorted, cell_start, fine_window_[1]);
            } else {
                hipLaunchKernelGGL(k_sort_coarse_t<false>, dim3(groups), dim3(kSortThreads), base_lds + SortT<false>::kWindow * 48, stream, n, chunk,
                                   w.flags, w.aabb, acc, sort_shift_, sort_buckets_, offsets, w.group, w.mask, no_u32, entity_of_slot, coarse);
                hipLaunchKernelGGL(k_sort_fine_t<false>, dim3(sort_buckets_), dim3(kSortThreads), fine_lds + static_cast<size_t>(fine_window_[0]) * 48,
                                   stream, acc, sort_shift_, sort_buckets_, groups, offsets, coarse, sorted, cell_start, fine_window_[0]);
            }
        } else if (compact_records) {
            hipLaunchKernelGGL(k_sort_coarse<true>, dim3(groups), dim3(kSortThreads), 0, stream, n, chunk, w.flags, w.aabb, acc, sort_shift_,
                               sort_buckets_, offsets, w.group, w.mask, palette->class_of_slot, entity_of_slot, coarse);
            hipLaunchKernelGGL(k_sort_fine<true>, dim3(sort_buckets_), dim3(kFineThreads), fine_lds, stream, acc, sort_shift_, sort_buckets_, groups,
                               offsets, coarse, sorted, cell_start);
        } else {
            hipLaunchKernelGGL(k_sort_coarse<false>, dim3(groups), dim3(kSortThreads), 0, stream, n, chunk, w.flags, w.aabb, acc, sort_shift_,
                               sort_buckets_, offsets, w.group, w.mask, static_cast<const uint32_t*>(nullptr), entity_of_slot, coarse);
            hipLaunchKernelGGL(k_sort_fine<false>, dim3(sort_buckets_), dim3(kFineThreads), fine_lds, stream, acc, sort_shift_, sort_buckets_, groups,
                               offsets, coarse, sorted, cell_start);
        }
    } else {
        BP_TRY(hipMemsetAsync(cell_count, 0, (static_cast<size_t>(table_size_) + 2 * kScanTile) * 4, stream));
        hipLaunchKernelGGL(k_bp_count, dim3(slot_blocks), dim3(256), 0, stream, n, w.flags, w.aabb, acc, cell_count, body_cell,
                           body_rank, large_list);
        if (three_kernel_scan_) {
            hipLaunchKernelGGL(k_scan_blocks, dim3(scan_blocks), dim3(256), 0, stream, cell_count, cell_start, block_sums, scan_n);
            hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, stream, block_sums, scan_blocks);
            hipLaunchKernelGGL(k_scan_add, dim3(scan_blocks), dim3(256), 0, stream, cell_start, block_sums, scan_n);
        } else {
            scan_epoch_ = (scan_epoch_ + 1u) & 0x3fffffffu;
            if (scan_epoch_ == 0u) scan_epoch_ = 1u; // 0 is what a never-written status word holds
            hipLaunchKernelGGL(k_scan_lookback, dim3(blocks_for(scan_n, kScanTile)), dim3(256), 0, stream, cell_count, cell_start,
                               static_cast<unsigned long long*>(scan_status_), acc, scan_n, scan_epoch_, 1u);
        }
        if (compact_records) {
            hipLaunchKernelGGL(k_bp_scatter<true>, dim3(slot_blocks), dim3(256), 0, stream, n, w.flags, w.aabb, cell_start, body_cell,
                               body_rank, w.group, w.mask, palette->class_of_slot, entity_of_slot, sorted);
        } else {
            hipLaunchKernelGGL(k_bp_scatter<false>, dim3(slot_blocks), dim3(256), 0, stream, n, w.flags, w.aabb, cell_start, body_cell,
                               body_rank, w.group, w.mask, static_cast<const uint32_t*>(nullptr), entity_of_slot, sorted);
        }
        // persistent grid sized to residency: 18 KiB of LDS per workgroup -> 8 per CU on 256 CUs
    }
    const dim3 pair_grid(std::min<uint32_t>(slot_blocks, 8 * 256));
    const uint4* no_table = nullptr;
    if (!block_pairs_) {
        // wave-granular search, persistent grid sized to residency (LDS: 26 KiB per workgroup with 32-byte records, 30 KiB
        // with full ones).  A software-pipelined variant (all five rows' cell_start loads up front, the next row's chunk
        // prefetched into registers during the tests) was measured SLOWER: 324 us against 308 us — it needs 95 VGPRs.
        const bool small = compact_records && !window && palette->n_classes <= 32u && small_palette_;
        const dim3 wgrid(std::min<uint32_t>(blocks_for(n, 256), (small ? kWaveResidentSmall : (compact_records ? kWaveResidentCompact : kWaveResidentFull)) * 256));
        if (window) {
            hipLaunchKernelGGL((k_bp_pairs_wave<true, false>), wgrid, dim3(256), 0, stream, acc, cell_start, sorted, no_table, sink);
        } else if (compact_records && palette->n_classes <= 32u && small_palette_) {
            hipLaunchKernelGGL((k_bp_pairs_wave<false, true, true>), wgrid, dim3(256), 0, stream, acc, cell_start, sorted, palette->table, sink);
        } else if (compact_records) {
            hipLaunchKernelGGL((k_bp_pairs_wave<false, true>), wgrid, dim3(256), 0, stream, acc, cell_start, sorted, palette->table, sink);
        } else {
            hipLaunchKernelGGL((k_bp_pairs_wave<false, false>), wgrid, dim3(256), 0, stream, acc, cell_start, sorted, no_table, sink);
        }
    } else if (window) {
        hipLaunchKernelGGL((k_bp_pairs<true, false>), pair_grid, dim3(256), 0, stream, acc, cell_start, sorted, no_table, sink);
    } else if (compact_records) {
        hipLaunchKernelGGL((k_bp_pairs<false, true>), pair_grid, dim3(256), 0, stream, acc, cell_start, sorted, palette->table, sink);
    } else {
        hipLaunchKernelGGL((k_bp_pairs<false, false>), pair_grid, dim3(256), 0, stream, acc, cell_start, sorted, no_table, sink);
    }
    hipLaunchKernelGGL(k_bp_large, dim3(std::min<uint32_t>(slot_blocks, 1024)), dim3(256), 0, stream, n, acc, large_list,
                       body_cell, w.aabb, w.flags, w.group, w.mask, entity_of_slot, sink);
    // The pairs now sit in 64 shard slices; the compact list is built on demand (compact()): a tick whose pairs nobody
    // downloads does not pay the 49 us copy (4 M bodies, 12.6 M pairs).
    compacted_ = false;
    BP_TRY(hipGetLastError());
    return BGE_OK;
}

int Broadphase::query_boxes(hipStream_t stream, const WorldView& w, const uint32_t* entity_of_slot, const FilterPalette* palette,
                            const BoxQuery& q)
{
    if (!ran_) {
        error_ = "query_boxes before run";
        return BGE_ERR_STATE;
    }
    if (q.n_boxes == 0) return BGE_OK;
    const Accum* acc = static_cast<const Accum*>(counters_);
    hipLaunchKernelGGL(k_bp_classify_boxes, dim3(blocks_for(q.n_boxes, 256)), dim3(256), 0, stream, acc, q.n_boxes, q.aabb, q.counters,
                       q.big_list, q.grid_list);
    const dim3 grid(std::min<uint32_t>(blocks_for(q.n_boxes, 4), 2048u));
    const uint32_t* cell_start = static_cast<const uint32_t*>(cell_start_);
    const float4* sorted = static_cast<const float4*>(sorted_aabb_);
    const uint32_t* large_list = static_cast<const uint32_t*>(large_list_);
    if (last_compact_) {
        hipLaunchKernelGGL(k_bp_query_boxes<true>, grid, dim3(256), 0, stream, acc, cell_start, sorted, palette->table, q.aabb, q.group, q.mask,
                           q.entity, q.grid_list, q.counters, large_list, w.aabb, w.flags, w.group, w.mask, entity_of_slot, q.out, q.cap);
    } else {
        hipLaunchKernelGGL(k_bp_query_boxes<false>, grid, dim3(256), 0, stream, acc, cell_start, sorted, static_cast<const uint4*>(nullptr),
                           q.aabb, q.group, q.mask, q.entity, q.grid_list, q.counters, large_list, w.aabb, w.flags, w.group, w.mask,
                           entity_of_slot, q.out, q.cap);
    }
    BP_TRY(hipGetLastError());
    return BGE_OK;
}

PairSlices Broadphase::slices() const
{
    const Accum* acc = static_cast<const Accum*>(counters_);
    return PairSlices{static_cast<const uint2*>(scan_stage_), &acc->shard_count[0][0], shard_capacity(capacity_), kShards};
}

int Broadphase::compact(hipStream_t stream)
{
    if (!ran_ || compacted_) return BGE_OK;
    hipLaunchKernelGGL(k_bp_compact, dim3(kShards * kCompactParts), dim3(256), 0, stream, static_cast<Accum*>(counters_),
                       static_cast<const uint2*>(scan_stage_), shard_capacity(capacity_), static_cast<uint2*>(pairs_), capacity_);
    BP_TRY(hipGetLastError());
    compacted_ = true;
    return BGE_OK;
}

int Broadphase::download(hipStream_t stream, uint32_t* pairs2, uint64_t cap, uint64_t* total)
{
    *total = 0;
    if (!ran_) return BGE_OK;
    const Accum* acc = static_cast<const Accum*>(counters_);
    unsigned long long counts[kShards][8];
    uint32_t scan_error = 0;
    BP_TRY(hipMemcpyAsync(counts, acc->shard_count, sizeof counts, hipMemcpyDeviceToHost, stream));
    BP_TRY(hipMemcpyAsync(&scan_error, &acc->scan_error, 4, hipMemcpyDeviceToHost, stream));
    BP_TRY(hipStreamSynchronize(stream));
    if (scan_error) {
        error_ = "internal error: the cell scan's look-back timed out";
        return BGE_ERR_HIP;
    }
    const uint64_t shard_cap = shard_capacity(capacity_);
    unsigned long long both[2] = {0, 0}; // found on the device; kept in the shard slices (and hence in the compact list)
    for (uint32_t s = 0; s < kShards; ++s) {
        both[0] += counts[s][0];
        both[1] += std::min<unsigned long long>(counts[s][0], shard_cap);
    }
    both[1] = std::min<unsigned long long>(both[1], capacity_);
    *total = both[0];
    if (pairs2 && both[1] < both[0]) {
        error_ = std::to_string(both[0]) + " pairs found but the device kept " + std::to_string(both[1]) +
                 " (pair_capacity " + std::to_string(capacity_) + ", split over " + std::to_string(kShards) +
                 " emission shards): create the world with a larger pair_capacity";
        return BGE_ERR_INVALID;
    }
    const uint64_t take = std::min<uint64_t>(std::min<uint64_t>(both[1], cap), capacity_);
    if (take && pairs2) {
        if (int rc = compact(stream)) return rc;
        BP_TRY(hipMemcpyAsync(pairs2, pairs_, take * 8, hipMemcpyDeviceToHost, stream));
        BP_TRY(hipStreamSynchronize(stream));
    }
    return BGE_OK;
}

} // namespace bge
