// bge_broadphase.hpp — AABB broadphase on the device (see bge_broadphase.hip).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>

#include "bge_kernels.hpp"

namespace bge {

// Slab window of the sharded broadphase: report a pair only where max(min_a[axis], min_b[axis]) is in [lo, hi).
struct PairWindow {
    uint32_t axis;
    float lo, hi;
};

// Collision-filter palette of the world (null pointers: full 48-byte records carrying group and mask).
struct FilterPalette {
    const uint32_t* class_of_slot; // [slots]
    const uint4* table;            // [256] (group, mask, static, 0), unused entries zero
    uint32_t n_classes = 256;      // entries in use
};

// Boxes (trigger ghosts) against the bodies of the last run(): hits are (box index, body entity) appended at
// out[counters[0]++].  Boxes that span too many cells for a walk of the grid are listed in big_list (counters[1] of them)
// for the caller's own pass over all bodies.  A body pairs with a box when it is not Static, is not the box's own entity and
// (box.group & body.mask) && (body.group & box.mask).
struct BoxQuery {
    uint32_t n_boxes;
    const float* aabb;       // [n][6]; min > max: not in the world
    const uint32_t* group;   // [n]
    const uint32_t* mask;    // [n]
    const uint32_t* entity;  // [n]
    uint32_t* counters;      // [3], zeroed by the caller: hits, boxes in big_list, boxes walked through the grid
    uint32_t* big_list;      // [n]
    uint32_t* grid_list;     // [n] scratch
    uint2* out;
    uint32_t cap;
};

// The pair list of the last run() where the search leaves it: 64 slices of `shard_cap` pairs, slice s holding counts[8 * s] of them
// (more found than kept when a count exceeds shard_cap)
struct PairSlices {
    const uint2* stage;
    const unsigned long long* counts;
    uint64_t shard_cap;
    uint32_t shards;
};

class Broadphase {
public:
    PairSlices slices() const;
    // n_slots: upper bound of bodies; pair_capacity: pairs kept per tick
    int configure(uint64_t n_slots, uint64_t pair_capacity);
    // Collect the overlapping pairs of the AABBs the tick kernel just wrote.
    int run(hipStream_t stream, const WorldView& w, uint64_t n_slots_ticked, const uint32_t* entity_of_slot,
            const PairWindow* window = nullptr, const FilterPalette* palette = nullptr, const float4* wave_partials = nullptr);
    int query_boxes(hipStream_t stream, const WorldView& w, const uint32_t* entity_of_slot, const FilterPalette* palette, const BoxQuery& q);
    int download(hipStream_t stream, uint32_t* pairs2, uint64_t cap, uint64_t* total);
    // The search leaves the pairs in 64 shard slices; this builds the compact list (idempotent until the next run).
    int compact(hipStream_t stream);
    void release();
    const char* error() const { return error_.c_str(); }
    void* pairs_device() const { return pairs_; }
    uint64_t capacity() const { return capacity_; }
    uint64_t configured_slots() const { return n_slots_; }

private:
    int fail(int code, const char* what, hipError_t e);
    std::string error_;
    uint64_t n_slots_ = 0;
    uint64_t capacity_ = 0;
    uint32_t table_size_ = 0;
    void* pairs_ = nullptr;      // uint32[capacity][2] compact list
    void* scan_stage_ = nullptr; // sharded staging of the pair list
    void* counters_ = nullptr;   // device scalars (pair count, bounds, ...)
    void* cell_count_ = nullptr; // uint32[table_size + 1]
    void* cell_start_ = nullptr; // uint32[table_size + 1]
    void* scan_tmp_ = nullptr;
    void* scan_status_ = nullptr; // uint64[tiles] status words of the single-pass scan
    uint32_t scan_epoch_ = 0;
    bool three_kernel_scan_ = false;
    bool full_records_ = false;
    // two-level LDS counting sort (k_sort_*)
    bool lds_sort_ = false;
    uint32_t sort_shift_ = 12, sort_buckets_ = 0;
    void *sort_matrix_ = nullptr, *sort_offsets_ = nullptr, *sort_status_ = nullptr, *coarse_ = nullptr;
    bool block_pairs_ = false;
    bool fused_bounds_ = true;      // grid from the tick kernel's per-wave partials (TickParams::bp_partial) when the caller has them; BGE_BP_BOUNDS=pass for A/B
    bool fused_params_ = true;      // the last workgroup of k_bp_reduce_partials chooses the grid; BGE_BP_PARAMS=split launches k_bp_params instead
    uint32_t sort_groups_ = 512;    // chunk workgroups of the coarse passes (BGE_BP_SORT_GROUPS lowers it: tests)
    uint32_t fine_window_[2] = {0, 0}; // records in k_sort_fine_t's LDS window: [0] 48-byte records, [1] 32-byte records
    bool small_palette_ = true;     // wave search: one compatibility word per class when the palette has <= 32 classes
    bool transposed_coarse_ = true; // k_sort_coarse_t (bucket-ordered write-out) instead of k_sort_coarse<true>; BGE_BP_COARSE=scatter for A/B
    void* sorted_slot_ = nullptr; // uint32[n_slots]
    void* sorted_aabb_ = nullptr; // float4[n_slots][3] sorted records
    void* body_cell_ = nullptr;   // int32[n_slots][4]
    void* large_list_ = nullptr;  // uint32[n_slots]
    bool ran_ = false;
    bool last_compact_ = false;   // record format of the last run (32-byte records with a filter class)
    bool compacted_ = false;
};

} // namespace bge
