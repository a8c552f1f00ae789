// bge_route.hpp — sharded broadphase: routing of body records to spatial slabs (see bge_route.hip).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>

#include "bge_kernels.hpp"

namespace bge {

constexpr uint32_t kMaxSlabs = 64;       // ranks of one broadphase exchange
constexpr uint32_t kMaxHistBins = 4096;
constexpr uint32_t kRecordBytes = 48;    // min.xyz | global id, max.xyz | 0, group | mask | static | 0

class ShardRouter {
public:
    // Step 1: how many records this world sends to each slab.  cuts[0..nranks]: slab s covers [cuts[s], cuts[s+1]) along
    // `axis`; cuts[0] and cuts[nranks] are ignored (treated as -inf / +inf).  counts_host[nranks].
    int count(hipStream_t stream, const WorldView& w, uint64_t n_slots, uint32_t axis, uint32_t nranks, const float* cuts_host,
              uint64_t* counts_host);
    // Step 2: the records, grouped by destination slab (slab d at record offset sum(counts[0..d))), into `send_device`.
    int pack(hipStream_t stream, const WorldView& w, uint64_t n_slots, const uint32_t* global_of_slot, void* send_device);
    // Step 3 (receiving side): records -> the structure-of-arrays view the broadphase kernels read.
    int unpack(hipStream_t stream, const void* records_device, uint64_t n_records, WorldView* view, const uint32_t** entity_ids);
    // min / max corner of all body AABBs (for the common slab cuts); n_bodies may be null
    int bounds(hipStream_t stream, const WorldView& w, uint64_t n_slots, float mn[3], float mx[3], uint64_t* n_bodies);
    // Histogram of the bodies' min corner along `axis` over [lo, hi] in `bins` equal bins (<= kMaxHistBins), for balanced cuts.
    int histogram(hipStream_t stream, const WorldView& w, uint64_t n_slots, uint32_t axis, float lo, float hi, uint32_t bins,
                  uint64_t* hist_host);
    void release();
    const char* error() const { return error_.c_str(); }
    uint64_t total_routed() const { return total_; }

private:
    int fail(int code, const char* what, hipError_t e);
    int ensure(void** p, size_t* have, size_t need);
    std::string error_;
    void* scalars_ = nullptr; // counts[64] | cursor[64] | cuts[65] | bounds[6] | n
    void* hist_ = nullptr;    // unsigned long long[kMaxHistBins]
    uint32_t axis_ = 0, nranks_ = 0;
    uint64_t total_ = 0;
    bool counted_ = false;
    // receive-side arrays
    void *rx_flags_ = nullptr, *rx_aabb_ = nullptr, *rx_group_ = nullptr, *rx_mask_ = nullptr, *rx_entity_ = nullptr;
    size_t rx_flags_b_ = 0, rx_aabb_b_ = 0, rx_group_b_ = 0, rx_mask_b_ = 0, rx_entity_b_ = 0;
};

} // namespace bge
