// bge_comm.hpp — per-frame all-gather of root world matrices over RCCL (see bge_comm.cpp).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>

namespace bge {

class RootComm {
public:
    static int unique_id(void* out128, std::string& err);
    int init(int nranks, int rank, const void* id128, uint64_t rows_per_rank);
    bool ready() const { return comm_ != nullptr; }
    uint64_t rows_per_rank() const { return rows_; }
    int nranks() const { return nranks_; }
    const void* last_table() const { return frame_ ? table_[(frame_ - 1) % kRing] : nullptr; }
    // Buffer ring: frame t uses pair t % kRing.  The compute stream waits on the side stream only every kWaitEvery
    // frames (on the gather of frame t - kWaitEvery); a pair is therefore reused no earlier than kRing - kWaitEvery + 1
    // frames after a gather that is known to be complete.  A cross-stream wait costs ~4 us of dispatch gap on the
    // compute stream (measured), a tick of 2 M entities ~38 us.
    // A root's row on the wire: 12 floats — its world matrix without the fourth column, which is exactly (0, 0, 0, 1)
    // for a root (world = local = bx::mtxSRT).  25 % fewer bytes over xGMI than the 4x4.
    static constexpr uint32_t kRowFloats = 12;
    static constexpr int kRing = 8;
    static constexpr int kWaitEvery = 4;
    // Returns the send buffer of this frame after making `compute` wait until the gather that last read it is done.
    int begin_frame(hipStream_t compute, float** send);
    // Enqueue the all-gather of this frame's send buffer on the side stream (after everything queued on `compute`).
    int gather(hipStream_t compute, void** table_device);
    int wait(hipStream_t compute);
    int rank() const { return rank_; }
    // How a frame's root table is gathered: 0 = one ncclAllGather; 1 = DIRECT: one ncclSend + ncclRecv per peer inside one
    // group, i.e. every rank pushes its block over its own xGMI link to each peer (the node is a full mesh, so no hop
    // relays another rank's data).  Which one is faster depends on the RCCL build and the message size: bench.py times
    // both on the node it runs on.
    void set_mode(int mode) { mode_ = mode; }
    int mode() const { return mode_; }
    // Small helpers of the sharded broadphase (bge_route.hip), all enqueued on `stream`:
    int all_reduce_max(hipStream_t stream, float* device_values, size_t n);
    int all_reduce_sum_u64(hipStream_t stream, uint64_t* device_values, size_t n);
    int all_gather_bytes(hipStream_t stream, const void* send_device, void* recv_device, size_t bytes_per_rank);
    // Variable-size exchange: counts are in elements of elem_bytes (a multiple of 4); peer p's block starts at the sum of
    // the counts before it.  One ncclSend + ncclRecv per peer inside one group: every pair of ranks talks over its own
    // xGMI link, there is no ring.
    int all_to_all_v(hipStream_t stream, const void* send_device, const uint64_t* send_counts, void* recv_device,
                     const uint64_t* recv_counts, size_t elem_bytes);
    void destroy();
    const char* error() const { return error_.c_str(); }

private:
    int fail(int code, const std::string& what);
    std::string error_;
    void* comm_ = nullptr; // ncclComm_t
    int nranks_ = 0, rank_ = 0;
    uint64_t rows_ = 0;
    uint64_t frame_ = 0;
    hipStream_t side_ = nullptr;
    float* send_[kRing] = {};
    float* table_[kRing] = {};
    hipEvent_t packed_[kRing] = {};   // compute -> side: roots of the frame are packed
    hipEvent_t gathered_[kRing] = {}; // side -> compute: the gather of that frame is complete
    bool in_flight_[kRing] = {};
    int mode_ = 0;
};

} // namespace bge
