// bge_comm.cpp — the sharded tick's only collective: ONE ncclAllGather (RCCL over xGMI) of the root world
// matrices per frame, on a side stream, over a ring of buffers so that frame t's gather overlaps the following ticks.
//
// xGMI on an MI355X node is a full mesh of point-to-point links; the gathered message is small (64 B per root:
// 2 MiB per rank for 31,250 subtree roots), so the collective is latency-bound and the point of the side
// stream is to keep that latency off the compute stream.  librccl.so.1 is loaded with dlopen so that the
// library also loads on hosts without RCCL and shares the copy a host process (e.g. PyTorch) already mapped.
#include "bge_comm.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>

#include "../../include/bge_world.h"

namespace bge {

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl& rccl()
{
    static Rccl r = [] {
        Rccl x;
        std::string last;
        // BGE_RCCL_SONAME replaces the search list (deployments with a private RCCL build; the tests point it at a
        // name that does not exist to exercise the BGE_ERR_UNSUPPORTED path)
        const char* forced = std::getenv("BGE_RCCL_SONAME");
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            x.handle = dlopen(forced ? forced : name, RTLD_NOW | RTLD_GLOBAL);
            if (x.handle) break;
            const char* e = dlerror(); // dlerror() clears the message: read it exactly once per failure
            if (e) last = e;
            if (forced) break;
        }
        if (!x.handle) {
            x.why = std::string("cannot load librccl: ") + (last.empty() ? "unknown" : last);
            return x;
        }
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(x.handle, "ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(x.handle, "ncclCommInitRank"));
        x.AllGather = reinterpret_cast<decltype(x.AllGather)>(dlsym(x.handle, "ncclAllGather"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.handle, "ncclCommDestroy"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.handle, "ncclGetErrorString"));
        x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(dlsym(x.handle, "ncclAllReduce"));
        x.Send = reinterpret_cast<decltype(x.Send)>(dlsym(x.handle, "ncclSend"));
        x.Recv = reinterpret_cast<decltype(x.Recv)>(dlsym(x.handle, "ncclRecv"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(x.handle, "ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(x.handle, "ncclGroupEnd"));
        if (!x.GetUniqueId || !x.CommInitRank || !x.AllGather || !x.CommDestroy || !x.GetErrorString || !x.AllReduce || !x.Send ||
            !x.Recv || !x.GroupStart || !x.GroupEnd) {
            x.why = "librccl is missing an expected symbol";
            x.handle = nullptr;
        }
        return x;
    }();
    return r;
}

} // namespace

int RootComm::fail(int code, const std::string& what)
{
    error_ = what;
    return code;
}

#define COMM_HIP(expr)                                                                         \
    do {                                                                                       \
        const hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) return fail(BGE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define COMM_NCCL(expr)                                                                        \
    do {                                                                                       \
        const ncclResult_t r_ = (expr);                                                        \
        if (r_ != ncclSuccess) return fail(BGE_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(r_)); \
    } while (0)

int RootComm::unique_id(void* out128, std::string& err)
{
    Rccl& r = rccl();
    if (!r.handle) {
        err = r.why;
        return BGE_ERR_UNSUPPORTED;
    }
    ncclUniqueId id;
    const ncclResult_t rc = r.GetUniqueId(&id);
    if (rc != ncclSuccess) {
        err = std::string("ncclGetUniqueId: ") + r.GetErrorString(rc);
        return BGE_ERR_HIP;
    }
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(out128, &id, sizeof id);
    return BGE_OK;
}

int RootComm::init(int nranks, int rank, const void* id128, uint64_t rows_per_rank)
{
    Rccl& r = rccl();
    if (!r.handle) return fail(BGE_ERR_UNSUPPORTED, r.why);
    destroy();
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    COMM_NCCL(r.CommInitRank(&comm, nranks, id, rank));
    comm_ = comm;
    nranks_ = nranks;
    rank_ = rank;
    rows_ = rows_per_rank ? rows_per_rank : 1;
    frame_ = 0;
    COMM_HIP(hipStreamCreateWithFlags(&side_, hipStreamNonBlocking));
    for (int b = 0; b < kRing; ++b) {
        COMM_HIP(hipMalloc(reinterpret_cast<void**>(&send_[b]), rows_ * kRowFloats * 4));
        COMM_HIP(hipMalloc(reinterpret_cast<void**>(&table_[b]), rows_ * kRowFloats * 4 * static_cast<size_t>(nranks)));
        COMM_HIP(hipMemset(send_[b], 0, rows_ * kRowFloats * 4));
        COMM_HIP(hipEventCreateWithFlags(&packed_[b], hipEventDisableTiming));
        COMM_HIP(hipEventCreateWithFlags(&gathered_[b], hipEventDisableTiming));
        in_flight_[b] = false;
    }
    return BGE_OK;
}

int RootComm::begin_frame(hipStream_t compute, float** send)
{
    const int b = static_cast<int>(frame_ % kRing);
    if (frame_ >= static_cast<uint64_t>(kWaitEvery) && frame_ % kWaitEvery == 0) {
        // everything up to the gather of frame (t - kWaitEvery) is complete once this wait passes; the pairs used by
        // frames t .. t + kWaitEvery - 1 were last read by gathers of frames <= t - (kRing - kWaitEvery + 1)
        const int done = static_cast<int>((frame_ - kWaitEvery) % kRing);
        if (in_flight_[done]) COMM_HIP(hipStreamWaitEvent(compute, gathered_[done], 0));
    }
    *send = send_[b];
    return BGE_OK;
}

int RootComm::gather(hipStream_t compute, void** table_device)
{
    const int b = static_cast<int>(frame_ % kRing);
    COMM_HIP(hipEventRecord(packed_[b], compute));
    COMM_HIP(hipStreamWaitEvent(side_, packed_[b], 0));
    if (mode_ == 1) {
        // direct schedule over the xGMI mesh: one send and one receive per peer, all inside one group
        const size_t words = rows_ * kRowFloats;
        COMM_NCCL(rccl().GroupStart());
        ncclResult_t first_error = ncclSuccess;
        for (int p = 0; p < nranks_; ++p) {
            ncclResult_t e = rccl().Send(send_[b], words, ncclFloat32, p, static_cast<ncclComm_t>(comm_), side_);
            if (e != ncclSuccess && first_error == ncclSuccess) first_error = e;
            e = rccl().Recv(table_[b] + static_cast<size_t>(p) * words, words, ncclFloat32, p, static_cast<ncclComm_t>(comm_), side_);
            if (e != ncclSuccess && first_error == ncclSuccess) first_error = e;
        }
        COMM_NCCL(rccl().GroupEnd());
        if (first_error != ncclSuccess) return fail(BGE_ERR_HIP, std::string("ncclSend/ncclRecv: ") + rccl().GetErrorString(first_error));
    } else {
        COMM_NCCL(rccl().AllGather(send_[b], table_[b], rows_ * kRowFloats, ncclFloat32, static_cast<ncclComm_t>(comm_), side_));
    }
    COMM_HIP(hipEventRecord(gathered_[b], side_));
    in_flight_[b] = true;
    if (table_device) *table_device = table_[b];
    ++frame_;
    return BGE_OK;
}

int RootComm::wait(hipStream_t compute)
{
    for (int b = 0; b < kRing; ++b) {
        if (in_flight_[b]) COMM_HIP(hipStreamWaitEvent(compute, gathered_[b], 0));
    }
    return BGE_OK;
}

int RootComm::all_reduce_max(hipStream_t stream, float* device_values, size_t n)
{
    if (!comm_) return fail(BGE_ERR_STATE, "communicator not initialised");
    COMM_NCCL(rccl().AllReduce(device_values, device_values, n, ncclFloat32, ncclMax, static_cast<ncclComm_t>(comm_), stream));
    return BGE_OK;
}

int RootComm::all_reduce_sum_u64(hipStream_t stream, uint64_t* device_values, size_t n)
{
    if (!comm_) return fail(BGE_ERR_STATE, "communicator not initialised");
    COMM_NCCL(rccl().AllReduce(device_values, device_values, n, ncclUint64, ncclSum, static_cast<ncclComm_t>(comm_), stream));
    return BGE_OK;
}

int RootComm::all_gather_bytes(hipStream_t stream, const void* send_device, void* recv_device, size_t bytes_per_rank)
{
    if (!comm_) return fail(BGE_ERR_STATE, "communicator not initialised");
    COMM_NCCL(rccl().AllGather(send_device, recv_device, bytes_per_rank, ncclUint8, static_cast<ncclComm_t>(comm_), stream));
    return BGE_OK;
}

int RootComm::all_to_all_v(hipStream_t stream, const void* send_device, const uint64_t* send_counts, void* recv_device,
                           const uint64_t* recv_counts, size_t elem_bytes)
{
    if (!comm_) return fail(BGE_ERR_STATE, "communicator not initialised");
    if (elem_bytes % 4) return fail(BGE_ERR_INVALID, "element size must be a multiple of 4 bytes");
    const size_t words = elem_bytes / 4;
    const char* s = static_cast<const char*>(send_device);
    char* r = static_cast<char*>(recv_device);
    COMM_NCCL(rccl().GroupStart());
    size_t so = 0, ro = 0;
    ncclResult_t first_error = ncclSuccess;
    for (int p = 0; p < nranks_; ++p) {
        if (send_counts[p]) {
            const ncclResult_t e = rccl().Send(s + so * elem_bytes, send_counts[p] * words, ncclUint32, p, static_cast<ncclComm_t>(comm_), stream);
            if (e != ncclSuccess && first_error == ncclSuccess) first_error = e;
        }
        if (recv_counts[p]) {
            const ncclResult_t e = rccl().Recv(r + ro * elem_bytes, recv_counts[p] * words, ncclUint32, p, static_cast<ncclComm_t>(comm_), stream);
            if (e != ncclSuccess && first_error == ncclSuccess) first_error = e;
        }
        so += send_counts[p];
        ro += recv_counts[p];
    }
    COMM_NCCL(rccl().GroupEnd()); // always close the group, even after a failed call inside it
    if (first_error != ncclSuccess) return fail(BGE_ERR_HIP, std::string("ncclSend/ncclRecv: ") + rccl().GetErrorString(first_error));
    return BGE_OK;
}

void RootComm::destroy()
{
    if (side_) (void)hipStreamSynchronize(side_);
    if (comm_) (void)rccl().CommDestroy(static_cast<ncclComm_t>(comm_));
    comm_ = nullptr;
    for (int b = 0; b < kRing; ++b) {
        if (send_[b]) (void)hipFree(send_[b]);
        if (table_[b]) (void)hipFree(table_[b]);
        if (packed_[b]) (void)hipEventDestroy(packed_[b]);
        if (gathered_[b]) (void)hipEventDestroy(gathered_[b]);
        send_[b] = table_[b] = nullptr;
        packed_[b] = gathered_[b] = nullptr;
        in_flight_[b] = false;
    }
    if (side_) (void)hipStreamDestroy(side_);
    side_ = nullptr;
    rows_ = 0;
}

} // namespace bge
