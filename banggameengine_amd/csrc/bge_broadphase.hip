// bge_broadphase.hip — placeholder until the grid broadphase lands (next commit).
#include "bge_broadphase.hpp"

#include "../../include/bge_world.h"

namespace bge {
int Broadphase::fail(int code, const char* what, hipError_t e)
{
    error_ = std::string(what) + ": " + hipGetErrorString(e);
    return code;
}
int Broadphase::configure(uint64_t n_slots, uint64_t pair_capacity)
{
    n_slots_ = n_slots;
    capacity_ = pair_capacity;
    return BGE_OK;
}
int Broadphase::run(hipStream_t, const WorldView&, uint64_t, const uint32_t*)
{
    error_ = "broadphase not built yet";
    return BGE_ERR_UNSUPPORTED;
}
int Broadphase::download(hipStream_t, uint32_t*, uint64_t, uint64_t*)
{
    error_ = "broadphase not built yet";
    return BGE_ERR_UNSUPPORTED;
}
void Broadphase::release() {}
} // namespace bge
