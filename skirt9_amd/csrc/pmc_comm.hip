// pmc_comm.hip -- the multi-GPU entry points of include/pmc.h (RCCL)
#include "pmc_context.h"
#include <rccl/rccl.h>

extern "C" {

// ---------------------------------------------------------------------------------------------------
// Several MI355X on one segment: history ranges and the ONE collective per segment over RCCL / xGMI.  The photon
// histories of a segment are independent; every device holds a replica of the scene and runs a static index range.
// What the devices exchange is the sum of their detector arrays onto the root at the end of the segment --
// ProcessManager::sumToRoot behind FluxRecorder::flush (SKIRT/core/FluxRecorder.cpp:487-493,
// SKIRT/mpi/ProcessManager.cpp:223-255) -- and, when the radiation field is stored, the sum of that table onto all
// devices (MediumSystem.cpp:1304-1313).  Both run on the context's stream, behind the segment's kernels.
namespace
{
    int ncclFail(ncclResult_t r, const char* what)
    {
        return fail(PMC_ERR_DEVICE, std::string(what) + ": " + ncclGetErrorString(r));
    }
}

void pmc_history_range(uint64_t num_packets, int32_t rank, int32_t num_ranks, uint64_t* first, uint64_t* count)
{
    if (num_ranks < 1) num_ranks = 1;
    // floor(rank * N / G) without overflow of the product
    const auto cut = [&](uint64_t r) -> uint64_t { return (uint64_t)(((unsigned __int128)r * num_packets) / (uint64_t)num_ranks); };
    const uint64_t a = cut((uint64_t)rank), b = cut((uint64_t)rank + 1);
    if (first) *first = a;
    if (count) *count = b - a;
}

int pmc_comm_init_all(int32_t num_devices, const int32_t* devices, void** comms)
{
    if (num_devices < 1 || !devices || !comms) return fail(PMC_ERR_INVALID, "pmc_comm_init_all: invalid argument");
    static_assert(sizeof(ncclComm_t) == sizeof(void*), "ncclComm_t is a pointer");
    ncclResult_t r = ncclCommInitAll(reinterpret_cast<ncclComm_t*>(comms), num_devices, devices);
    return r == ncclSuccess ? PMC_OK : ncclFail(r, "ncclCommInitAll");
}

int pmc_comm_unique_id(void* unique_id)
{
    static_assert(sizeof(ncclUniqueId) == PMC_COMM_ID_BYTES, "ncclUniqueId size");
    if (!unique_id) return fail(PMC_ERR_INVALID, "pmc_comm_unique_id: null argument");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return ncclFail(r, "ncclGetUniqueId");
    std::memcpy(unique_id, &id, sizeof(id));
    return PMC_OK;
}

int pmc_comm_init_rank(int32_t device, int32_t num_ranks, int32_t rank, const void* unique_id, void** comm)
{
    if (!unique_id || !comm || num_ranks < 1 || rank < 0 || rank >= num_ranks) return fail(PMC_ERR_INVALID, "pmc_comm_init_rank: invalid argument");
    if (hipSetDevice(device) != hipSuccess) return fail(PMC_ERR_DEVICE, "hipSetDevice failed");
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = ncclCommInitRank(reinterpret_cast<ncclComm_t*>(comm), num_ranks, id, rank);
    return r == ncclSuccess ? PMC_OK : ncclFail(r, "ncclCommInitRank");
}

int pmc_comm_size(void* comm, int32_t* num_ranks, int32_t* rank)
{
    if (!comm) return fail(PMC_ERR_INVALID, "pmc_comm_size: null argument");
    int n = 0, me = 0;
    ncclResult_t r = ncclCommCount(reinterpret_cast<ncclComm_t>(comm), &n);
    if (r == ncclSuccess) r = ncclCommUserRank(reinterpret_cast<ncclComm_t>(comm), &me);
    if (r != ncclSuccess) return ncclFail(r, "ncclCommCount");
    if (num_ranks) *num_ranks = n;
    if (rank) *rank = me;
    return PMC_OK;
}

void pmc_comm_destroy(void* comm)
{
    if (comm) ncclCommDestroy(reinterpret_cast<ncclComm_t>(comm));
}

int pmc_reduce_frames(pmc_ctx* ctx, void* comm, int32_t root)
{
    if (!ctx || !comm) return fail(PMC_ERR_INVALID, "pmc_reduce_frames: null argument");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(PMC_ERR_DEVICE, "hipSetDevice failed");
    ncclComm_t c = reinterpret_cast<ncclComm_t>(comm);
    int rank = 0;
    ncclResult_t r = ncclCommUserRank(c, &rank);
    if (r != ncclSuccess) return ncclFail(r, "ncclCommUserRank");
    r = ncclReduce(ctx->frames, ctx->frames, size_t(ctx->frameSize), ncclDouble, ncclSum, root, c, ctx->stream);
    if (r != ncclSuccess) return ncclFail(r, "ncclReduce");
    if (rank != root && hipMemsetAsync(ctx->frames, 0, size_t(ctx->frameSize) * sizeof(double), ctx->stream) != hipSuccess)
        return fail(PMC_ERR_DEVICE, "hipMemsetAsync failed");
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(PMC_ERR_DEVICE, "hipStreamSynchronize failed");
    return PMC_OK;
}

int pmc_allreduce_radiation_field(pmc_ctx* ctx, void* comm)
{
    if (!ctx || !comm) return fail(PMC_ERR_INVALID, "pmc_allreduce_radiation_field: null argument");
    if (!ctx->rfSize) return PMC_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(PMC_ERR_DEVICE, "hipSetDevice failed");
    ncclResult_t r = ncclAllReduce(ctx->dev.rf, ctx->dev.rf, size_t(ctx->rfSize), ncclDouble, ncclSum, reinterpret_cast<ncclComm_t>(comm), ctx->stream);
    if (r != ncclSuccess) return ncclFail(r, "ncclAllReduce");
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(PMC_ERR_DEVICE, "hipStreamSynchronize failed");
    return PMC_OK;
}
}
