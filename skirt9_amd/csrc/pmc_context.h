// pmc_context.h -- what the translation units of the host side of libpmc.so share: the context behind a pmc_ctx handle, the error text of the
// calling thread, the launchers of pmc_kernels.hip.  pmc_api.hip holds the ABI entry points (scene upload: pmc_create), pmc_tables.hip the
// table builders (octree flattening, Voronoi run / cone / observer tables), pmc_run.hip the generation loop (pmc_run_primary), pmc_comm.hip the
// RCCL calls, pmc_tuning.hip the switch table of include/pmc_tuning.h.
#ifndef PMC_CONTEXT_H
#define PMC_CONTEXT_H

#include "pmc_device.h"
#include "../../include/pmc_layout.h"
#include "../../include/pmc_tuning.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <rccl/rccl.h>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <map>
#include <vector>

extern "C" const char* pmcTune(const char* name);
extern "C" hipError_t pmcUploadScene(int slot, const DevScene* scene, hipStream_t stream);
extern "C" hipError_t pmcConfigureKernels(size_t walkLds, size_t transitionLds);
extern "C" int pmcWalkBlocksPerCU(int gridKind, int kind, int wide, int block, size_t ldsBytes);
extern "C" hipError_t pmcLaunchStatMerge(int slot, int blocks, hipStream_t stream);
extern "C" int pmcPeelBlock(void);
extern "C" int pmcPropBlock(void);
extern "C" hipError_t pmcLaunchWalk(int slot, int gridKind, int storeRf, int taskBase, int numTaskRecords, int taskCounter, uint64_t seed, int grid,
                                    int block, size_t ldsBytes, const WalkStreamArgs* tasks, hipStream_t stream);
extern "C" hipError_t pmcLaunchPeel(int slot, int wide, int slotBase, int numSlots, const int* list, int cursor, int obs, int sgn, int grid, size_t ldsBytes,
                                    const PeelRec* sortedRec, const unsigned long long* sortedCount, unsigned long long* xcdCursor, hipStream_t stream);
extern "C" int pmcVoroPropWavesPerSimd(void);
extern "C" hipError_t pmcLaunchVoroProp(int slot, const int32_t* list, const unsigned long long* count, unsigned long long* xcdCursor, int segments, uint64_t seed,
                                        int flavour, int grid, hipStream_t stream);
extern "C" int pmcVoroPeelWavesPerSimd(void);
extern "C" hipError_t pmcLaunchVoroPeel(int slot, int rec, int tab, const int32_t* list, const unsigned long long* count, unsigned long long* xcdCursor, int segments,
                                        int severalMedia, int grid, hipStream_t stream);
extern "C" hipError_t pmcLaunchProp(int slot, int wide, int storeRf, int slotBase, int numSlots, const int* list, int cursor, uint64_t seed, int grid,
                                    size_t ldsBytes, const RfLogArgs* rfLog, hipStream_t stream);
extern "C" hipError_t pmcLaunchRfFlush(int slot, const uint32_t* keys, const double* vals, uint32_t* sortedKeys, double* sortedVals, unsigned long long n,
                                       int numParts, void* temp, int numCU, hipStream_t stream);
extern "C" int pmcPeelHasQueues(int wide, size_t ldsBytes);
extern "C" int pmcExperimentBuild(void);
extern "C" const unsigned long long* pmcPeelSortedCount(void* temp);
extern "C" size_t pmcPeelSortTempBytes();
extern "C" hipError_t pmcLaunchPeelSortCounts(int slot, int slotBase, int numSlots, PeelSortArgs* ps, PeelRec* const* sorted, int32_t* const* lists, void* const* temp,
                                              int* groups, hipStream_t stream);
extern "C" size_t pmcRfTempBytes(int numParts);
extern "C" int pmcRfMaxParts();
extern "C" hipError_t pmcLaunchTransition(int slot, int slotBase, int numSlots, int group, uint64_t seed, const int* list, int listLen, int maxBlocks,
                                          size_t ldsBytes, const StatLogArgs* statLog, uint64_t count, uint64_t keep, hipStream_t stream);
extern "C" hipError_t pmcLaunchLaunch(int slot, int slotBase, int numSlots, int group, uint64_t first, uint64_t count, uint64_t seed, int initial,
                                      int maxBlocks, size_t ldsBytes, const StatLogArgs* statLog, hipStream_t stream);
extern "C" hipError_t pmcLaunchStatFlush(int slot, const uint32_t* keys, const double* vals, uint32_t* sortedKeys, double* sortedVals, unsigned long long n,
                                         int numParts, void* temp, int numCU, const uint32_t* chunkFill, hipStream_t stream);
extern "C" int pmcStatBucketBits();
extern "C" hipError_t pmcLaunchCycleStart(int slot, int gridKind, int slotBase, int numSlots, int listCounter, int* listOut, const int* listIn,
                                          int listLen, int maxBlocks, size_t ldsBytes, const PeelSortArgs* sort, hipStream_t stream);
extern "C" hipError_t pmcLaunchTrace(int slot, int gridKind, int wide, int uniform, const double r[3], const double k[3],
                                     const double* kdev, int32_t* m, double* ds, int32_t cap, int32_t* n, size_t ldsBytes,
                                     hipStream_t stream);

// error text of the calling thread (pmc_last_error); defined in pmc_api.hip
void pmcSetError(const std::string& message);
namespace
{
    inline int fail(int code, const std::string& message)
    {
        pmcSetError(message);
        return code;
    }
    inline int hipFail(hipError_t e, const char* what)
    {
        return fail(PMC_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
    }
}

#define HIP_TRY(call)                                           \
    do                                                          \
    {                                                           \
        hipError_t e_ = (call);                                 \
        if (e_ != hipSuccess) return hipFail(e_, #call);        \
    } while (0)

struct pmc_ctx
{
    int device{0};
    int slot{-1};
    bool sceneDirty{true};
    hipStream_t stream{nullptr};
    // slot groups: the generations of group g are enqueued on groupStream[g] (group 0 uses `stream`)
    int numGroups{3};
    hipStream_t groupStream[PMC_MAX_GROUPS]{};
    // octree: the peel-off kernels of a generation run on a side stream of the group, next to its propagation kernel
    hipStream_t peelStream[PMC_MAX_GROUPS]{};
    hipEvent_t evA[PMC_MAX_GROUPS]{}, evB[PMC_MAX_GROUPS]{}, evC[PMC_MAX_GROUPS]{}, evJoin[PMC_MAX_GROUPS]{}, evProp[PMC_MAX_GROUPS]{};
    hipEvent_t evStart{nullptr}, evStop{nullptr};
    bool timed{false};
    float totalMs{0}, walkMs{0}, transitionMs{0};
    float peelMs{0}, propMs{0};  // octree: the spans of the peel-off kernels and of the propagation kernel, summed over the generations
    int generations{0};
    DevScene dev{};
    std::vector<void*> allocations;
    std::vector<void*> slotAllocations;
    double* frames{nullptr};
    int64_t frameSize{0};
    int64_t rfSize{0};  // doubles of the radiation field table (0: not stored)
    size_t walkLds{0}, transitionLds{0}, launchLds{0};
    int block{256};
    int grid{0};          // workgroups of the generic walk kernel / the octree propagation kernel
    int peelGrid{0};      // workgroups of an octree peel-off kernel
    int wide{0};          // octree deeper than level 10: 21-bit index fields (pmc_walk_tree.inc Pack)
    int numCU{256};
    int64_t numSlots{0};         // requested pool size
    int64_t allocatedSlots{0};   // size of the allocated slot arrays
    unsigned long long* pinned{nullptr};
    unsigned long long internalErrorsSeen{0};
    pmc_progress_fn progress{nullptr};    // pmc_set_progress
    void* progressUser{nullptr};
    double progressInterval{3.};
    size_t steppedDownFree{0};            // free device memory when the default pool last stepped down (0: it has not)
    bool slotsConfigured{false};          // the number of slots was set explicitly (PMC_NUM_SLOTS, pmc_set_num_slots)
    bool groupsConfigured{false};         // the number of slot groups was set explicitly (PMC_NUM_GROUPS)
    unsigned long long overflowsSeen{0};  // statistics-list overflows already reported (pmc_run_primary)
    int32_t* statPoolIota{nullptr};       // 0, 1, 2, ...: the free list of a statistics pool none of whose blocks is in use
    int64_t statPoolBlocks{0};
    int statPoolGrowths{0};               // times the pool has grown (pmc_run_primary)
    // radiation field on an octree: per slot group the log of a generation's contributions (two buffers each for the
    // partitioning sort) and the sort's temporary storage
    std::vector<void*> rfAllocations;
    uint32_t* rfKeys[PMC_MAX_GROUPS][2]{};
    double* rfVals[PMC_MAX_GROUPS][2]{};
    unsigned long long rfCap[PMC_MAX_GROUPS]{};
    void* rfTemp[PMC_MAX_GROUPS]{};
    // sorted peel-off records (pmc_device.h PeelRec): per group the records in slot order and in tile order, the keys, the sort's counters
    PeelRec* peelRec[PMC_MAX_GROUPS][PMC_SORT_OBS]{};  // per group and sorted observer (octree)
    int32_t* peelList[PMC_MAX_GROUPS][PMC_SORT_OBS]{};  // (Cartesian, Voronoi) the slots in tile order instead
    void* peelTemp[PMC_MAX_GROUPS][PMC_SORT_OBS]{};
    unsigned long long* xcdCursors{nullptr};  // [PMC_MAX_GROUPS][PMC_SORT_OBS + 1][8] (+ 8 that stay zero) the walk kernels' cursors over the eighths of their sorted records / lists
    int peelCap[PMC_MAX_GROUPS]{};
    size_t rfTempBytes{0};
    // statistics log per slot group (pmc_device.h StatLogArgs): the log and its partitioned copy, the sort's counters
    uint32_t* statKeys[PMC_MAX_GROUPS][2]{};
    double* statVals[PMC_MAX_GROUPS][2]{};
    unsigned long long statCap[PMC_MAX_GROUPS]{};
    void* statTemp[PMC_MAX_GROUPS]{};
    unsigned long long* statWaveBase[PMC_MAX_GROUPS]{};
    uint32_t* statWaveFill[PMC_MAX_GROUPS]{};
    uint32_t* statChunkFill[PMC_MAX_GROUPS]{};

    template<typename T> int upload(const T* host, size_t count, const T** out)
    {
        *out = nullptr;
        if (!count) return PMC_OK;
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, count * sizeof(T));
        if (e != hipSuccess) return hipFail(e, "hipMalloc");
        allocations.push_back(d);
        e = hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice);
        if (e != hipSuccess) return hipFail(e, "hipMemcpy");
        *out = static_cast<const T*>(d);
        return PMC_OK;
    }
    // (planning pass of allocateSlots: the requests are only added up)
    bool planning{false};
    size_t plannedBytes{0};

    template<typename T> int allocate(size_t count, T** out, bool zero, std::vector<void*>* owner = nullptr)
    {
        *out = nullptr;
        if (!count) return PMC_OK;
        if (planning)
        {
            plannedBytes += (count * sizeof(T) + 255) & ~size_t(255);
            return PMC_OK;
        }
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, count * sizeof(T));
        if (e != hipSuccess) return hipFail(e, "hipMalloc");
        (owner ? *owner : allocations).push_back(d);
        if (zero)
        {
            e = hipMemset(d, 0, count * sizeof(T));
            if (e != hipSuccess) return hipFail(e, "hipMemset");
        }
        else if (pmcTune("PMC_POISON_ALLOCATIONS"))
        {
            // (test aid: what the engine does not initialise holds neither zeros -- fresh device memory -- nor plausible values -- memory of a
            // context destroyed before: a read of it shows)
            e = hipMemset(d, 0xA5, count * sizeof(T));
            if (e != hipSuccess) return hipFail(e, "hipMemset");
        }
        *out = static_cast<T*>(d);
        return PMC_OK;
    }
};

// pmc_tables.hip: the device tables of an octree (flattened tree, links, coordinate table; devToCell = device cell index -> caller's cell index)
// and of a Voronoi grid (cell records, neighbour entries, cone masks, tables of runs per cone and per observer)
int pmcUploadOctreeGrid(pmc_ctx* ctx, const pmc_scene* scene, const pmc_medium& med, std::vector<int32_t>& devToCell);
int pmcUploadVoronoiGrid(pmc_ctx* ctx, const pmc_scene* scene, const pmc_medium& med);
// pmc_run.hip: the pool of packet slots
int pmcAllocateSlots(pmc_ctx* ctx, int64_t n);

#endif
