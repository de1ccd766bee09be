// pmc_api.hip -- the extern "C" ABI of include/pmc.h: scene upload, the generation loop, downloads.
//
// pmc_create turns the reference-shaped tables of pmc_scene into the device layout of pmc_device.h.  For the
// octree this means: verify that every node box is consistent with one per-axis dyadic coordinate table (true for
// any tree built by recursive midpoint subdivision, OctTreeNode.cpp:22-33), build that table from the reference's
// own doubles, and replace the per-wall neighbour lists by four links per wall (per quadrant of the wall: the
// neighbour leaf covering it, or the internal node below which finer neighbours are found).  The reference lists
// themselves are uploaded too, re-indexed by cell, for the exact fallback path.
//
// pmc_run_primary drives the two kernels of pmc_kernels.hip in generations over a pool of packet slots.

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

// ---- tuning switches (include/pmc_tuning.h): a process-wide table set through pmc_tuning_set; the library reads three settings
// from the environment (PMC_NUM_SLOTS, PMC_NUM_GROUPS, PMC_STAT_POOL_BLOCKS) and nothing else
namespace
{
    std::mutex g_tuneMutex;
    // name -> value; the values live in a pool that is never shrunk, so that a pointer handed out by pmcTune stays valid when another
    // host thread sets or clears switches meanwhile (one host thread per device drives pmc_create / pmc_run_primary in the CLI and in
    // the multi-device tests).  A switch is SAMPLED where it is used -- the table layouts at pmc_create, the kernel selection at
    // pmc_run_primary: changing switches while a context is being created or is running gives that context either value.
    std::map<std::string, const std::string*>& tuneTable()
    {
        static std::map<std::string, const std::string*> table;
        return table;
    }
    const std::string* internTuneValue(const char* value)
    {
        static std::deque<std::string> pool;
        for (const auto& v : pool)
            if (v == value) return &v;
        pool.emplace_back(value);
        return &pool.back();
    }
}
// the value of a tuning switch, or null (the pointer stays valid for the life of the process)
extern "C" const char* pmcTune(const char* name)
{
    std::lock_guard<std::mutex> lock(g_tuneMutex);
    auto& table = tuneTable();
    auto at = table.find(name);
    return at == table.end() ? nullptr : at->second->c_str();
}
extern "C" int pmc_tuning_set(const char* name, const char* value)
{
    if (!name) return PMC_ERR_INVALID;
    std::lock_guard<std::mutex> lock(g_tuneMutex);
    if (value)
        tuneTable()[name] = internTuneValue(value);
    else
        tuneTable().erase(name);
    return PMC_OK;
}
extern "C" void pmc_tuning_clear(void)
{
    std::lock_guard<std::mutex> lock(g_tuneMutex);
    tuneTable().clear();
}

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
                                        int grid, hipStream_t stream);
extern "C" int pmcVoroPeelWavesPerSimd(void);
extern "C" hipError_t pmcLaunchVoroPeel(int slot, int rec, int tab, const int32_t* list, const unsigned long long* count, unsigned long long* xcdCursor, int segments,
                                        int grid, hipStream_t stream);
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
                                          size_t ldsBytes, const StatLogArgs* statLog, hipStream_t stream);
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

namespace
{
    thread_local std::string t_error;
    // constant-memory scene slots (pmc_kernels.hip c_scene): every device has its own copy of the symbol, so the table
    // of live contexts is per device; guarded, because contexts may be created from one host thread per GPU
    constexpr int MAX_DEVICES = 64;
    bool g_slotUsed[MAX_DEVICES][PMC_MAX_CONTEXTS] = {{false}};
    std::mutex g_slotMutex;

    int fail(int code, const std::string& message)
    {
        t_error = message;
        return code;
    }
    int hipFail(hipError_t e, const char* what)
    {
        return fail(PMC_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
    }
}

// error text of the calling thread, for the other translation units of the library (pmc_sampler.hip)
void pmcSetError(const std::string& message)
{
    t_error = message;
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

namespace
{
    // ---- octree flattening -------------------------------------------------------------------------
    struct TreeBuild
    {
        int lmax{0};
        int tabn{0};
        std::vector<double> table;       // [3][tabn]
        std::vector<LeafRec> leaves;     // by cell index m
        std::vector<CellRec> cells;      // by device cell index: the walk step's hot record
        std::vector<NodeRec> internals;  // by internal index
        std::vector<int32_t> nbrStart, nbrList;
        uint32_t rootLink{0};
        int coarseLevel{0};
        std::vector<uint32_t> coarse;    // [2^Lc]^3 (z, y, x): link of the covering node at level <= Lc
        // device numbering of the cells: dev = perm[m], depth-first order of the tree (the eight leaves of a node whose
        // children are all leaves are consecutive, in child order); cellExt[dev] = m (or -1 for padding); cellSlots =
        // entries per device table
        std::vector<int32_t> perm, cellExt;
        int cellSlots{0};
    };

    int buildTree(const pmc_grid& g, const double* density, TreeBuild& T)
    {
        const int numNodes = g.num_nodes;
        if (numNodes < 1) return fail(PMC_ERR_INVALID, "octree without nodes");
        int maxLevel = 0;
        for (int id = 0; id < numNodes; ++id) maxLevel = std::max(maxLevel, g.node_level[id]);
        if (maxLevel > PMC_MAX_LEVEL)
            return fail(PMC_ERR_UNSUPPORTED, "octree deeper than " + std::to_string(PMC_MAX_LEVEL) + " levels");
        T.lmax = maxLevel;
        T.tabn = (1 << maxLevel) + 1;
        const double unset = std::nan("");
        T.table.assign(3 * size_t(T.tabn), unset);

        // fine lower-corner indices of every node, from the topology alone
        std::vector<int32_t> fx(numNodes), fy(numNodes), fz(numNodes);
        std::vector<int32_t> internalIndex(numNodes, -1);
        int numInternal = 0;
        fx[0] = fy[0] = fz[0] = 0;
        if (g.node_level[0] != 0) return fail(PMC_ERR_INVALID, "octree root is not at level 0");
        for (int id = 0; id < numNodes; ++id)
        {
            int first = g.node_first_child[id];
            if (first < 0) continue;
            internalIndex[id] = numInternal++;
            if (first + 8 > numNodes) return fail(PMC_ERR_INVALID, "octree child index out of range");
            int half = 1 << (maxLevel - g.node_level[id] - 1);
            for (int l = 0; l < 8; ++l)
            {
                int c = first + l;
                if (g.node_level[c] != g.node_level[id] + 1) return fail(PMC_ERR_INVALID, "octree child level mismatch");
                fx[c] = fx[id] + ((l & 1) ? half : 0);
                fy[c] = fy[id] + ((l & 2) ? half : 0);
                fz[c] = fz[id] + ((l & 4) ? half : 0);
            }
        }
        // coordinate table from the reference's own box doubles, with consistency check
        auto put = [&](int axis, int index, double value) -> bool {
            double& slot = T.table[size_t(axis) * T.tabn + index];
            if (std::isnan(slot))
            {
                slot = value;
                return true;
            }
            return slot == value;
        };
        for (int id = 0; id < numNodes; ++id)
        {
            const double* b = g.node_box + 6 * size_t(id);
            int size = 1 << (maxLevel - g.node_level[id]);
            bool ok = put(0, fx[id], b[0]) && put(0, fx[id] + size, b[3]) && put(1, fy[id], b[1]) && put(1, fy[id] + size, b[4])
                      && put(2, fz[id], b[2]) && put(2, fz[id] + size, b[5]);
            if (!ok)
                return fail(PMC_ERR_UNSUPPORTED,
                            "octree node boxes are not consistent with a dyadic coordinate table (node " + std::to_string(id) + ")");
        }
        // entries that are no node's wall (inside coarse leaves) get the dyadic midpoints: no decision ever depends on
        // them, but the table becomes strictly monotonic, which the index search of topDown (pmc_walk.inc) relies on
        for (int axis = 0; axis < 3; ++axis)
            for (int size = 1 << maxLevel; size >= 2; size >>= 1)
                for (int lo = 0; lo + size <= (1 << maxLevel); lo += size)
                {
                    double& mid = T.table[size_t(axis) * T.tabn + lo + size / 2];
                    if (std::isnan(mid))
                        mid = (T.table[size_t(axis) * T.tabn + lo] + T.table[size_t(axis) * T.tabn + lo + size]) / 2.;
                }
        // device numbering of the cells: depth-first order of the tree, children in child order (tuning aids:
        // PMC_CELL_ORDER=ref keeps the caller's numbering -- breadth-first by level in SKIRT; PMC_CELL_SHUFFLE=g scatters
        // groups of 2^g cells of that numbering over the table)
        {
            const int n = g.num_cells;
            T.perm.assign(n, -1);
            const char* order = pmcTune("PMC_CELL_ORDER");
            int gb = -1;
            if (const char* env = pmcTune("PMC_CELL_SHUFFLE")) gb = atoi(env);
            if (gb >= 0 && gb <= 16)
            {
                const int groupSize = 1 << gb;
                const int numGroups = (n + groupSize - 1) / groupSize;
                std::vector<int32_t> where(numGroups);
                for (int i = 0; i < numGroups; ++i) where[i] = i;
                uint64_t state = 0x9E3779B97F4A7C15ull;  // fixed: the numbering is a pure function of the scene
                for (int i = numGroups - 1; i > 0; --i)
                {
                    state = state * 6364136223846793005ull + 1442695040888963407ull;
                    const int j = int((state >> 33) % uint64_t(i + 1));
                    std::swap(where[i], where[j]);
                }
                T.cellSlots = numGroups * groupSize;
                for (int m = 0; m < n; ++m) T.perm[m] = where[m >> gb] * groupSize + (m & (groupSize - 1));
            }
            else if (order && !strcmp(order, "ref"))
            {
                T.cellSlots = n;
                for (int m = 0; m < n; ++m) T.perm[m] = m;
            }
            else
            {
                T.cellSlots = n;
                int next = 0;
                std::vector<int> stack{0};
                while (!stack.empty())
                {
                    const int id = stack.back();
                    stack.pop_back();
                    const int first = g.node_first_child[id];
                    if (first < 0)
                    {
                        const int m = g.node_cell[id];
                        if (m < 0 || m >= n || T.perm[m] >= 0) return fail(PMC_ERR_INVALID, "octree leaf without a valid cell index");
                        T.perm[m] = next++;
                    }
                    else
                        for (int l = 7; l >= 0; --l) stack.push_back(first + l);
                }
                if (next != n) return fail(PMC_ERR_INVALID, "octree leaves and cells do not match");
            }
            T.cellExt.assign(T.cellSlots, -1);
            for (int m = 0; m < n; ++m) T.cellExt[T.perm[m]] = m;
        }
        // box code (pmc_device.h LeafRec::code): LDS byte offsets of the three lower wall entries + size exponent
        auto code = [&](int id) -> uint64_t {
            const uint64_t ox = 8ull * uint64_t(fx[id]);
            const uint64_t oy = 8ull * uint64_t(T.tabn + fy[id]);
            const uint64_t oz = 8ull * uint64_t(2 * T.tabn + fz[id]);
            return ox | (oy << 20) | (oz << 40) | ((uint64_t)(maxLevel - g.node_level[id]) << 60);
        };
        // link word (pmc_device.h): size exponent | index << 4 | node flag
        auto linkOf = [&](int id) -> uint32_t {
            if (id < 0) return PMC_LINK_NONE;
            const uint32_t e = uint32_t(maxLevel - g.node_level[id]);
            return g.node_first_child[id] < 0 ? (e | (uint32_t(T.perm[g.node_cell[id]]) << 4))
                                              : (e | (uint32_t(internalIndex[id]) << 4) | PMC_LINK_NODE);
        };
        // the link of a cell record through a wall: as linkOf, but an internal node whose children are all leaves with
        // consecutive device indices in child order becomes an octet link (the walk picks the child without a load)
        auto wallLinkOf = [&](int id) -> uint32_t {
            if (id < 0 || g.node_first_child[id] < 0) return linkOf(id);
            const int first = g.node_first_child[id];
            for (int l = 0; l < 8; ++l)
                if (g.node_first_child[first + l] >= 0) return linkOf(id);
            const int base = T.perm[g.node_cell[first]];
            for (int l = 1; l < 8; ++l)
                if (T.perm[g.node_cell[first + l]] != base + l) return linkOf(id);
            return uint32_t(maxLevel - g.node_level[id]) | (uint32_t(base) << 4) | PMC_LINK_OCTET;
        };
        T.rootLink = linkOf(0);
        // top-down search table (pmc_walk.inc topDown): per cell of the regular grid of level Lc the node of level Lc
        // that covers it, or the coarser leaf
        {
            const int lc = std::min(maxLevel, 6);
            T.coarseLevel = lc;
            const int nc = 1 << lc;
            T.coarse.resize(size_t(nc) * nc * nc);
            for (int cz = 0; cz < nc; ++cz)
                for (int cy = 0; cy < nc; ++cy)
                    for (int cx = 0; cx < nc; ++cx)
                    {
                        const int px = cx << (maxLevel - lc), py = cy << (maxLevel - lc), pz = cz << (maxLevel - lc);
                        int node = 0;
                        while (g.node_first_child[node] >= 0 && g.node_level[node] < lc)
                        {
                            const int half = 1 << (maxLevel - g.node_level[node] - 1);
                            const int l = ((px - fx[node]) >= half ? 1 : 0) + ((py - fy[node]) >= half ? 2 : 0)
                                          + ((pz - fz[node]) >= half ? 4 : 0);
                            node = g.node_first_child[node] + l;
                        }
                        T.coarse[(size_t(cz) * nc + cy) * nc + cx] = linkOf(node);
                    }
        }

        // the node at level <= level(id) that covers the region just across `wall` of node id (-1: outside the grid)
        auto covering = [&](int id, int wall) -> int {
            int axis = wall >> 1, side = wall & 1;
            int size = 1 << (maxLevel - g.node_level[id]);
            int px = fx[id], py = fy[id], pz = fz[id];  // a fine cell index inside the neighbour region
            int* pa = axis == 0 ? &px : axis == 1 ? &py : &pz;
            *pa += side ? size : -1;
            int full = 1 << maxLevel;
            if (*pa < 0 || *pa >= full) return -1;
            int node = 0;
            while (g.node_first_child[node] >= 0 && g.node_level[node] < g.node_level[id])
            {
                int half = 1 << (maxLevel - g.node_level[node] - 1);
                int l = ((px - fx[node]) >= half ? 1 : 0) + ((py - fy[node]) >= half ? 2 : 0) + ((pz - fz[node]) >= half ? 4 : 0);
                node = g.node_first_child[node] + l;
            }
            return node;
        };

        const int numCells = g.num_cells;
        const int cellSlots = T.cellSlots;
        T.leaves.assign(cellSlots, LeafRec{});
        T.cells.assign(size_t(cellSlots), CellRec{});
        T.internals.assign(numInternal, NodeRec{});
        T.nbrStart.assign(6 * size_t(cellSlots) + 1, 0);
        T.nbrList.clear();
        std::vector<int32_t> nodeOfCell(numCells, -1);
        for (int id = 0; id < numNodes; ++id)
        {
            int m = g.node_cell[id];
            if (g.node_first_child[id] < 0)
            {
                if (m < 0 || m >= numCells) return fail(PMC_ERR_INVALID, "octree leaf without a valid cell index");
                nodeOfCell[m] = id;
            }
            else
            {
                NodeRec& rec = T.internals[internalIndex[id]];
                rec.code = code(id);
                for (int l = 0; l < 8; ++l) rec.child[l] = linkOf(g.node_first_child[id] + l);
            }
        }
        for (int m = 0; m < numCells; ++m)
            if (nodeOfCell[m] < 0) return fail(PMC_ERR_INVALID, "cell without a leaf node");
        for (int dev = 0; dev < cellSlots; ++dev)
        {
            const int m = T.cellExt[dev];
            for (int wall = 0; wall < 6; ++wall) T.nbrStart[6 * size_t(dev) + wall] = (int32_t)T.nbrList.size();
            if (m < 0) continue;  // padding of the last group
            const int id = nodeOfCell[m];
            LeafRec& rec = T.leaves[dev];
            rec.code = code(id);
            rec.density = density[m];
            CellRec& hot = T.cells[dev];
            hot.density = density[m];
            for (int wall = 0; wall < 6; ++wall)
            {
                // the leaf across the wall (same size or coarser), or the same-size internal node (finer neighbours: the walk
                // picks the child by the index bits of its position), or "outside"
                hot.link[wall] = wallLinkOf(covering(id, wall));
                // the reference's neighbour list of this leaf, in device numbering
                T.nbrStart[6 * size_t(dev) + wall] = (int32_t)T.nbrList.size();
                for (int qq = g.nbr_start[6 * size_t(id) + wall]; qq < g.nbr_start[6 * size_t(id) + wall + 1]; ++qq)
                {
                    int nb = g.nbr_list[qq];
                    if (g.node_first_child[nb] >= 0)
                        return fail(PMC_ERR_INVALID, "neighbour list of a leaf contains a non-leaf node");
                    T.nbrList.push_back(T.perm[g.node_cell[nb]]);
                }
            }
        }
        T.nbrStart[6 * size_t(cellSlots)] = (int32_t)T.nbrList.size();
        return PMC_OK;
    }

    int allocateSlotArrays(pmc_ctx* ctx, int64_t n);

    // the slot pool of n histories in flight: first added up and held against the free device memory (a clear message instead of
    // a failed hipMalloc half-way), then allocated
    int allocateSlots(pmc_ctx* ctx, int64_t n)
    {
        hipSetDevice(ctx->device);
        for (void* p : ctx->slotAllocations) hipFree(p);
        ctx->slotAllocations.clear();
        ctx->allocatedSlots = 0;
        ctx->planning = true;
        ctx->plannedBytes = 0;
        int rc = allocateSlotArrays(ctx, n);
        ctx->planning = false;
        if (rc) return rc;
        size_t freeBytes = 0, totalBytes = 0;
        // (the default number of slots is sized for the 288 GB of an MI355X; on a device, or next to other contexts, where it would take
        // more than half of the free memory the default steps down -- a number the caller has set is taken as it is)
        if (!ctx->slotsConfigured && n > (int64_t(1) << 20) && hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess && ctx->plannedBytes > freeBytes / 2)
        {
            // (the requested default, ctx->numSlots, stays as it is: a later segment asks again -- pmc_run_primary -- and gets the larger
            // pool once the memory is there)
            const int64_t less = std::max<int64_t>(int64_t(1) << 20, n / 2);
            fprintf(stderr, "libpmc: device %d has %.1f GB free, %lld packet slots would take %.1f GB: this segment runs with %lld slots (fewer histories in "
                            "flight, somewhat lower throughput; PMC_NUM_SLOTS / pmc_set_num_slots set the number)\n",
                    ctx->device, freeBytes * 1e-9, (long long)n, ctx->plannedBytes * 1e-9, (long long)less);
            ctx->steppedDownFree = freeBytes;
            return allocateSlots(ctx, less);
        }
        if (hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess && ctx->plannedBytes > freeBytes)
        {
            char text[512];
            snprintf(text, sizeof(text),
                     "the state of %lld photon histories in flight needs %.2f GB of device memory (%.0f bytes per history), %.2f GB of %.2f GB are "
                     "free: lower the number with pmc_set_num_slots or PMC_NUM_SLOTS",
                     (long long)n, ctx->plannedBytes * 1e-9, double(ctx->plannedBytes) / double(n), freeBytes * 1e-9, totalBytes * 1e-9);
            return fail(PMC_ERR_NOMEM, text);
        }
        rc = allocateSlotArrays(ctx, n);
        if (rc)
        {
            for (void* p : ctx->slotAllocations) hipFree(p);
            ctx->slotAllocations.clear();
            ctx->allocatedSlots = 0;
        }
        return rc;
    }

    int allocateSlotArrays(pmc_ctx* ctx, int64_t n)
    {
        SlotArrays& A = ctx->dev.slots;
        std::memset(&A, 0, sizeof(A));
        auto& own = ctx->slotAllocations;
        int rc;
        double** dbl[] = {&A.rx, &A.ry, &A.rz, &A.kx, &A.ky, &A.kz, &A.lambda, &A.W, &A.Lthreshold, &A.taupath, &A.tausample, &A.rngSpare, &A.sint,
                          &A.nint, &A.dustExt, &A.dustSca, &A.dustAsym};
        for (double** d : dbl)
            if ((rc = ctx->allocate<double>(n, d, false, &own))) return rc;
        if (ctx->dev.explicit_absorption && (rc = ctx->allocate<double>(n, &A.dustAbs, false, &own))) return rc;
        if (ctx->dev.num_media > 1 && !ctx->dev.mono && (rc = ctx->allocate<int32_t>(n * ctx->dev.num_media, &A.dustIdx, false, &own))) return rc;
        if ((rc = ctx->allocate<uint64_t>(n, &A.history, false, &own))) return rc;
        if ((rc = ctx->allocate<uint32_t>(n, &A.rngBlock, false, &own))) return rc;
        int32_t** ints[] = {&A.mode, &A.nscatt, &A.mint};
        for (int32_t** d : ints)
            if ((rc = ctx->allocate<int32_t>(n, d, true, &own))) return rc;
        if ((rc = ctx->allocate<double>(size_t(n) * size_t(ctx->dev.num_instruments), &A.ppW, false, &own))) return rc;
        if ((rc = ctx->allocate<double>(size_t(n) * size_t(ctx->dev.num_instruments), &A.ptau, false, &own))) return rc;
        if ((rc = ctx->allocate<int32_t>(size_t(n) * size_t(ctx->dev.num_instruments), &A.ell, true, &own))) return rc;
        if (ctx->dev.any_stats && (rc = ctx->allocate<int32_t>(size_t(n) * size_t(ctx->dev.num_instruments) * 16, &A.statHead, true, &own))) return rc;
        if (ctx->dev.rf_store && (rc = ctx->allocate<int32_t>(n, &A.rfell, true, &own))) return rc;
        if (ctx->dev.any_stats)
        {
            size_t entries = size_t(ctx->dev.num_instruments) * PMC_STAT_CAP * size_t(n);
            if ((rc = ctx->allocate<int32_t>(entries, &A.statBin, false, &own))) return rc;
            if ((rc = ctx->allocate<double>(entries, &A.statW, false, &own))) return rc;
            // continuation blocks of the lists (pmc_device.h DevScene::stat_pool_*): by default one block per four slots -- or,
            // for a ski file that asks for many scattering events per history (minScattEvents), what such histories need in
            // every slot at once; environment PMC_STAT_POOL_BLOCKS sets the number
            DevScene& D = ctx->dev;
            const int minEvents = D.min_scatt_events;
            int64_t blocks = minEvents > 16 ? n * int64_t((minEvents + 2 * PMC_STAT_CAP - 1) / PMC_STAT_CAP) : n / 4;
            blocks = std::max<int64_t>(blocks, 1024) * D.num_instruments;
            // (... and up to one block per slot and instrument where an eighth of the free device memory allows it: the sparse
            // generations at the end of a segment keep the blocks of retired histories out of the pool, and a long non-forced history
            // in an optically thick medium needs more than the default)
            {
                size_t freeBytes = 0, totalBytes = 0;
                if (hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess)
                {
                    const int64_t afford = int64_t(freeBytes / 8 / (PMC_STAT_CAP * 12 + 12));
                    blocks = std::max(blocks, std::min<int64_t>(n * int64_t(D.num_instruments), afford));
                }
            }
            if (const char* env = getenv("PMC_STAT_POOL_BLOCKS")) blocks = std::max<int64_t>(PMC_MAX_GROUPS, atoll(env));
            blocks = std::min<int64_t>(blocks, int64_t(1) << 30);
            if ((rc = ctx->allocate<int32_t>(size_t(blocks) * PMC_STAT_CAP, &D.stat_pool_bin, false, &own))) return rc;
            if ((rc = ctx->allocate<double>(size_t(blocks) * PMC_STAT_CAP, &D.stat_pool_w, false, &own))) return rc;
            if ((rc = ctx->allocate<int32_t>(size_t(blocks), &D.stat_pool_next, false, &own))) return rc;
            if ((rc = ctx->allocate<int32_t>(size_t(blocks), &D.stat_pool_free, false, &own))) return rc;
            if ((rc = ctx->allocate<int32_t>(size_t(blocks), &ctx->statPoolIota, false, &own))) return rc;
            if (!ctx->planning)
            {
                std::vector<int32_t> iota(static_cast<size_t>(blocks));
                for (size_t i = 0; i < iota.size(); ++i) iota[i] = (int32_t)i;
                if (hipMemcpy(ctx->statPoolIota, iota.data(), iota.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess)
                    return fail(PMC_ERR_DEVICE, "hipMemcpy failed");
                ctx->statPoolBlocks = blocks;
            }
        }
        TaskArrays& K = ctx->dev.tasks;
        std::memset(&K, 0, sizeof(K));
        // task records: the propagation walk + one peel-off walk per instrument, per slot
        const size_t nt = size_t(n) * size_t(1 + ctx->dev.num_instruments);
        double** tdbl[] = {&K.rx, &K.ry, &K.rz, &K.kx, &K.ky, &K.kz, &K.s0, &K.ds, &K.target};
        for (double** d : tdbl)
            if ((rc = ctx->allocate<double>(nt, d, false, &own))) return rc;
        int32_t** tints[] = {&K.cell, &K.cijk};
        for (int32_t** d : tints)
            if ((rc = ctx->allocate<int32_t>(nt, d, false, &own))) return rc;
        // ended-history counts per tile of 64 slots (padded: the scan reads and writes 16 bytes at a time)
        if ((rc = ctx->allocate<uint32_t>(size_t(n) / 64 + 64, &K.endedCount, true, &own))) return rc;
        if ((rc = ctx->allocate<int32_t>(2 * size_t(n), &K.liveList, false, &own))) return rc;
        if ((rc = ctx->allocate<uint32_t>(nt, &K.bits, false, &own))) return rc;
    if (ctx->dev.grid_kind == PMC_GRID_OCTREE && (rc = ctx->allocate<uint64_t>(nt, &K.pidx, false, &own))) return rc;
        if (ctx->planning) return PMC_OK;
        A.num_slots = n;
        ctx->allocatedSlots = n;
        ctx->sceneDirty = true;
        return PMC_OK;
    }
}

extern "C" {

int pmc_abi_version(void)
{
    return PMC_ABI_VERSION;
}

// what this binary was built with (the Makefile hands its HIPFLAGS over): results are bit-compatible with the reference only under
// -ffp-contract=off (the reference build has no fused multiply-add), and the detector atomics are hardware f64 adds
// only under -munsafe-fp-atomics
#define PMC_STRINGIFY2(x) #x
#define PMC_STRINGIFY(x) PMC_STRINGIFY2(x)
#ifndef PMC_BUILD_FLAGS
#define PMC_BUILD_FLAGS "unknown (not built by the Makefile)"
#endif
const char* pmc_build_info(void)
{
    return "libpmc ABI " PMC_STRINGIFY(PMC_ABI_VERSION) ", gfx950, " __VERSION__ ", flags: " PMC_BUILD_FLAGS
#if defined(__FP_FAST_FMA) || defined(__FAST_MATH__)
           " [fast-math macros defined]"
#endif
        ;
}

namespace
{
    // a * b + c with run-time operands: 0 when the product is rounded before the sum (-ffp-contract=off), -2^-60 when the compiler fused the two
    __global__ void contractionProbeKernel(double a, double b, double c, double* out) { out[0] = a * b + c; }

    // one launch per process and device, at the first pmc_create: a binary whose compiler contracted a * b + c computes other last bits than
    // the reference in every optical depth and exit distance, silently; it is refused instead
    int checkContraction(int device)
    {
        static std::mutex m;
        static bool checked[64] = {false};
        std::lock_guard<std::mutex> lock(m);
        if (device >= 0 && device < 64 && checked[device]) return PMC_OK;
        double* d = nullptr;
        double h = 1.;
        if (hipMalloc(&d, sizeof(double)) != hipSuccess) return fail(PMC_ERR_DEVICE, "hipMalloc failed");
        hipLaunchKernelGGL(contractionProbeKernel, dim3(1), dim3(1), 0, 0, 1. + 0x1p-30, 1. - 0x1p-30, -1., d);
        const hipError_t e = hipMemcpy(&h, d, sizeof(double), hipMemcpyDeviceToHost);
        hipFree(d);
        if (e != hipSuccess) return hipFail(e, "contraction self-test");
        if (h != 0.)
            return fail(PMC_ERR_DEVICE, std::string("this libpmc.so was compiled with floating-point contraction (a * b + c fused): its results would differ from "
                                                    "the reference's in the last bits everywhere; rebuild with -ffp-contract=off.  ") + pmc_build_info());
        if (device >= 0 && device < 64) checked[device] = true;
        return PMC_OK;
    }
}

const char* pmc_last_error(void)
{
    return t_error.c_str();
}

int64_t pmc_frame_layout_of(const pmc_scene* scene, int32_t instrument, pmc_frame_layout* out)
{
    if (!scene) return fail(PMC_ERR_INVALID, "null scene");
    return pmc_layout_compute(scene, instrument, out);
}

void pmc_destroy(pmc_ctx* ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    for (int g = 0; g < PMC_MAX_GROUPS; ++g)
    {
        if (ctx->groupStream[g]) hipStreamSynchronize(ctx->groupStream[g]);
        if (ctx->peelStream[g]) hipStreamSynchronize(ctx->peelStream[g]);
    }
    for (void* p : ctx->allocations) hipFree(p);
    for (void* p : ctx->slotAllocations) hipFree(p);
    for (void* p : ctx->rfAllocations) hipFree(p);
    if (ctx->pinned) hipHostFree(ctx->pinned);
    for (hipEvent_t e : {ctx->evStart, ctx->evStop})
        if (e) hipEventDestroy(e);
    for (int g = 0; g < PMC_MAX_GROUPS; ++g)
    {
        for (hipEvent_t e : {ctx->evA[g], ctx->evB[g], ctx->evC[g], ctx->evJoin[g], ctx->evProp[g]})
            if (e) hipEventDestroy(e);
        if (g > 0 && ctx->groupStream[g]) hipStreamDestroy(ctx->groupStream[g]);
        if (ctx->peelStream[g]) hipStreamDestroy(ctx->peelStream[g]);
    }
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    if (ctx->slot >= 0)
    {
        std::lock_guard<std::mutex> lock(g_slotMutex);
        g_slotUsed[ctx->device][ctx->slot] = false;
    }
    delete ctx;
}

int pmc_create(const pmc_scene* scene, int32_t device, pmc_ctx** out)
{
    if (!scene || !out) return fail(PMC_ERR_INVALID, "null argument");
    *out = nullptr;
    if (scene->abi_version != PMC_ABI_VERSION) return fail(PMC_ERR_INVALID, "pmc_scene ABI version mismatch");
    if (pmcExperimentBuild())
    {
        static std::atomic<bool> said{false};
        if (!said.exchange(true)) fprintf(stderr, "libpmc: this library was built with an ablation / perturbation macro (a tuning experiment): its results are NOT those of the engine\n");
    }
    if (scene->num_media > PMC_MAX_MEDIA) return fail(PMC_ERR_UNSUPPORTED, "more than PMC_MAX_MEDIA medium components");
    if (scene->num_media > 1 && !scene->media) return fail(PMC_ERR_INVALID, "num_media > 1 without pmc_scene::media");
    if (scene->num_instruments < 1 || scene->num_instruments > PMC_MAX_INSTRUMENTS)
        return fail(PMC_ERR_UNSUPPORTED, "between 1 and " + std::to_string(PMC_MAX_INSTRUMENTS) + " instruments are supported");
    if (scene->grid.kind != PMC_GRID_CARTESIAN && scene->grid.kind != PMC_GRID_OCTREE && scene->grid.kind != PMC_GRID_VORONOI)
        return fail(PMC_ERR_UNSUPPORTED, "unsupported grid kind");
    if (scene->instruments[0].same_observer_as_preceding) return fail(PMC_ERR_INVALID, "first instrument cannot share an observer");

    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count < 1)
        return fail(PMC_ERR_DEVICE, "no HIP device available: the MI355X engine cannot run (there is no CPU fallback)");
    if (device < 0 || device >= count) return fail(PMC_ERR_INVALID, "invalid device index");
    HIP_TRY(hipSetDevice(device));
    if (int rcProbe = checkContraction(device)) return rcProbe;

    if (device >= MAX_DEVICES) return fail(PMC_ERR_UNSUPPORTED, "device index beyond the context table");
    pmc_ctx* ctx = new pmc_ctx();
    ctx->device = device;
    {
        std::lock_guard<std::mutex> lock(g_slotMutex);
        for (int sl = 0; sl < PMC_MAX_CONTEXTS && ctx->slot < 0; ++sl)
            if (!g_slotUsed[device][sl])
            {
                g_slotUsed[device][sl] = true;
                ctx->slot = sl;
            }
    }
    if (ctx->slot < 0)
    {
        delete ctx;
        return fail(PMC_ERR_NOMEM, "too many live pmc contexts (at most " + std::to_string(PMC_MAX_CONTEXTS) + ")");
    }
    int rc = PMC_OK;
    auto bail = [&](int code) {
        pmc_destroy(ctx);
        return code;
    };
    if (hipStreamCreate(&ctx->stream) != hipSuccess) return bail(fail(PMC_ERR_DEVICE, "hipStreamCreate failed"));
    ctx->groupStream[0] = ctx->stream;
    for (int g = 1; g < PMC_MAX_GROUPS; ++g)
        if (hipStreamCreate(&ctx->groupStream[g]) != hipSuccess) return bail(fail(PMC_ERR_DEVICE, "hipStreamCreate failed"));
    for (int g = 0; g < PMC_MAX_GROUPS; ++g)
        if (hipStreamCreate(&ctx->peelStream[g]) != hipSuccess) return bail(fail(PMC_ERR_DEVICE, "hipStreamCreate failed"));
    for (hipEvent_t* ev : {&ctx->evStart, &ctx->evStop})
        if (hipEventCreate(ev) != hipSuccess) return bail(fail(PMC_ERR_DEVICE, "hipEventCreate failed"));
    for (int g = 0; g < PMC_MAX_GROUPS; ++g)
        for (hipEvent_t* ev : {&ctx->evA[g], &ctx->evB[g], &ctx->evC[g], &ctx->evJoin[g], &ctx->evProp[g]})
            if (hipEventCreate(ev) != hipSuccess) return bail(fail(PMC_ERR_DEVICE, "hipEventCreate failed"));
    if (const char* env = getenv("PMC_NUM_GROUPS"))
    {
        ctx->numGroups = std::min(PMC_MAX_GROUPS, std::max(1, atoi(env)));
        ctx->groupsConfigured = true;  // (an explicit setting: also a Voronoi scene runs with it)
    }
    if (hipHostMalloc(reinterpret_cast<void**>(&ctx->pinned), 16 * sizeof(unsigned long long)) != hipSuccess)
        return bail(fail(PMC_ERR_DEVICE, "hipHostMalloc failed"));

    DevScene& D = ctx->dev;
    const pmc_grid& g = scene->grid;
    // component 0 of the medium system: pmc_scene::media[0] when there are several components (pmc.h: `media` replaces `medium` then, which
    // the caller may leave zeroed), else pmc_scene::medium.  Its cell densities go into the hot cell records, its dust tables into DevScene.
    const pmc_medium& med = scene->num_media > 1 ? scene->media[0] : scene->medium;
    if (!med.number_density) return bail(fail(PMC_ERR_INVALID, "medium component 0 without number_density"));
    std::vector<int32_t> devToCell;  // octree: device cell index -> caller's cell index (else empty: the same numbering)
    D.grid_kind = g.kind;
    D.gx0 = g.xmin, D.gy0 = g.ymin, D.gz0 = g.zmin;
    D.gx1 = g.xmax, D.gy1 = g.ymax, D.gz1 = g.zmax;
    D.eps = g.eps;
    D.num_cells = g.num_cells;
    if (g.kind == PMC_GRID_CARTESIAN)
    {
        D.nx = g.nx, D.ny = g.ny, D.nz = g.nz;
        if ((rc = ctx->upload(g.xv, g.nx + 1, &D.xv))) return bail(rc);
        if ((rc = ctx->upload(g.yv, g.ny + 1, &D.yv))) return bail(rc);
        if ((rc = ctx->upload(g.zv, g.nz + 1, &D.zv))) return bail(rc);
        if ((rc = ctx->upload(med.number_density, g.num_cells, &D.cell_density))) return bail(rc);
        D.lds_grid_len = (g.nx + 1) + (g.ny + 1) + (g.nz + 1);
        D.lmax = 0;
        if (int64_t(g.nx) * g.ny * g.nz != int64_t(g.num_cells)) return bail(fail(PMC_ERR_INVALID, "Cartesian grid: cell count does not match the border arrays"));
    }
    else if (g.kind == PMC_GRID_VORONOI)
    {
        if (g.num_cells < 1 || !g.site || !g.vnbr_start || !g.vnbr_list || g.vblock_n < 1 || !g.vblock_start || !g.vblock_list)
            return bail(fail(PMC_ERR_INVALID, "Voronoi grid tables are missing"));
        std::vector<double> rec(4 * size_t(g.num_cells));
        for (int m = 0; m < g.num_cells; ++m)
        {
            rec[4 * size_t(m)] = g.site[3 * size_t(m)], rec[4 * size_t(m) + 1] = g.site[3 * size_t(m) + 1];
            rec[4 * size_t(m) + 2] = g.site[3 * size_t(m) + 2], rec[4 * size_t(m) + 3] = med.number_density[m];
        }
        const size_t nb3 = size_t(g.vblock_n) * g.vblock_n * g.vblock_n;
        for (int m = 0; m < g.num_cells; ++m)
            for (int q = g.vnbr_start[m]; q < g.vnbr_start[m + 1]; ++q)
                if (g.vnbr_list[q] < -6 || g.vnbr_list[q] >= g.num_cells)
                    return bail(fail(PMC_ERR_INVALID, "Voronoi neighbour list holds an invalid index"));
        if ((rc = ctx->upload(rec.data(), rec.size(), &D.vsite))) return bail(rc);
        if ((rc = ctx->upload(g.vnbr_start, size_t(g.num_cells) + 1, &D.vnbr_start))) return bail(rc);
        if ((rc = ctx->upload(g.vnbr_list, size_t(g.vnbr_start[g.num_cells]), &D.vnbr_list))) return bail(rc);
        {
            const size_t np = size_t(g.vnbr_start[g.num_cells]);
            std::vector<double> pair(4 * np, 0.);
            for (size_t q = 0; q < np; ++q)
            {
                const int mi = g.vnbr_list[q];
                if (mi >= 0)
                {
                    pair[4 * q] = g.site[3 * size_t(mi)], pair[4 * q + 1] = g.site[3 * size_t(mi) + 1];
                    pair[4 * q + 2] = g.site[3 * size_t(mi) + 2];
                }
                const long long bits = mi;
                std::memcpy(&pair[4 * q + 3], &bits, sizeof(double));
            }
            if ((rc = ctx->upload(pair.data(), pair.size(), &D.vpair))) return bail(rc);
        }
        {
            // the header record of a cell (DevScene::vhead)
            std::vector<double> head(8 * size_t(g.num_cells), 0.);
            for (int m = 0; m < g.num_cells; ++m)
            {
                for (int a = 0; a < 3; ++a) head[8 * size_t(m) + a] = g.site[3 * size_t(m) + a];
                head[8 * size_t(m) + 3] = med.number_density[m];
                const int32_t bounds[2] = {g.vnbr_start[m], g.vnbr_start[m + 1]};
                std::memcpy(&head[8 * size_t(m) + 4], bounds, sizeof(double));
            }
            if ((rc = ctx->upload(head.data(), head.size(), &D.vhead))) return bail(rc);
        }
        D.vcull = nullptr;
        for (int i = 0; i < 16; ++i) D.vobs_of_inst[i] = -1;
        if (!pmcTune("PMC_VORO_NO_CULL"))
        {
            // neighbours that no direction of a cone can leave the cell through (DevScene::vcull).  A cone = the directions with
            // one sign pattern and one order of |k_x|, |k_y|, |k_z|: the non-negative combinations of three extreme rays, so
            // n . k <= 0 on the cone <=> n . e <= 0 for the three rays; the margin (1e-9 |n| |e|) is far above the rounding of
            // the kernel's n . k.
            static const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
            // every cone is divided once more at the midpoints of its edges (PMC_VORO_CONES = 192: four sub-cones, the rays
            // e1, e1 + e2, e1 + e3 | e2, e1 + e2, e2 + e3 | e3, e1 + e3, e2 + e3 | e1 + e2, e1 + e3, e2 + e3)
            static const int sub[4][3][3] = {{{1, 0, 0}, {1, 1, 0}, {1, 0, 1}}, {{0, 1, 0}, {1, 1, 0}, {0, 1, 1}}, {{0, 0, 1}, {1, 0, 1}, {0, 1, 1}},
                                             {{1, 1, 0}, {1, 0, 1}, {0, 1, 1}}};
            // 32-bit masks, 768 bytes per cell: a cell of the 10^5- and 10^6-site grids of the BASELINE scene has 15.2 / 15.4
            // neighbours on average, 99 % of the cells at most 24, 35 at most (neighbours beyond the 32nd are always read)
            const int ncell = g.num_cells;
            std::vector<uint32_t> cull(size_t(ncell) * PMC_VORO_CONES, 0u);
            auto cullCells = [&](int mFirst, int mLast) {
            for (int m = mFirst; m < mLast; ++m)
                for (int q = g.vnbr_start[m]; q < g.vnbr_start[m + 1] && q - g.vnbr_start[m] < 32; ++q)
                {
                    const int j = q - g.vnbr_start[m];
                    const int mi = g.vnbr_list[q];
                    double nv[3] = {0., 0., 0.};
                    double norm = 0.;
                    if (mi >= 0)
                    {
                        for (int a = 0; a < 3; ++a) nv[a] = g.site[3 * size_t(mi) + a] - g.site[3 * size_t(m) + a];
                        norm = std::sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
                    }
                    for (int sgn = 0; sgn < 8; ++sgn)
                        for (int p = 0; p < 6; ++p)
                        {
                            // the extreme rays of the cone: e1 along the largest component, e2 = e1 + the second, e3 = e2 + the third
                            double e[3][3] = {{0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}};
                            for (int r = 0; r < 3; ++r)
                                for (int t = r; t < 3; ++t) e[t][perm[p][r]] = ((sgn >> perm[p][r]) & 1) ? -1. : 1.;
                            for (int c = 0; c < PMC_VORO_CONES / 48; ++c)
                            {
                                bool skip;
                                if (mi >= 0)
                                {
                                    skip = norm > 0.;
                                    for (int r = 0; r < 3 && skip; ++r)
                                    {
                                        double ray[3];
                                        for (int a = 0; a < 3; ++a)
                                            ray[a] = PMC_VORO_CONES == 48 ? e[r][a] : sub[c][r][0] * e[0][a] + sub[c][r][1] * e[1][a] + sub[c][r][2] * e[2][a];
                                        const double dot = nv[0] * ray[0] + nv[1] * ray[1] + nv[2] * ray[2];
                                        if (!(dot <= -1e-9 * norm * 4.)) skip = false;
                                    }
                                }
                                else
                                {
                                    // walls -1 .. -6: x min, x max, y min, y max, z min, z max (reached only by k_a < 0 / k_a > 0)
                                    const int wall = -mi - 1;
                                    if (wall > 5) continue;
                                    const bool negative = ((sgn >> (wall >> 1)) & 1) != 0;
                                    skip = (wall & 1) ? negative : !negative;
                                }
                                // (cone-major: the masks of one cone are consecutive -- the walks towards an observer all use one cone)
                                if (skip) cull[size_t((sgn * 6 + p) * (PMC_VORO_CONES / 48) + c) * size_t(ncell) + size_t(m)] |= 1u << j;
                            }
                        }
                }
            };
            {
                // (cells are independent: all host cores)
                const int workers = std::max(1, std::min<int>(64, (int)std::thread::hardware_concurrency()));
                std::vector<std::thread> pool;
                for (int t = 0; t < workers; ++t)
                    pool.emplace_back(cullCells, int(int64_t(ncell) * t / workers), int(int64_t(ncell) * (t + 1) / workers));
                for (auto& t : pool) t.join();
            }
            if ((rc = ctx->upload(cull.data(), cull.size(), &D.vcull))) return bail(rc);
            // ---- a table of RUNS (DevScene::vobs_run, vgen_run): per cell ONE run of 64-byte units -- header {site, density, number of entries}, then its
            // entries (first[m] .. first[m + 1] of `entries`: {site x, y, z, neighbour index}) in groups of PMC_VORO_RUN_LANES, a group as {x, y} of each
            // entry followed by {z, tag} of each: the lanes that share a walk read a group with two coalesced loads.  An entry's tag carries the unit
            // at which its neighbour's run starts next to the neighbour's index
            const auto uploadRuns = [&](const std::vector<double>& entries, const std::vector<int32_t>& first, const double** runsOut, const uint32_t** startOut) -> int {
                constexpr size_t LANES = PMC_VORO_RUN_LANES, GROUP_UNITS = LANES / 2;
                // (cells with more entries than a link can name -- 30 -- say so in their header; PMC_VORO_LINK_COUNT_MAX lowers the limit: a test of that path)
                uint32_t linkCountMax = PMC_VORO_RUN_COUNT_UNKNOWN - 1u;
                if (const char* v = pmcTune("PMC_VORO_LINK_COUNT_MAX")) linkCountMax = std::min<uint32_t>(linkCountMax, (uint32_t)std::max(0, atoi(v)));
                std::vector<uint32_t> start(size_t(ncell) + 1);
                size_t units = 0;
                for (int m = 0; m < ncell; ++m)
                {
                    // (the link to the run: its first unit and, up to 30, the number of its entries -- pmc_device.h PMC_VORO_RUN_UNIT_BITS)
                    const uint32_t entriesOf = uint32_t(first[m + 1] - first[m]);
                    start[m] = uint32_t(units & PMC_VORO_RUN_UNIT_MASK) | ((entriesOf > linkCountMax ? PMC_VORO_RUN_COUNT_UNKNOWN : entriesOf) << PMC_VORO_RUN_UNIT_BITS);
                    units += 1 + GROUP_UNITS * ((size_t(entriesOf) + LANES - 1) / LANES);
                }
                if (units + PMC_VORO_RUN_PAD >= (size_t(1) << PMC_VORO_RUN_UNIT_BITS))
                    return fail(PMC_ERR_UNSUPPORTED, "Voronoi table of runs beyond 2^27 units of 64 bytes");
                std::vector<double> orun(8 * (units + PMC_VORO_RUN_PAD), 0.);  // (padding: a walk may request a group that the run does not have)
                const unsigned long long noEntry = (unsigned long long)(uint32_t)(-7);
                for (int m = 0; m < ncell; ++m)
                {
                    double* head = &orun[8 * size_t(start[m] & PMC_VORO_RUN_UNIT_MASK)];
                    for (int a = 0; a < 3; ++a) head[a] = g.site[3 * size_t(m) + a];
                    head[3] = med.number_density[m];
                    const int32_t count[2] = {first[m + 1] - first[m], 0};
                    std::memcpy(&head[4], count, sizeof(double));
                    const size_t groups = (size_t(count[0]) + LANES - 1) / LANES;
                    for (size_t e = 0; e < groups * LANES; ++e)
                    {
                        double* group = head + 8 + 4 * LANES * (e / LANES);
                        double* xy = group + 2 * (e % LANES);
                        double* zt = group + 2 * LANES + 2 * (e % LANES);
                        unsigned long long tag = noEntry;
                        if (e < size_t(count[0]))
                        {
                            const double* src = &entries[4 * (size_t(first[m]) + e)];
                            xy[0] = src[0], xy[1] = src[1], zt[0] = src[2];
                            long long bits;
                            std::memcpy(&bits, &src[3], sizeof(double));
                            const int mi = int(bits);
                            tag = (unsigned long long)(uint32_t)mi | (mi >= 0 ? (unsigned long long)start[mi] << 32 : 0ull);
                        }
                        std::memcpy(&zt[1], &tag, sizeof(double));
                    }
                }
                int rcu;
                if ((rcu = ctx->upload(orun.data(), orun.size(), runsOut))) return rcu;
                return ctx->upload(start.data(), size_t(ncell), startOut);
            };
            // ---- all neighbours of a cell as a run (DevScene::vgen_run): what a PROPAGATION walk in voroPropKernel reads -- no mask, one run of
            // memory (4.75 lines per visit instead of header + mask + scattered entries: 6.2); left out where device memory is short
            if (scene->num_media <= 1 && !pmcTune("PMC_VORO_NO_PROP_KERNEL"))
            {
                size_t freeBytes = 0, totalBytes = 0;
                const size_t need = 32 * size_t(g.vnbr_start[ncell]) + 96 * size_t(ncell);
                if (hipMemGetInfo(&freeBytes, &totalBytes) != hipSuccess || need <= freeBytes / 4)
                {
                    std::vector<double> all(4 * size_t(g.vnbr_start[ncell]), 0.);
                    std::vector<int32_t> firstAll(size_t(ncell) + 1);
                    for (int m = 0; m <= ncell; ++m) firstAll[m] = g.vnbr_start[m];
                    for (int q = 0; q < g.vnbr_start[ncell]; ++q)
                    {
                        const int mi = g.vnbr_list[q];
                        double* e = &all[4 * size_t(q)];
                        if (mi >= 0) e[0] = g.site[3 * size_t(mi)], e[1] = g.site[3 * size_t(mi) + 1], e[2] = g.site[3 * size_t(mi) + 2];
                        const long long bits = mi;
                        std::memcpy(&e[3], &bits, sizeof(double));
                    }
                    if ((rc = uploadRuns(all, firstAll, &D.vgen_run, &D.vgen_start))) return bail(rc);
                    // ... and per main cone the entries its sub-cones' masks keep (a neighbour beyond the 32nd is always kept)
                    if (PMC_VORO_CONES == 192 && !pmcTune("PMC_VORO_NO_CONE_TABLES") && (hipMemGetInfo(&freeBytes, &totalBytes) != hipSuccess || 48 * need <= freeBytes / 4))
                    {
                        const int workers = std::max(1, std::min<int>(16, (int)std::thread::hardware_concurrency()));
                        for (int c0 = 0; c0 < 48; c0 += workers)
                        {
                            const int nc = std::min(workers, 48 - c0);
                            std::vector<std::vector<double>> kept(nc);
                            std::vector<std::vector<int32_t>> firstKept(nc);
                            std::vector<std::thread> pool;
                            for (int w = 0; w < nc; ++w)
                                pool.emplace_back([&, w]() {
                                    const int c = c0 + w;
                                    std::vector<double>& e = kept[w];
                                    std::vector<int32_t>& f = firstKept[w];
                                    f.resize(size_t(ncell) + 1);
                                    e.reserve(all.size() * 3 / 4);
                                    for (int m = 0; m < ncell; ++m)
                                    {
                                        f[m] = int32_t(e.size() / 4);
                                        uint32_t mask = 0xFFFFFFFFu;  // culled by every sub-cone
                                        for (int sub = 0; sub < 4; ++sub) mask &= cull[size_t(4 * c + sub) * size_t(ncell) + size_t(m)];
                                        for (int q = g.vnbr_start[m]; q < g.vnbr_start[m + 1]; ++q)
                                        {
                                            const int j = q - g.vnbr_start[m];
                                            if (j < 32 && ((mask >> j) & 1u)) continue;
                                            e.insert(e.end(), &all[4 * size_t(q)], &all[4 * size_t(q)] + 4);
                                        }
                                    }
                                    f[ncell] = int32_t(e.size() / 4);
                                });
                            for (auto& t : pool) t.join();
                            for (int w = 0; w < nc; ++w)
                                if ((rc = uploadRuns(kept[w], firstKept[w], &D.vcone_run[c0 + w], &D.vcone_start[c0 + w]))) return bail(rc);
                        }
                    }
                }
            }
            // ---- per observer: the kept neighbour entries of its cone, packed (DevScene::vobs_*).  All peel-off walks towards an observer
            // have ONE direction, hence one cone and one mask per cell
            int observers = 0;
            for (int i = 0; i < scene->num_instruments; ++i) observers += scene->instruments[i].same_observer_as_preceding ? 0 : 1;
            if (observers <= 4 && scene->num_instruments <= 16 && !pmcTune("PMC_VORO_NO_OBSERVER_LISTS"))
            {
                int k = -1;
                for (int i = 0; i < scene->num_instruments; ++i)
                {
                    const pmc_instrument& ins = scene->instruments[i];
                    if (ins.same_observer_as_preceding)
                    {
                        D.vobs_of_inst[i] = (int8_t)k;
                        continue;
                    }
                    // (memory: at most 32 B per neighbour entry + 64 B per cell and observer; the tables are an acceleration only, so an
                    //  observer whose tables would take more than a quarter of the free device memory goes without, as do the later ones:
                    //  their walks use the cone masks above)
                    {
                        size_t freeBytes = 0, totalBytes = 0;
                        const size_t need = 32 * size_t(g.vnbr_start[ncell]) + 64 * size_t(ncell);
                        if (hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess && need > freeBytes / 4)
                        {
                            fprintf(stderr, "libpmc: per-observer Voronoi tables left out from observer %d on (%.1f GiB each, %.1f GiB free): peel-off walks use the cone masks\n",
                                    k + 1, double(need) / double(1 << 30), double(freeBytes) / double(1 << 30));
                            break;
                        }
                    }
                    ++k;
                    D.vobs_of_inst[i] = (int8_t)k;
                    // the cone of the observer's direction: as voroCone (pmc_walk.inc)
                    const double kx = ins.kobs[0], ky = ins.kobs[1], kz = ins.kobs[2];
                    const int sgn = (kx < 0. ? 1 : 0) | (ky < 0. ? 2 : 0) | (kz < 0. ? 4 : 0);
                    const double ax = std::fabs(kx), ay = std::fabs(ky), az = std::fabs(kz);
                    int pp;
                    if (ax >= ay)
                        pp = ay >= az ? 0 : ax >= az ? 1 : 4;
                    else
                        pp = ax >= az ? 2 : ay >= az ? 3 : 5;
                    int cone = sgn * 6 + pp;
                    if (PMC_VORO_CONES != 48)
                    {
                        const double x = std::fmax(ax, std::fmax(ay, az)), z = std::fmin(ax, std::fmin(ay, az)), y = (ax + ay + az) - x - z;
                        const double a = x - y, b = y - z, c = z;
                        const int subc = a >= b + c ? 0 : b >= a + c ? 1 : c >= a + b ? 2 : 3;
                        cone = cone * 4 + subc;
                    }
                    const bool exactCull = !pmcTune("PMC_VORO_CONE_CULL_ONLY");
                    // pass 1: the kept entries in list order, {site x, y, z, neighbour index} and where every cell's entries start
                    std::vector<double> opair;
                    std::vector<int32_t> first(size_t(ncell) + 1);
                    opair.reserve(4 * size_t(g.vnbr_start[ncell]) * 2 / 3);
                    for (int m = 0; m < ncell; ++m)
                    {
                        const uint32_t mask = cull[size_t(cone) * size_t(ncell) + size_t(m)];
                        first[m] = int32_t(opair.size() / 4);
                        for (int q = g.vnbr_start[m]; q < g.vnbr_start[m + 1]; ++q)
                        {
                            const int j = q - g.vnbr_start[m];
                            if (j < 32 && ((mask >> j) & 1u)) continue;
                            const int mi = g.vnbr_list[q];
                            double e[4] = {0., 0., 0., 0.};
                            if (mi >= 0)
                            {
                                e[0] = g.site[3 * size_t(mi)], e[1] = g.site[3 * size_t(mi) + 1], e[2] = g.site[3 * size_t(mi) + 2];
                                // ONE direction per observer: the candidate test of the walk itself (voroCandidate: n . k > 0, the same
                                // doubles in the same order, no contraction) decides here which sites can ever be the exit -- half of them
                                if (exactCull)
                                {
                                    const double nx = e[0] - g.site[3 * size_t(m)], ny = e[1] - g.site[3 * size_t(m) + 1], nz = e[2] - g.site[3 * size_t(m) + 2];
                                    const double ndotk = nx * kx + ny * ky + nz * kz;
                                    if (!(ndotk > 0)) continue;
                                }
                            }
                            const long long bits = mi;
                            std::memcpy(&e[3], &bits, sizeof(double));
                            opair.insert(opair.end(), e, e + 4);
                        }
                    }
                    first[ncell] = int32_t(opair.size() / 4);
                    if ((rc = uploadRuns(opair, first, &D.vobs_run[k], &D.vobs_start[k]))) return bail(rc);
                }
            }
        }
        D.vblock_n = g.vblock_n;
        if ((rc = ctx->upload(g.vblock_start, nb3 + 1, &D.vblock_start))) return bail(rc);
        if ((rc = ctx->upload(g.vblock_list, size_t(g.vblock_start[nb3]), &D.vblock_list))) return bail(rc);
        D.lds_grid_len = 0;
        D.lmax = 0;
    }
    else
    {
        TreeBuild T;
        if ((rc = buildTree(g, med.number_density, T))) return bail(rc);
        D.lmax = T.lmax;
        D.root_link = T.rootLink;
        if (size_t(T.cellSlots) > PMC_LINK_MAX_INDEX || T.internals.size() > PMC_LINK_MAX_INDEX)
            return bail(fail(PMC_ERR_UNSUPPORTED, "octree with 2^26 cells or nodes or more (26-bit link index)"));
        D.tab_stride_bytes = 8u * uint32_t(T.tabn);
        D.fine_scale[0] = double(1 << T.lmax) / (g.xmax - g.xmin);
        D.fine_scale[1] = double(1 << T.lmax) / (g.ymax - g.ymin);
        D.fine_scale[2] = double(1 << T.lmax) / (g.zmax - g.zmin);
        if ((rc = ctx->upload(T.table.data(), T.table.size(), &D.coord_tab))) return bail(rc);
        if ((rc = ctx->upload(T.leaves.data(), T.leaves.size(), &D.leaves))) return bail(rc);
        if ((rc = ctx->upload(T.cells.data(), T.cells.size(), &D.cell_tab))) return bail(rc);
        if ((rc = ctx->upload(T.internals.data(), T.internals.size(), &D.nodes))) return bail(rc);
        D.coarse_level = T.coarseLevel;
        if ((rc = ctx->upload(T.coarse.data(), T.coarse.size(), &D.coarse_tab))) return bail(rc);
        if ((rc = ctx->upload(T.nbrStart.data(), T.nbrStart.size(), &D.nbr_start))) return bail(rc);
        if ((rc = ctx->upload(T.nbrList.data(), T.nbrList.size(), &D.nbr_list))) return bail(rc);
        if ((rc = ctx->upload(T.cellExt.data(), T.cellExt.size(), &D.cell_ext))) return bail(rc);
        devToCell = T.cellExt;
        D.cell_slots = T.cellSlots;
        // (levels 13-15: 0.2-0.8 MB: not in LDS; the walk reads the six walls of a step from global memory)
        D.tab_in_lds = T.lmax <= 12 ? 1 : 0;
        D.lds_grid_len = D.tab_in_lds ? 3 * T.tabn : 0;
    }

    // ---- medium
    D.num_lambda = med.num_lambda;
    if ((rc = ctx->upload(med.lambda_border, med.num_lambda, &D.lambda_border))) return bail(rc);
    if ((rc = ctx->upload(med.sigma_ext, med.num_lambda, &D.sigma_ext))) return bail(rc);
    if ((rc = ctx->upload(med.sigma_sca, med.num_lambda, &D.sigma_sca))) return bail(rc);
    if ((rc = ctx->upload(med.asymmpar, med.num_lambda, &D.asymmpar))) return bail(rc);
    D.explicit_absorption = scene->options.explicit_absorption ? 1 : 0;
    D.sigma_abs = nullptr;
    if (D.explicit_absorption)
    {
        if (!med.sigma_abs) return bail(fail(PMC_ERR_INVALID, "explicit absorption needs pmc_medium::sigma_abs"));
        if ((rc = ctx->upload(med.sigma_abs, med.num_lambda, &D.sigma_abs))) return bail(rc);
    }
    // several components: every one's densities (in the device numbering of the cells) and dust tables
    D.num_media = scene->num_media > 1 ? scene->num_media : 1;
    std::memset(D.med, 0, sizeof(D.med));
    if (D.num_media > 1)
    {
        const size_t slots = devToCell.empty() ? size_t(g.num_cells) : devToCell.size();
        std::vector<double> dens(slots);
        for (int h = 0; h < D.num_media; ++h)
        {
            const pmc_medium& mh = scene->media[h];
            if (!mh.number_density || !mh.lambda_border || !mh.sigma_ext || !mh.sigma_sca || !mh.asymmpar || mh.num_lambda < 1)
                return bail(fail(PMC_ERR_INVALID, "incomplete medium component " + std::to_string(h)));
            if (D.explicit_absorption && !mh.sigma_abs) return bail(fail(PMC_ERR_INVALID, "explicit absorption needs pmc_medium::sigma_abs of every component"));
            // (without explicit absorption the absorption cross sections are loaded with the others but never used)
            const std::vector<double> noAbs(mh.sigma_abs ? 0 : mh.num_lambda, 0.);
            const double* const sigmaAbs = mh.sigma_abs ? mh.sigma_abs : noAbs.data();
            for (size_t dev = 0; dev < slots; ++dev)
            {
                const int64_t m = devToCell.empty() ? int64_t(dev) : int64_t(devToCell[dev]);
                dens[dev] = m >= 0 ? mh.number_density[m] : 0.;
            }
            DevMedium& M = D.med[h];
            M.num_lambda = mh.num_lambda;
            if ((rc = ctx->upload(dens.data(), dens.size(), &M.density))) return bail(rc);
            if ((rc = ctx->upload(mh.lambda_border, mh.num_lambda, &M.lambda_border))) return bail(rc);
            if ((rc = ctx->upload(mh.sigma_ext, mh.num_lambda, &M.sigma_ext))) return bail(rc);
            if ((rc = ctx->upload(mh.sigma_sca, mh.num_lambda, &M.sigma_sca))) return bail(rc);
            if ((rc = ctx->upload(sigmaAbs, mh.num_lambda, &M.sigma_abs))) return bail(rc);
            if ((rc = ctx->upload(mh.asymmpar, mh.num_lambda, &M.asymmpar))) return bail(rc);
        }
    }
    int walkDoubles = D.lds_grid_len;   // walk kernels and cycle start kernel: the grid tables, at offset 0
    int transDoubles = 0;               // transition and launch kernels: no grid tables
    int launchOnlyDoubles = 0;          // what only the launch kernel stages, behind the regions the two kernels share

    D.force_scattering = scene->options.force_scattering;
    D.voro_prop_checkpoints = pmcTune("PMC_PROP_NO_CHECKPOINTS") == nullptr ? 1 : 0;
    // (bit 0: peel-off walks of voroPeelKernel, bit 1: propagation walks of voroPropKernel)
    D.voro_defer_scan = (scene->grid.kind == PMC_GRID_VORONOI && scene->num_media <= 1 && pmcTune("PMC_VORO_NO_PEEL_KERNEL") == nullptr
                         && pmcTune("PMC_VORO_NO_DEFERRED_SCAN") == nullptr) ? 1 : 0;
    if (scene->grid.kind == PMC_GRID_VORONOI && D.vgen_run && !scene->radiation_field.store && !scene->options.explicit_absorption
        && pmcTune("PMC_VORO_NO_DEFERRED_SCAN") == nullptr)
        D.voro_defer_scan |= 2;
    D.min_weight_reduction = scene->options.min_weight_reduction;
    D.min_scatt_events = scene->options.min_scatt_events;
    D.path_length_bias = scene->options.path_length_bias;

    // ---- sources
    const int numSources = scene->num_sources > 1 ? scene->num_sources : 1;
    if (numSources > PMC_MAX_SOURCES) return bail(fail(PMC_ERR_UNSUPPORTED, "more than " + std::to_string(PMC_MAX_SOURCES) + " sources"));
    if (numSources > 1 && (!scene->sources || !scene->source_first)) return bail(fail(PMC_ERR_INVALID, "source tables missing"));
    D.num_sources = numSources;
    for (int si = 0; si < numSources; ++si)
    {
        const pmc_source& src = numSources > 1 ? scene->sources[si] : scene->source;
        DevSource& Q = D.src[si];
        D.src_first[si] = numSources > 1 ? scene->source_first[si] : 0;
        Q.source_kind = src.kind;
        std::memcpy(Q.src_pos, src.position, sizeof(Q.src_pos));
        Q.reff = src.reff;
        Q.sersic_n = src.sersic_n;
        std::memcpy(Q.src_box, src.box, sizeof(Q.src_box));
        Q.packet_luminosity = src.packet_luminosity;
        Q.lambda_mode = src.lambda_mode;
        Q.num_oligo = src.num_oligo;
        Q.lambda_bias = src.lambda_bias;
        Q.num_sed = src.num_sed;
        Q.bias_kind = src.bias_kind;
        Q.bias_min = src.bias_min;
        Q.bias_max = src.bias_max;
        Q.sed_kind = src.sed_kind;
        Q.sed_f1 = src.sed_f1;
        Q.sed_f2 = src.sed_f2;
        Q.sed_ltot = src.sed_ltot;
        Q.angular_kind = src.kind == PMC_SOURCE_POINT ? src.angular_kind : PMC_ANGULAR_ISOTROPIC;
        std::memcpy(Q.angular_axis, src.angular_axis, sizeof(Q.angular_axis));
        Q.angular_cos_delta = src.angular_cos_delta;
        if (Q.angular_kind < PMC_ANGULAR_ISOTROPIC || Q.angular_kind > PMC_ANGULAR_NETZER) return bail(fail(PMC_ERR_UNSUPPORTED, "unsupported angular distribution"));
        if (Q.angular_kind == PMC_ANGULAR_NETZER && !D.netzer_cos)
        {
            // NetzerAngularDistribution::setupSelfBefore (NetzerAngularDistribution.cpp:12-30; NR::buildLinearGrid, NR.hpp:203-209)
            const int n = PMC_NETZER_POINTS;
            std::vector<double> ct(n + 1), X(n + 1);
            const double dx = (+1. - -1.) / n;
            for (int i = 0; i <= n; i++) ct[i] = -1. + i * dx;
            X[0] = 0;
            for (int i = 1; i < n; i++)
            {
                const double c = ct[i];
                const double sign = c > 0 ? 1. : -1;
                X[i] = (1. / 2.) + (2. / 7.) * c * c * c + sign * (3. / 14.) * c * c;
            }
            X[n] = 1.;
            if ((rc = ctx->upload(ct.data(), ct.size(), &D.netzer_cos))) return bail(rc);
            if ((rc = ctx->upload(X.data(), X.size(), &D.netzer_X))) return bail(rc);
        }
        if (src.kind == PMC_SOURCE_SERSIC)
        {
            if (src.sersic_n < 2) return bail(fail(PMC_ERR_INVALID, "Sersic source without tables"));
            if ((rc = ctx->upload(src.sersic_s, src.sersic_n, &Q.sersic_s))) return bail(rc);
            if ((rc = ctx->upload(src.sersic_M, src.sersic_n, &Q.sersic_M))) return bail(rc);
            if (numSources == 1) launchOnlyDoubles += 2 * src.sersic_n;
        }
        else if (src.kind != PMC_SOURCE_POINT && src.kind != PMC_SOURCE_UNIFORM_BOX && src.kind != PMC_SOURCE_EXP_DISK
                 && src.kind != PMC_SOURCE_PLUMMER)
            return bail(fail(PMC_ERR_UNSUPPORTED, "unsupported source kind"));
        if (src.lambda_mode == PMC_LAMBDA_OLIGO)
        {
            if (src.num_oligo < 1) return bail(fail(PMC_ERR_INVALID, "oligochromatic source without wavelengths"));
            if ((rc = ctx->upload(src.oligo_lambda, src.num_oligo, &Q.oligo_lambda))) return bail(rc);
            if ((rc = ctx->upload(src.oligo_weight, src.num_oligo, &Q.oligo_weight))) return bail(rc);
        }
        else if (src.lambda_mode == PMC_LAMBDA_TABULATED)
        {
            if (src.num_sed < 2) return bail(fail(PMC_ERR_INVALID, "tabulated source without SED table"));
            if ((rc = ctx->upload(src.sed_lambda, src.num_sed, &Q.sed_lambda))) return bail(rc);
            if ((rc = ctx->upload(src.sed_p, src.num_sed, &Q.sed_p))) return bail(rc);
            if ((rc = ctx->upload(src.sed_P, src.num_sed, &Q.sed_P))) return bail(rc);
        }
        else
            return bail(fail(PMC_ERR_UNSUPPORTED, "unsupported wavelength sampling mode"));
    }
    D.src_first[numSources] = numSources > 1 ? scene->source_first[numSources] : ~0ull;
    for (int si = 1; si < numSources; ++si)
        if (D.src[si].lambda_mode != D.src[0].lambda_mode) return bail(fail(PMC_ERR_INVALID, "sources with different wavelength regimes"));
    // one wavelength for every history: its dust properties are found here once, as launchHistory finds them (DustMix::indexForLambda:
    // NR::locateClip on the index borders)
    auto sourceOf = [&](int si) -> const pmc_source& { return numSources > 1 ? scene->sources[si] : scene->source; };
    D.mono = (sourceOf(0).lambda_mode == PMC_LAMBDA_OLIGO && pmcTune("PMC_NO_MONO") == nullptr) ? 1 : 0;
    for (int si = 0; si < numSources && D.mono; ++si)
        if (sourceOf(si).num_oligo != 1 || sourceOf(si).oligo_lambda[0] != sourceOf(0).oligo_lambda[0]) D.mono = 0;
    D.mono_lambda = D.mono_ext = D.mono_sca = D.mono_asym = D.mono_abs = 0.;
    if (D.mono)
    {
        const double lambda = sourceOf(0).oligo_lambda[0];
        int il = 0;
        if (!(lambda < med.lambda_border[0]))
        {
            int jl = -1, ju = med.num_lambda - 1;
            while (ju - jl > 1)
            {
                const int jm = (ju + jl) >> 1;
                if (lambda < med.lambda_border[jm])
                    ju = jm;
                else
                    jl = jm;
            }
            il = jl;
        }
        D.mono_lambda = lambda;
        D.mono_ext = med.sigma_ext[il], D.mono_sca = med.sigma_sca[il], D.mono_asym = med.asymmpar[il];
        if (D.explicit_absorption) D.mono_abs = med.sigma_abs[il];
        for (int h = 0; h < D.num_media && D.num_media > 1; ++h)
        {
            const pmc_medium& mh = scene->media[h];
            int ih = 0;
            if (!(lambda < mh.lambda_border[0]))
            {
                int jl = -1, ju = mh.num_lambda - 1;
                while (ju - jl > 1)
                {
                    const int jm = (ju + jl) >> 1;
                    if (lambda < mh.lambda_border[jm])
                        ju = jm;
                    else
                        jl = jm;
                }
                ih = jl;
            }
            D.med[h].mono_ext = mh.sigma_ext[ih], D.med[h].mono_sca = mh.sigma_sca[ih], D.med[h].mono_abs = mh.sigma_abs ? mh.sigma_abs[ih] : 0.;
            D.med[h].mono_asym = mh.asymmpar[ih];
        }
    }

    // ---- instruments and frame layout
    D.num_instruments = scene->num_instruments;
    D.lds_sed_off = transDoubles;
    int sedDoubles = 0;
    D.any_stats = 0;
    D.stat_acc_records = 0;
    for (int i = 0; i < scene->num_instruments; ++i)
    {
        const pmc_instrument& I = scene->instruments[i];
        DevInstrument& d = D.inst[i];
        d.kx = I.kobs[0], d.ky = I.kobs[1], d.kz = I.kobs[2];
        // RN(1/k) as PathSegmentGenerator's users divide by it; an axis with |k| <= 1e-15 is ignored (TreeSpatialGrid.cpp:160-175)
        const double ignored = std::nan("");
        d.ikx = std::fabs(d.kx) > 1e-15 ? 1. / d.kx : ignored;
        d.iky = std::fabs(d.ky) > 1e-15 ? 1. / d.ky : ignored;
        d.ikz = std::fabs(d.kz) > 1e-15 ? 1. / d.kz : ignored;
        d.sgn = (d.kx < 0. ? 1u : 0u) | (d.ky < 0. ? 2u : 0u) | (d.kz < 0. ? 4u : 0u);
        d.costheta = I.costheta, d.sintheta = I.sintheta, d.cosphi = I.cosphi, d.sinphi = I.sinphi;
        d.cosomega = I.cosomega, d.sinomega = I.sinomega;
        d.xpmin = I.xpmin, d.xpsiz = I.xpsiz, d.ypmin = I.ypmin, d.ypsiz = I.ypsiz;
        d.nxp = I.nxp, d.nyp = I.nyp;
        d.same_observer = I.same_observer_as_preceding;
        d.include_sed = I.include_flux_density;
        d.include_ifu = I.include_surface_brightness;
        d.record_components = I.record_components;
        d.num_levels = I.num_scattering_levels;
        d.record_stats = I.record_statistics;
        d.aperture_r2 = I.aperture_radius2;
        if (!(I.redshift >= 0.)) return bail(fail(PMC_ERR_INVALID, "negative instrument redshift"));
        d.zp1 = 1. + I.redshift;
        d.num_lambda = I.num_lambda;
        d.num_border = I.num_border;
        if ((rc = ctx->upload(I.border, I.num_border, &d.border))) return bail(rc);
        if ((rc = ctx->upload(I.ellv, I.num_border + 1, &d.ellv))) return bail(rc);
        d.mono_ell = -1;
        if (D.mono)
        {
            // (DisjointWavelengthGrid::bin of the redshifted wavelength, as launchHistory finds it: upper_bound on the borders)
            const double lambdaObs = D.mono_lambda * d.zp1;
            int lo = 0, hi = I.num_border;
            while (lo < hi)
            {
                const int mid = (lo + hi) >> 1;
                if (lambdaObs < I.border[mid])
                    hi = mid;
                else
                    lo = mid + 1;
            }
            d.mono_ell = I.ellv[lo];
        }
        pmc_frame_layout L;
        ctx->frameSize = pmc_layout_compute(scene, i, &L);
        d.sed_offset = L.sed_offset, d.ifu_offset = L.ifu_offset, d.wsed_offset = L.wsed_offset, d.wifu_offset = L.wifu_offset;
        d.npix = L.npix;
        d.num_components = (int32_t)L.num_components;
        d.sed_lds_offset = sedDoubles;
        if (d.include_sed) sedDoubles += (d.num_components + (d.record_stats ? 5 : 0)) * d.num_lambda;
        if (d.record_stats) D.any_stats = 1;
        d.stat_acc_offset = D.stat_acc_records;
        if (d.record_stats && d.include_ifu) D.stat_acc_records += d.npix * d.num_lambda;
    }
    D.stat_acc = nullptr;
    if (D.stat_acc_records && (rc = ctx->allocate<double>(size_t(D.stat_acc_records) * 8, &D.stat_acc, true))) return bail(rc);
    // ---- radiation field table
    const pmc_radiation_field& RF = scene->radiation_field;
    D.rf_store = RF.store ? 1 : 0;
    ctx->rfSize = 0;
    if (D.rf_store)
    {
        if (!scene->options.force_scattering)
            return bail(fail(PMC_ERR_INVALID, "the radiation field can only be stored with forced scattering (Configuration.cpp:476-482)"));
        if (RF.num_lambda < 1 || RF.num_border < 1 || !RF.border || !RF.ellv)
            return bail(fail(PMC_ERR_INVALID, "radiation field wavelength grid is missing"));
        D.rf_num_lambda = RF.num_lambda;
        D.rf_num_border = RF.num_border;
        if ((rc = ctx->upload(RF.border, RF.num_border, &D.rf_border))) return bail(rc);
        if ((rc = ctx->upload(RF.ellv, RF.num_border + 1, &D.rf_ellv))) return bail(rc);
        ctx->rfSize = int64_t(scene->grid.num_cells) * RF.num_lambda;
        if ((rc = ctx->allocate<double>(size_t(ctx->rfSize), &D.rf, true))) return bail(rc);
    }
    D.lds_sed_len = sedDoubles;
    transDoubles += sedDoubles;
    D.lds_hot_off = transDoubles;
    transDoubles += 2 * 256;  // hot-bin table (pmc_transition.inc HOT_BINS keys + values)
    D.lds_sort_off = transDoubles;
    transDoubles += (64 + 2 * 1024 + 8) / 2;  // integer scratch: regrouping arrays and list-append counters
    D.lds_total_transition = transDoubles;
    // launch kernel: Sersic tables, then the index borders of the dust mix (DustMix::_lambdav: the binary search of every new
    // history's wavelength) if they fit in 64 KiB
    D.lds_src_off = transDoubles;
    D.lds_dust_off = transDoubles + launchOnlyDoubles;
    D.dust_in_lds = med.num_lambda <= 8192;
    if (D.dust_in_lds) launchOnlyDoubles += med.num_lambda;
    D.lds_total_launch = transDoubles + launchOnlyDoubles;
    D.lds_total_walk = walkDoubles;
    ctx->walkLds = size_t(walkDoubles) * sizeof(double);
    ctx->transitionLds = size_t(transDoubles) * sizeof(double);
    ctx->launchLds = size_t(D.lds_total_launch) * sizeof(double);
    if (ctx->walkLds > 160 * 1024 || ctx->launchLds > 160 * 1024)
        return bail(fail(PMC_ERR_UNSUPPORTED, "scene tables need more than 160 KiB of LDS"));
    if (pmcConfigureKernels(ctx->walkLds, ctx->launchLds) != hipSuccess)
        return bail(fail(PMC_ERR_DEVICE, "hipFuncSetAttribute failed"));

    // ---- launch geometry of the persistent walk kernel: as many workgroups as stay resident
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return bail(fail(PMC_ERR_DEVICE, "hipGetDeviceProperties failed"));
    ctx->numCU = prop.multiProcessorCount;
    ctx->block = 256;
    ctx->wide = D.grid_kind == PMC_GRID_OCTREE && D.lmax > 10;
    {
        // persistent walk kernels: as many workgroups as stay resident (the transition / launch kernels of the other
        // slot group get the CUs between generations)
        int perCU = pmcWalkBlocksPerCU(D.grid_kind, D.grid_kind == PMC_GRID_OCTREE ? 2 : 0, ctx->wide,
                                       D.grid_kind == PMC_GRID_OCTREE ? pmcPropBlock() : ctx->block, ctx->walkLds);
        if (perCU < 1) perCU = 1;
        // (octree propagation kernel: ONE workgroup of 256 lanes per CU.  The walk kernels are bound by the memory system's
        // rate of random gathers, which falls when too many of them are in flight -- 32 MB table, 8 / 16 / 32 waves per CU:
        // 1.9 / 0.94 / 0.83e11 records/s, profiles/microbench/gather_modes_mi355x.txt -- and the kernels of three slot groups
        // overlap; 1 / 2 / 3 per CU: 697 / 720 / 756 ms per 1e8 packets)
        if (const char* env = pmcTune("PMC_WALK_BLOCKS_PER_CU")) perCU = std::min(perCU, std::max(1, atoi(env)));  // tuning aid
        else perCU = std::min(perCU, D.grid_kind == PMC_GRID_OCTREE ? 1 : 3);
        ctx->grid = ctx->numCU * perCU;
        if (D.grid_kind == PMC_GRID_OCTREE)
        {
            int peelPerCU = pmcWalkBlocksPerCU(D.grid_kind, 1, ctx->wide, pmcPeelBlock(), ctx->walkLds);
            // (one workgroup of 512 lanes per CU: with the pipelined step more resident waves only queue up in the memory
            // system -- 1 / 2 / 3 per CU: 97 / 111 / 124 ms per 5e7 packets, profiles/README.md)
            peelPerCU = std::min(std::max(peelPerCU, 1), 1);
            if (const char* env = pmcTune("PMC_PEEL_BLOCKS_PER_CU")) peelPerCU = std::max(1, atoi(env));
            ctx->peelGrid = ctx->numCU * peelPerCU;
        }
    }

    // ---- packet slots
    // 24 Mi histories in flight (about 1.7 KB of state per slot with one statistics-recording instrument: 40 GB of the 288).  A segment of
    // 1e8 packets then runs 83 generations instead of the 177 of 8 Mi slots -- fewer launches, fewer ragged kernel tails -- and gains 3-7 %
    // (8 / 12 / 16 / 24 / 32 Mi: 2.00 / 2.05 / 2.05 / 2.07 / 2.07e8 packets/s, profiles/sweeps/r05_d3_slots.txt)
    int64_t slots = 24 * 1024 * 1024;
    if (const char* env = getenv("PMC_NUM_SLOTS"))
    {
        slots = std::max<int64_t>(1024, atoll(env));
        ctx->slotsConfigured = true;
    }
    ctx->numSlots = slots;

    // ---- outputs
    if ((rc = ctx->allocate<double>(ctx->frameSize, &ctx->frames, true))) return bail(rc);
    D.frames = ctx->frames;
    if ((rc = ctx->allocate<unsigned long long>(PMC_NUM_COUNTERS, &D.counters, true))) return bail(rc);
    *out = ctx;
    return PMC_OK;
}

int pmc_set_launch(pmc_ctx* ctx, int32_t block, int32_t grid)
{
    if (!ctx) return fail(PMC_ERR_INVALID, "null context");
    if (block > 0)
    {
        if (block % 64 || block > 256) return fail(PMC_ERR_INVALID, "block must be a multiple of 64 and at most 256");
        ctx->block = block;
    }
    if (grid > 0) ctx->grid = grid;
    return PMC_OK;
}

int pmc_set_num_slots(pmc_ctx* ctx, int64_t num_slots)
{
    if (!ctx) return fail(PMC_ERR_INVALID, "null context");
    if (num_slots < 64 || num_slots > (int64_t(1) << 30)) return fail(PMC_ERR_INVALID, "num_slots out of range");
    ctx->numSlots = num_slots;
    ctx->slotsConfigured = true;
    return PMC_OK;
}

int pmc_bind_frames(pmc_ctx* ctx, double* device_ptr, int64_t num_doubles)
{
    if (!ctx || !device_ptr) return fail(PMC_ERR_INVALID, "null argument");
    if (num_doubles != ctx->frameSize) return fail(PMC_ERR_INVALID, "frame buffer size mismatch");
    ctx->frames = device_ptr;
    ctx->dev.frames = device_ptr;
    ctx->sceneDirty = true;
    return PMC_OK;
}

int pmc_clear_frames(pmc_ctx* ctx)
{
    if (!ctx) return fail(PMC_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemsetAsync(ctx->frames, 0, size_t(ctx->frameSize) * sizeof(double), ctx->stream));
    return PMC_OK;
}

int pmc_run_primary(pmc_ctx* ctx, uint64_t first, uint64_t count, uint64_t seed)
{
    if (!ctx) return fail(PMC_ERR_INVALID, "null context");
    if (count == 0) return PMC_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    DevScene& D = ctx->dev;
    const int64_t want = std::min<int64_t>(ctx->numSlots, (int64_t)std::min<uint64_t>(count, uint64_t(1) << 30));
    bool grow = want > ctx->allocatedSlots;
    if (grow && ctx->allocatedSlots > 0 && ctx->steppedDownFree)
    {
        // (a default pool that has stepped down: ask again only when more memory is free than there was then -- not a
        // reallocation per segment)
        size_t freeBytes = 0, totalBytes = 0;
        grow = hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess && freeBytes > ctx->steppedDownFree + ctx->steppedDownFree / 4;
    }
    if (grow)
    {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->steppedDownFree = 0;
        int rc = allocateSlots(ctx, want);
        if (rc) return rc;
    }
    // (the default steps down where the device memory is short: allocateSlots)
    const int numSlots = (int)std::min<int64_t>(want, ctx->allocatedSlots);
    if (ctx->sceneDirty)
    {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(pmcUploadScene(ctx->slot, &D, ctx->stream));
        ctx->sceneDirty = false;
    }
    hipStream_t st = ctx->stream;
    unsigned long long* ctr = D.counters;
    float walkMs = 0, transMs = 0, peelMs = 0, propMs = 0;
    const bool serialWalks = pmcTune("PMC_SERIAL_WALKS") != nullptr;  // tuning aid: peel-off and propagation kernels one after the other
    const bool genDump = pmcTune("PMC_GEN_DUMP") != nullptr;  // tuning aid: live slots and kernel times of every generation
    int generations = 0;
    // ---- slot groups: group g owns the slots [base[g], base[g] + size[g]) and the stream groupStream[g].  The
    // generations of different groups are independent (histories come from one shared cursor), so while the host
    // waits for one group the other groups' kernels keep the device busy: the tail of a walk kernel and the
    // latency-bound transition kernel overlap with the walk kernel of another group.
    int G = ctx->numGroups;
    if (numSlots < G * 65536) G = 1;
    {
        // (Voronoi: the walk kernel is nine tenths of the step, and its walks run as ONE stream in tile order: a second and third group would
        // put two more streams in flight next to it and triple the cells the L2s have to hold -- 5e7 packets: 2.18 / 2.13 / 2.04e7 packets/s
        // with one / two / three groups)
        int observers = 0;
        for (int i = 0; i < D.num_instruments; ++i) observers += D.inst[i].same_observer ? 0 : 1;
        if (D.grid_kind == PMC_GRID_VORONOI && observers <= PMC_SORT_OBS && !ctx->groupsConfigured && pmcTune("PMC_NO_PEEL_SORT") == nullptr) G = 1;
    }
    int base[PMC_MAX_GROUPS], size[PMC_MAX_GROUPS];
    bool active[PMC_MAX_GROUPS], haveWalk[PMC_MAX_GROUPS];
    // sparse generations (the end of a segment, when no history is left to launch): the cycle start kernel compacts the live
    // slots of the group into a list, and the kernels of the next generation run over the list with as many workgroups as it
    // needs -- their time then follows the live histories, not the size of the slot pool (a third of the generations of a
    // 1e8-packet segment run fewer than a tenth of the slots).  Such a generation is walks -> transition -> cycle start: the
    // transition kernel retires the histories that end (nothing is left to launch into their slots), the cycle start kernel
    // writes the list of the generation after it into the other half of TaskArrays::liveList.
    bool listBuilt[PMC_MAX_GROUPS] = {false, false, false, false};
    // sorted peel-off records (pmc_device.h PeelRec): an octree whose peel-off kernel runs with task queues, ONE observer (its records
    // are written by the cycle start kernel in slot order, sorted by detector tile, and read in tile order by the peel-off kernel)
    bool peelSorted[PMC_MAX_GROUPS] = {false, false, false, false};
    int numSortObs = 0, sortObs[PMC_SORT_OBS] = {0, 0, 0, 0};
    const bool xcdAffinity = pmcTune("PMC_NO_XCD_AFFINITY") == nullptr;
    if (!ctx->xcdCursors)
    {
        int rc;
        // (per group PMC_SORT_OBS + 1 sets of eight: set 0 the generic kernel's stream, 1 + k the Voronoi peel-off kernel of sorted observer k, and the
        // octree's peel-off kernels sets 0 .. PMC_SORT_OBS - 1; one more set behind them all that is never written: a count of zero)
        if ((rc = ctx->allocate<unsigned long long>((size_t(PMC_MAX_GROUPS) * (PMC_SORT_OBS + 1) + 1) * 8, &ctx->xcdCursors, true, &ctx->rfAllocations))) return rc;
    }
    const auto cursorSet = [&](int g, int k) { return ctx->xcdCursors + (size_t(g) * (PMC_SORT_OBS + 1) + size_t(k)) * 8; };
    const unsigned long long* const zeroCount = ctx->xcdCursors + size_t(PMC_MAX_GROUPS) * (PMC_SORT_OBS + 1) * 8;
    // Voronoi, one medium component: the peel-off walks towards an observer that has a table of runs go through a kernel of their own
    // (a switch set after pmc_create: the generic kernel knows a walk whose first cell is still to be scanned as well)
    const bool voroPeelKernels = D.grid_kind == PMC_GRID_VORONOI && D.num_media <= 1 && pmcTune("PMC_VORO_NO_PEEL_KERNEL") == nullptr;
    // ... and the propagation walks of the plain flavour, on the table of runs with all neighbours (when pmc_create built it)
    const bool voroPropKernel = D.grid_kind == PMC_GRID_VORONOI && D.num_media <= 1 && D.vgen_run && !D.rf_store && !D.explicit_absorption
                                && pmcTune("PMC_VORO_NO_PROP_KERNEL") == nullptr;
    const bool octree = D.grid_kind == PMC_GRID_OCTREE;
    if (pmcTune("PMC_NO_PEEL_SORT") == nullptr && (!octree || pmcPeelHasQueues((ctx->wide ? 1 : 0) | (D.num_media > 1 ? 2 : 0), ctx->walkLds)))
    {
        int observers = 0;
        for (int i = 0; i < D.num_instruments; ++i)
            if (!D.inst[i].same_observer)
            {
                if (observers < PMC_SORT_OBS) sortObs[observers] = i;
                ++observers;
            }
        if (observers <= PMC_SORT_OBS) numSortObs = observers;  // (more observers than that: all of them from the task arrays)
    }
    // (Cartesian, Voronoi: one more list through the same sort -- the slots' PROPAGATION walks by the sign octant of their direction: with the
    // XCD affinity of the walk stream the L2 of an XCD then sees the propagation walks of about one octant)
    int propSortIndex = (!octree && numSortObs > 0 && numSortObs < PMC_SORT_OBS && pmcTune("PMC_NO_PROP_SORT") == nullptr) ? numSortObs : -1;
    int numSortLists = numSortObs + (propSortIndex >= 0 ? 1 : 0);
    int listHalf[PMC_MAX_GROUPS] = {0, 0, 0, 0};  // the half of liveList that holds the group's current list
    int listTasksPerLane = 1;  // walks per lane that size the walk kernels' grids in a sparse generation
    if (const char* env = pmcTune("PMC_LIST_TASKS_PER_LANE")) listTasksPerLane = std::max(1, atoi(env));
    const bool sparseLists = D.grid_kind == PMC_GRID_OCTREE && pmcTune("PMC_NO_LIVE_LISTS") == nullptr;
    {
        const int per = ((numSlots / G) + PMC_TRANSITION_ALIGN - 1) / PMC_TRANSITION_ALIGN * PMC_TRANSITION_ALIGN;
        for (int g = 0; g < G; ++g)
        {
            base[g] = std::min(numSlots, g * per);
            size[g] = std::min(per, numSlots - base[g]);
            active[g] = size[g] > 0;
            haveWalk[g] = false;
        }
    }
    for (int g = 0; g < G && numSortObs > 0; ++g)
    {
        const int padded = (size[g] + 4095) / 4096 * 4096;
        if (ctx->peelCap[g] >= padded && (octree ? (void*)ctx->peelRec[g][numSortObs - 1] : (void*)ctx->peelList[g][numSortLists - 1])) continue;
        HIP_TRY(hipDeviceSynchronize());
        // (a group that grows, or more observers than last time: the old buffers go first)
        for (int k = 0; k < PMC_SORT_OBS; ++k)
            for (void* old : {static_cast<void*>(ctx->peelRec[g][k]), static_cast<void*>(ctx->peelList[g][k])})
                if (old)
                {
                    hipFree(old);
                    auto at = std::find(ctx->rfAllocations.begin(), ctx->rfAllocations.end(), old);
                    if (at != ctx->rfAllocations.end()) ctx->rfAllocations.erase(at);
                }
        for (int k = 0; k < PMC_SORT_OBS; ++k) ctx->peelRec[g][k] = nullptr, ctx->peelList[g][k] = nullptr;
        ctx->peelCap[g] = 0;
        // (no room for the records: the peel-off walks run from the task arrays, in slot order)
        size_t freeBytes = 0, totalBytes = 0;
        if (hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess
            && size_t(numSortLists) * (size_t(padded) * sizeof(PeelRec) + pmcPeelSortTempBytes()) + (size_t(1) << 30) > freeBytes)
        {
            numSortObs = 0, propSortIndex = -1, numSortLists = 0;
            break;
        }
        int rc;
        for (int k = 0; k < numSortLists; ++k)
        {
            if (octree && (rc = ctx->allocate<PeelRec>(padded, &ctx->peelRec[g][k], false, &ctx->rfAllocations))) return rc;
            if (!octree && (rc = ctx->allocate<int32_t>(padded, &ctx->peelList[g][k], false, &ctx->rfAllocations))) return rc;
            if (!ctx->peelTemp[g][k])
            {
                uint8_t* t = nullptr;
                if ((rc = ctx->allocate<uint8_t>(pmcPeelSortTempBytes(), &t, false, &ctx->rfAllocations))) return rc;
                ctx->peelTemp[g][k] = t;
            }
        }
        ctx->peelCap[g] = padded;
    }
    // ---- radiation field on an octree: the contributions of a generation go to a log per slot group (pmc_device.h RfLogArgs),
    // which is partitioned by key range and summed after the generation.  128 entries per slot (config 2: 60 per propagation
    // walk on average); a wave that finds the log full falls back to atomic adds into the table.
    const int64_t rfSize = ctx->rfSize;
    // (tables beyond 2^26 entries have more partitions than the counting sort's LDS histogram holds: atomics)
    // (the keys of the log count cells in the device numbering: cell_slots of them, padding included)
    const int64_t rfKeys = D.grid_kind == PMC_GRID_OCTREE ? int64_t(D.cell_slots) * D.rf_num_lambda : rfSize;
    const int64_t rfParts = (rfKeys + (int64_t(1) << PMC_RF_BUCKET_BITS) - 1) >> PMC_RF_BUCKET_BITS;
    const bool rfLogged = D.rf_store && D.grid_kind == PMC_GRID_OCTREE && rfParts <= pmcRfMaxParts() && pmcTune("PMC_RF_ATOMICS") == nullptr;
    if (D.rf_store && D.grid_kind == PMC_GRID_OCTREE && rfParts > pmcRfMaxParts())
    {
        // (a table beyond 2^26 entries: one atomic per contribution, several times slower -- said once, not silently)
        static std::atomic<bool> said{false};
        if (!said.exchange(true))
            fprintf(stderr, "libpmc: the radiation field table has %lld entries, more than the log's counting sort partitions (%d x %d): contributions are added atomically\n",
                    (long long)rfKeys, pmcRfMaxParts(), 1 << PMC_RF_BUCKET_BITS);
    }
    const int rfBuckets = rfLogged ? int(rfParts) : 0;
    const uint32_t rfPadKey = uint32_t(rfBuckets) << PMC_RF_BUCKET_BITS;
    if (rfLogged)
        for (int g = 0; g < G; ++g)
        {
            unsigned long long perSlot = 128ull;
            if (const char* env = pmcTune("PMC_RF_LOG_PER_SLOT")) perSlot = std::max(1, atoi(env));  // (tests: a log that overflows)
            // (positions in the partitioned log are 32-bit: at most 2^31 - 1 entries, in whole chunks; a wave that finds the log
            // full adds its contributions atomically)
            const unsigned long long want = std::min<unsigned long long>(
                std::max<unsigned long long>(((unsigned long long)size[g] * perSlot + PMC_RF_LOG_CHUNK - 1) / PMC_RF_LOG_CHUNK, 1ull) * PMC_RF_LOG_CHUNK,
                (0x7FFFFFFFull / PMC_RF_LOG_CHUNK) * PMC_RF_LOG_CHUNK);
            if (want <= ctx->rfCap[g]) continue;
            HIP_TRY(hipDeviceSynchronize());
            // (a log that grows: the old buffers go first)
            auto release = [&](void* p) {
                if (!p) return;
                hipFree(p);
                auto at = std::find(ctx->rfAllocations.begin(), ctx->rfAllocations.end(), p);
                if (at != ctx->rfAllocations.end()) ctx->rfAllocations.erase(at);
            };
            for (int k = 0; k < 2; ++k)
            {
                release(ctx->rfKeys[g][k]), release(ctx->rfVals[g][k]);
                ctx->rfKeys[g][k] = nullptr, ctx->rfVals[g][k] = nullptr;
            }
            ctx->rfCap[g] = 0;
            // no room for the log (24 bytes per entry): the group's contributions go to the table as atomics (cap 0)
            size_t freeBytes = 0, totalBytes = 0;
            bool room = hipMemGetInfo(&freeBytes, &totalBytes) != hipSuccess || size_t(want) * 24 + (size_t(1) << 30) <= freeBytes;
            for (int k = 0; k < 2 && room; ++k)
                room = ctx->allocate<uint32_t>(want, &ctx->rfKeys[g][k], false, &ctx->rfAllocations) == PMC_OK
                       && ctx->allocate<double>(want, &ctx->rfVals[g][k], false, &ctx->rfAllocations) == PMC_OK;
            if (!room)
            {
                for (int k = 0; k < 2; ++k)
                {
                    release(ctx->rfKeys[g][k]), release(ctx->rfVals[g][k]);
                    ctx->rfKeys[g][k] = nullptr, ctx->rfVals[g][k] = nullptr;
                }
                continue;
            }
            ctx->rfCap[g] = want;
        }
    if (rfLogged && ctx->rfTempBytes < pmcRfTempBytes(rfBuckets))
    {
        HIP_TRY(hipDeviceSynchronize());
        for (int h = 0; h < PMC_MAX_GROUPS; ++h)
            if (ctx->rfTemp[h])
            {
                hipFree(ctx->rfTemp[h]);
                auto at = std::find(ctx->rfAllocations.begin(), ctx->rfAllocations.end(), ctx->rfTemp[h]);
                if (at != ctx->rfAllocations.end()) ctx->rfAllocations.erase(at);
                ctx->rfTemp[h] = nullptr;
            }
        ctx->rfTempBytes = pmcRfTempBytes(rfBuckets);
    }
    if (rfLogged)
        for (int g = 0; g < G; ++g)
            if (!ctx->rfTemp[g])
            {
                uint8_t* t = nullptr;
                int rc;
                if ((rc = ctx->allocate<uint8_t>(std::max<size_t>(ctx->rfTempBytes, 16), &t, false, &ctx->rfAllocations))) return rc;
                ctx->rfTemp[g] = t;
            }
    // the log of group g (n entries claimed) -> table, on the group's stream
    auto rfFlush = [&](int g, unsigned long long claimed) -> int {
        const unsigned long long n = std::min(claimed, ctx->rfCap[g]);
        if (!rfLogged || n == 0) return PMC_OK;
        hipStream_t sg = ctx->groupStream[g];
        HIP_TRY(pmcLaunchRfFlush(ctx->slot, ctx->rfKeys[g][0], ctx->rfVals[g][0], ctx->rfKeys[g][1], ctx->rfVals[g][1], n, rfBuckets, ctx->rfTemp[g], ctx->numCU, sg));
        return PMC_OK;
    };
    // ---- statistics: the contributions of ended histories go to a log per slot group (pmc_device.h StatLogArgs), which is partitioned by
    // record range and summed in LDS when it has filled up, and at the end of the segment
    const int statBits = pmcStatBucketBits();
    const int64_t statParts = (D.stat_acc_records + (int64_t(1) << statBits) - 1) >> statBits;
    const bool statLogged = D.any_stats && D.stat_acc_records > 0 && statParts <= pmcRfMaxParts() && pmcTune("PMC_STAT_ATOMICS") == nullptr;
    if (statLogged)
        for (int g = 0; g < G; ++g)
        {
            // (3.7 entries per history on configs[1]: the log of a group holds a segment of 1e8 packets; it is flushed when half full)
            unsigned long long want = (128ull << 20);
            if (const char* env = pmcTune("PMC_STAT_LOG_ENTRIES")) want = std::max(1, atoi(env));  // (tests: a log that overflows)
            want = std::max<unsigned long long>((want + PMC_RF_LOG_CHUNK - 1) / PMC_RF_LOG_CHUNK, 1ull) * PMC_RF_LOG_CHUNK;
            if (ctx->statCap[g] == want && ctx->statTemp[g] && ctx->statChunkFill[g]) continue;
            HIP_TRY(hipDeviceSynchronize());
            auto release = [&](void* p) {
                if (!p) return;
                hipFree(p);
                auto at = std::find(ctx->rfAllocations.begin(), ctx->rfAllocations.end(), p);
                if (at != ctx->rfAllocations.end()) ctx->rfAllocations.erase(at);
            };
            for (int k = 0; k < 2; ++k)
            {
                release(ctx->statKeys[g][k]), release(ctx->statVals[g][k]);
                ctx->statKeys[g][k] = nullptr, ctx->statVals[g][k] = nullptr;
            }
            release(ctx->statChunkFill[g]);
            ctx->statChunkFill[g] = nullptr;
            ctx->statCap[g] = 0;
            size_t freeBytes = 0, totalBytes = 0;
            bool room = hipMemGetInfo(&freeBytes, &totalBytes) != hipSuccess || size_t(want) * 24 + (size_t(2) << 30) <= freeBytes;
            room = room && ctx->allocate<uint32_t>(want / PMC_RF_LOG_CHUNK, &ctx->statChunkFill[g], false, &ctx->rfAllocations) == PMC_OK;
            if (room && !ctx->statWaveBase[g])
                room = ctx->allocate<unsigned long long>(PMC_STAT_LOG_WAVES, &ctx->statWaveBase[g], true, &ctx->rfAllocations) == PMC_OK
                       && ctx->allocate<uint32_t>(PMC_STAT_LOG_WAVES, &ctx->statWaveFill[g], false, &ctx->rfAllocations) == PMC_OK;
            for (int k = 0; k < 2 && room; ++k)
                room = ctx->allocate<uint32_t>(want, &ctx->statKeys[g][k], false, &ctx->rfAllocations) == PMC_OK
                       && ctx->allocate<double>(want, &ctx->statVals[g][k], false, &ctx->rfAllocations) == PMC_OK;
            if (room && !ctx->statTemp[g])
            {
                uint8_t* t = nullptr;
                room = ctx->allocate<uint8_t>(pmcRfTempBytes(pmcRfMaxParts()), &t, false, &ctx->rfAllocations) == PMC_OK;
                ctx->statTemp[g] = t;
            }
            if (!room)
            {
                // (no room for the log: this group's sums are added atomically)
                for (int k = 0; k < 2; ++k)
                {
                    release(ctx->statKeys[g][k]), release(ctx->statVals[g][k]);
                    ctx->statKeys[g][k] = nullptr, ctx->statVals[g][k] = nullptr;
                }
                continue;
            }
            ctx->statCap[g] = want;
        }
    auto statLogOf = [&](int g) -> StatLogArgs {
        StatLogArgs a = {nullptr, nullptr, 0ull, 0, nullptr, nullptr, nullptr};
        if (statLogged && ctx->statCap[g])
            a = {ctx->statKeys[g][0], ctx->statVals[g][0], ctx->statCap[g], PMC_CTR_STATLOG(g), ctx->statWaveBase[g], ctx->statWaveFill[g], ctx->statChunkFill[g]};
        return a;
    };
    // an empty log: no wave holds a chunk, every chunk counts as full until a wave leaves it open or short
    auto statLogReset = [&](int g, hipStream_t stream) -> int {
        if (!statLogged || !ctx->statCap[g]) return PMC_OK;
        HIP_TRY(hipMemsetAsync(ctr + PMC_CTR_STATLOG(g), 0, sizeof(unsigned long long), stream));
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->statWaveFill[g]), (int)PMC_STAT_NO_CHUNK, PMC_STAT_LOG_WAVES, stream));
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->statChunkFill[g]), PMC_RF_LOG_CHUNK, size_t(ctx->statCap[g] / PMC_RF_LOG_CHUNK), stream));
        return PMC_OK;
    };
    // the log of group g (`claimed` entries) -> accumulator records, on `stream`; the cursor starts again at zero
    auto statFlush = [&](int g, unsigned long long claimed, hipStream_t stream) -> int {
        if (!statLogged || !ctx->statCap[g]) return PMC_OK;
        const unsigned long long n = std::min(claimed, ctx->statCap[g]) / PMC_RF_LOG_CHUNK * PMC_RF_LOG_CHUNK;
        if (n)
            HIP_TRY(pmcLaunchStatFlush(ctx->slot, ctx->statKeys[g][0], ctx->statVals[g][0], ctx->statKeys[g][1], ctx->statVals[g][1], n, int(statParts), ctx->statTemp[g],
                                       ctx->numCU, ctx->statChunkFill[g], stream));
        return statLogReset(g, stream);
    };
    // ---- statistics: every slot group starts with its share of the pool of list blocks, all of them free
    if (D.any_stats && ctx->statPoolBlocks)
    {
        const int64_t per = ctx->statPoolBlocks / G;
        unsigned long long freeCount[PMC_MAX_GROUPS] = {0, 0, 0, 0};
        bool changed = false;
        for (int g = 0; g < PMC_MAX_GROUPS; ++g)
        {
            const int32_t firstBlock = g < G ? int32_t(g * per) : 0, count = g < G ? int32_t(per) : 0;
            changed = changed || D.stat_pool_first[g] != firstBlock || D.stat_pool_count[g] != count;
            D.stat_pool_first[g] = firstBlock;
            D.stat_pool_count[g] = count;
            freeCount[g] = (unsigned long long)count;
        }
        if (changed)
        {
            HIP_TRY(hipStreamSynchronize(st));
            HIP_TRY(pmcUploadScene(ctx->slot, &D, st));
        }
        HIP_TRY(hipMemcpyAsync(D.stat_pool_free, ctx->statPoolIota, size_t(ctx->statPoolBlocks) * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(ctr + PMC_CTR_STATFREE(0), freeCount, sizeof(freeCount), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));  // (freeCount lives on this frame)
    }
    for (int g = 0; g < G; ++g)
        if (int rc = statLogReset(g, st)) return rc;
    HIP_TRY(hipMemsetAsync(ctr + PMC_CTR_HISTORY, 0, sizeof(unsigned long long), st));
    HIP_TRY(hipMemsetAsync(ctr + 32, 0, 4 * PMC_MAX_GROUPS * sizeof(unsigned long long), st));
    HIP_TRY(hipMemsetAsync(ctr + PMC_CTR_TASK(0, 0), 0, PMC_CTR_TASKS_PER_GROUP * PMC_MAX_GROUPS * sizeof(unsigned long long), st));
    HIP_TRY(hipEventRecord(ctx->evStart, st));
    int launchBlocks = ctx->numCU * 4;  // persistent launch workgroups, as the transition kernel's
    if (const char* env = pmcTune("PMC_LAUNCH_BLOCKS_PER_CU")) launchBlocks = ctx->numCU * std::max(1, atoi(env));
    int cycleBlocks = ctx->numCU * 4;  // persistent cycle start workgroups (grid tables staged once per workgroup)
    if (const char* env = pmcTune("PMC_CYCLE_BLOCKS_PER_CU")) cycleBlocks = ctx->numCU * std::max(1, atoi(env));
    int transitionBlocks = ctx->numCU * 4;  // persistent transition workgroups (tables staged once per workgroup)
    if (const char* env = pmcTune("PMC_TRANSITION_BLOCKS_PER_CU")) transitionBlocks = ctx->numCU * std::max(1, atoi(env));
    auto enqueue = [&](int g, bool initial) -> int {
        hipStream_t sg = ctx->groupStream[g];
        // the list of live slots the previous generation left (as many as its live count, which came back with the stream)
        int* const listIn = (!initial && listBuilt[g]) ? D.tasks.liveList + int64_t(listHalf[g]) * D.slots.num_slots + base[g] : nullptr;
        const int listLen = listIn ? int(ctx->pinned[g]) : 0;
        if (!initial)
        {
            // (the radiation-field log of the group's previous generation: its size came back with the live count)
            if (int rc = rfFlush(g, ctx->pinned[PMC_MAX_GROUPS + g])) return rc;
            ctx->pinned[PMC_MAX_GROUPS + g] = 0;
            // (the statistics log of the group, once half full: its fill came back with the live count)
            if (ctx->pinned[2 * PMC_MAX_GROUPS + g] > ctx->statCap[g] / 2)
            {
                if (int rc = statFlush(g, ctx->pinned[2 * PMC_MAX_GROUPS + g], sg)) return rc;
                ctx->pinned[2 * PMC_MAX_GROUPS + g] = 0;
            }
            HIP_TRY(hipMemsetAsync(ctr + PMC_CTR_TASK(g, 0), 0, PMC_CTR_TASKS_PER_GROUP * sizeof(unsigned long long), sg));  // task cursors
            if (ctx->xcdCursors) HIP_TRY(hipMemsetAsync(cursorSet(g, 0), 0, (PMC_SORT_OBS + 1) * 8 * sizeof(unsigned long long), sg));
            HIP_TRY(hipEventRecord(ctx->evA[g], sg));
            if (D.grid_kind == PMC_GRID_OCTREE)
            {
                // the walks of the generation: one peel-off kernel per observer on the group's side stream, next to the
                // propagation kernel on the group's stream (they touch different task records and result fields)
                hipStream_t sp = ctx->peelStream[g];
                if (serialWalks) sp = sg;
                const int* list = listIn;
                const int numTasks = list ? listLen : size[g];
                const int peelLanes = pmcPeelBlock() * listTasksPerLane, propLanes = pmcPropBlock() * listTasksPerLane;
                const int peelGrid = list ? std::max(1, std::min(ctx->peelGrid, (numTasks + peelLanes - 1) / peelLanes)) : ctx->peelGrid;
                const int propGrid = list ? std::max(1, std::min(ctx->grid, (numTasks + propLanes - 1) / propLanes)) : ctx->grid;
                HIP_TRY(hipStreamWaitEvent(sp, ctx->evA[g], 0));
                for (int i = 0; i < D.num_instruments; ++i)
                    if (!D.inst[i].same_observer)
                    {
                        int k = -1;
                        for (int q = 0; q < numSortObs; ++q)
                            if (sortObs[q] == i) k = q;
                        const bool sorted = peelSorted[g] && !list && k >= 0;
                        HIP_TRY(pmcLaunchPeel(ctx->slot, (ctx->wide ? 1 : 0) | (D.num_media > 1 ? 2 : 0), base[g], numTasks, sorted ? nullptr : list, PMC_CTR_TASK(g, 1 + i), i,
                                              (int)D.inst[i].sgn, peelGrid, ctx->walkLds, sorted ? ctx->peelRec[g][k] : nullptr, sorted ? pmcPeelSortedCount(ctx->peelTemp[g][k]) : nullptr,
                                              sorted && xcdAffinity ? cursorSet(g, k) : nullptr, sp));
                    }
                HIP_TRY(hipEventRecord(ctx->evJoin[g], sp));
                RfLogArgs log = {ctx->rfKeys[g][0], ctx->rfVals[g][0], rfLogged ? ctx->rfCap[g] : 0ull, PMC_CTR_RFLOG(g), rfPadKey};
                if (serialWalks) HIP_TRY(hipEventRecord(ctx->evProp[g], sg));  // (in series: the propagation kernel starts where the peel-off kernels end)
                HIP_TRY(pmcLaunchProp(ctx->slot, ctx->wide, (D.rf_store ? 1 : 0) | (D.explicit_absorption ? 2 : 0) | (D.num_media > 1 ? 4 : 0), base[g], numTasks, list, PMC_CTR_TASK(g, 0), seed, propGrid, ctx->walkLds, &log, sg));
                if (!serialWalks) HIP_TRY(hipEventRecord(ctx->evProp[g], sg));
                HIP_TRY(hipStreamWaitEvent(sg, ctx->evJoin[g], 0));
            }
            else
            {
                // (sorted observers: one stream of single walks -- the propagation walks in slot order, then every observer's peel-off walks in the
                // order of the detector tile they start behind)
                WalkStreamArgs tasks;
                std::memset(&tasks, 0, sizeof(tasks));
                bool streamEmpty = false;
                if (peelSorted[g])
                {
                    tasks.numLists = numSortObs;
                    for (int k = 0; k < numSortObs; ++k)
                        tasks.rec[k] = 1 + sortObs[k], tasks.list[k] = ctx->peelList[g][k], tasks.count[k] = pmcPeelSortedCount(ctx->peelTemp[g][k]);
                    tasks.xcdCursor = xcdAffinity ? cursorSet(g, 0) : nullptr;
                    if (propSortIndex >= 0)
                        tasks.propList = ctx->peelList[g][propSortIndex], tasks.propCount = pmcPeelSortedCount(ctx->peelTemp[g][propSortIndex]);
                    // (lists that the Voronoi peel-off kernel takes, below: empty for the stream)
                    bool left = false;  // does the stream keep a list?
                    for (int k = 0; k < numSortObs; ++k)
                        if (voroPeelKernels && D.vobs_of_inst[sortObs[k]] >= 0)
                            tasks.count[k] = zeroCount;
                        else
                            left = true;
                    // (the list is in cone order: ONE cursor, all XCDs on the same cone table at a time -- 490 against 493 ms of walk kernels per 2e7 packets
                    // with an eighth of the list per XCD, profiles/sweeps/r05_i15)
                    const bool ownProp = voroPropKernel && propSortIndex >= 0;
                    if (ownProp)
                    {
                        int propBlocks = pmcVoroPropWavesPerSimd();
                        if (const char* v = pmcTune("PMC_VPROP_BLOCKS_PER_CU")) propBlocks = std::max(1, atoi(v));
                        HIP_TRY(pmcLaunchVoroProp(ctx->slot, tasks.propList, tasks.propCount, cursorSet(g, PMC_SORT_OBS), (xcdAffinity && pmcTune("PMC_VPROP_XCD_SEGMENTS")) ? 8 : 1, seed,
                                                  ctx->numCU * propBlocks, sg));
                        tasks.propCount = zeroCount;
                    }
                    else
                        left = true;
                    streamEmpty = !left;
                }
                if (!streamEmpty)
                HIP_TRY(pmcLaunchWalk(ctx->slot, D.grid_kind, (D.rf_store ? 1 : 0) | (D.explicit_absorption ? 2 : 0) | (D.num_media > 1 ? 4 : 0), base[g], size[g], PMC_CTR_TASK(g, 0), seed, ctx->grid, ctx->block,
                                      ctx->walkLds, peelSorted[g] ? &tasks : nullptr, sg));
                if (peelSorted[g])
                {
                    // the peel-off kernels on the group's side stream next to the propagation kernel (as on the octree: one is bound by the lines it
                    // gets from beyond L2, the others by instructions and the L1's access rate); `PMC_VORO_WALKS_IN_SERIES`: behind it, one stream
                    bool anyPeel = false;
                    for (int k = 0; k < numSortObs; ++k) anyPeel = anyPeel || (voroPeelKernels && D.vobs_of_inst[sortObs[k]] >= 0);
                    const bool side = anyPeel && voroPropKernel && propSortIndex >= 0 && !serialWalks && pmcTune("PMC_VORO_WALKS_IN_SERIES") == nullptr;
                    hipStream_t sp = side ? ctx->peelStream[g] : sg;
                    int peelBlocks = pmcVoroPeelWavesPerSimd();
                    if (const char* v = pmcTune("PMC_VPEEL_BLOCKS_PER_CU")) peelBlocks = std::max(1, atoi(v));
                    if (side) HIP_TRY(hipStreamWaitEvent(sp, ctx->evA[g], 0));
                    for (int k = 0; k < numSortObs; ++k)
                        if (voroPeelKernels && D.vobs_of_inst[sortObs[k]] >= 0)
                            HIP_TRY(pmcLaunchVoroPeel(ctx->slot, 1 + sortObs[k], D.vobs_of_inst[sortObs[k]], ctx->peelList[g][k], pmcPeelSortedCount(ctx->peelTemp[g][k]),
                                                      cursorSet(g, 1 + k), xcdAffinity ? 8 : 1, ctx->numCU * peelBlocks, sp));
                    if (side)
                    {
                        HIP_TRY(hipEventRecord(ctx->evJoin[g], sp));
                        HIP_TRY(hipStreamWaitEvent(sg, ctx->evJoin[g], 0));
                    }
                }
            }
            haveWalk[g] = true;
            HIP_TRY(hipEventRecord(ctx->evB[g], sg));
            HIP_TRY(hipMemsetAsync(ctr + PMC_CTR_LIVE(g), 0, sizeof(unsigned long long), sg));
            const StatLogArgs statLog = statLogOf(g);
            HIP_TRY(pmcLaunchTransition(ctx->slot, base[g], size[g], g, seed, listIn, listLen, transitionBlocks, ctx->transitionLds, &statLog, sg));
            if (!listIn) HIP_TRY(pmcLaunchLaunch(ctx->slot, base[g], size[g], g, first, count, seed, 0, launchBlocks, ctx->launchLds, &statLog, sg));
        }
        else
        {
            if (g > 0) HIP_TRY(hipStreamWaitEvent(sg, ctx->evStart, 0));
            HIP_TRY(hipEventRecord(ctx->evB[g], sg));
            HIP_TRY(pmcLaunchLaunch(ctx->slot, base[g], size[g], g, first, count, seed, 1, (size[g] + 255) / 256, ctx->launchLds, nullptr, sg));
        }
        // every live slot of the group is at the start of a cycle now: the start states of its walks -- and, once the live slots
        // of the previous generation were fewer than half of the group's, their list for the next generation.  (The launch kernel
        // fills every slot whose history has ended as long as SourceSystem has an index left: fewer live slots than slots means
        // that nothing is left to launch, and the live slots can only become fewer.)
        const bool buildList = sparseLists && !initial && (listIn || ctx->pinned[g] < (unsigned long long)(size[g] / 2));
        if (listIn) listHalf[g] ^= 1;
        int* const listOut = D.tasks.liveList + int64_t(listHalf[g]) * D.slots.num_slots + base[g];
        const bool sortNow = numSortObs > 0 && !buildList && !listIn;
        const double gdx = D.gx1 - D.gx0, gdy = D.gy1 - D.gy0, gdz = D.gz1 - D.gz0;
        PeelSortArgs sortArgs;
        std::memset(&sortArgs, 0, sizeof(sortArgs));
        sortArgs.numObs = numSortObs;
        sortArgs.propIndex = propSortIndex;
        sortArgs.cap = (uint32_t)ctx->peelCap[g];
        for (int i = 0; i < 16; ++i) sortArgs.sortIndex[i] = -1;
        for (int k = 0; k < numSortObs; ++k) sortArgs.obs[k] = sortObs[k], sortArgs.sortIndex[sortObs[k]] = (int8_t)k;
        sortArgs.centre[0] = 0.5 * (D.gx0 + D.gx1), sortArgs.centre[1] = 0.5 * (D.gy0 + D.gy1), sortArgs.centre[2] = 0.5 * (D.gz0 + D.gz1);
        sortArgs.scale = PMC_PEEL_TILES / std::sqrt(gdx * gdx + gdy * gdy + gdz * gdz);
        int sortGroups = 0;
        // (sorted peel-off records: the sort's count pass over the slots as the transition / launch kernels left them; the cycle start kernel,
        // with the same workgroups, is its scatter pass)
        if (sortNow)
            HIP_TRY(pmcLaunchPeelSortCounts(ctx->slot, base[g], size[g], &sortArgs, octree ? ctx->peelRec[g] : nullptr, octree ? nullptr : ctx->peelList[g], ctx->peelTemp[g], &sortGroups, sg));
        HIP_TRY(pmcLaunchCycleStart(ctx->slot, D.grid_kind, base[g], size[g], buildList ? PMC_CTR_LIST(g) : -1, listOut, listIn, listLen, sortNow ? sortGroups : cycleBlocks,
                                    ctx->walkLds, sortNow ? &sortArgs : nullptr, sg));
        peelSorted[g] = sortNow;
        listBuilt[g] = buildList;
        HIP_TRY(hipEventRecord(ctx->evC[g], sg));
        HIP_TRY(hipMemcpyAsync(ctx->pinned + g, ctr + PMC_CTR_LIVE(g), sizeof(unsigned long long), hipMemcpyDeviceToHost, sg));
        if (rfLogged && !initial)
            HIP_TRY(hipMemcpyAsync(ctx->pinned + PMC_MAX_GROUPS + g, ctr + PMC_CTR_RFLOG(g), sizeof(unsigned long long), hipMemcpyDeviceToHost, sg));
        if (ctx->progress)
            HIP_TRY(hipMemcpyAsync(ctx->pinned + 3 * PMC_MAX_GROUPS + g, ctr + PMC_CTR_HISTORY, sizeof(unsigned long long), hipMemcpyDeviceToHost, sg));
        if (statLogged && !initial && ctx->statCap[g])
            HIP_TRY(hipMemcpyAsync(ctx->pinned + 2 * PMC_MAX_GROUPS + g, ctr + PMC_CTR_STATLOG(g), sizeof(unsigned long long), hipMemcpyDeviceToHost, sg));
        return PMC_OK;
    };
    // on any failure: no kernel of this segment may still be running (or be timed) when the call returns
    auto abandon = [&](int code) {
        hipDeviceSynchronize();
        // (the statistics of the abandoned segment must not reach the frames with the next one)
        if (D.stat_acc_records) hipMemset(D.stat_acc, 0, size_t(D.stat_acc_records) * 8 * sizeof(double));
        ctx->timed = false;
        return code;
    };
    auto lastReport = std::chrono::steady_clock::now();
    uint64_t reported = 0;
    auto drive = [&]() -> int {
        for (int g = 0; g < G; ++g)
            if (active[g])
            {
                int rc = enqueue(g, true);
                if (rc) return rc;
            }
        int remaining = 0;
        for (int g = 0; g < G; ++g) remaining += active[g] ? 1 : 0;
        for (int g = 0; remaining > 0; g = (g + 1) % G)
        {
            if (!active[g]) continue;
            HIP_TRY(hipStreamSynchronize(ctx->groupStream[g]));
            float ms = 0, walkOfGen = 0;
            if (haveWalk[g])
            {
                HIP_TRY(hipEventElapsedTime(&ms, ctx->evA[g], ctx->evB[g]));
                walkMs += ms;
                walkOfGen = ms;
                if (D.grid_kind == PMC_GRID_OCTREE)
                {
                    // the two kernel kinds of the generation: side by side on two streams (each span starts at evA), or in series
                    HIP_TRY(hipEventElapsedTime(&ms, ctx->evA[g], ctx->evJoin[g]));
                    peelMs += ms;
                    if (serialWalks)
                        HIP_TRY(hipEventElapsedTime(&ms, ctx->evProp[g], ctx->evB[g]));
                    else
                        HIP_TRY(hipEventElapsedTime(&ms, ctx->evA[g], ctx->evProp[g]));
                    propMs += ms;
                }
            }
            HIP_TRY(hipEventElapsedTime(&ms, ctx->evB[g], ctx->evC[g]));
            transMs += ms;
            if (ctx->pinned[g] == 0)
            {
                // (the group's last log)
                if (int rc = rfFlush(g, ctx->pinned[PMC_MAX_GROUPS + g])) return rc;
                ctx->pinned[PMC_MAX_GROUPS + g] = 0;
                active[g] = false;
                --remaining;
                continue;
            }
            ++generations;
            if (ctx->progress)
            {
                // (the history cursor came back with the group's live count; it runs past `count` when the last indices are handed out)
                const auto now = std::chrono::steady_clock::now();
                if (std::chrono::duration<double>(now - lastReport).count() >= ctx->progressInterval)
                {
                    lastReport = now;
                    // (every group copies the cursor into a word of its own, on its own stream; this group's copy is complete -- its
                    // stream has just been waited for -- and the report never goes backwards: a running maximum)
                    reported = std::max<uint64_t>(reported, std::min<uint64_t>(ctx->pinned[3 * PMC_MAX_GROUPS + g], count));
                    ctx->progress(ctx->progressUser, reported, count);
                }
            }
            if (genDump)
                fprintf(stderr, "PMC_GEN %d group %d live %llu walk_ms %.3f transition_ms %.3f\n", generations, g, ctx->pinned[g],
                        haveWalk[g] ? walkOfGen : 0.f, ms);
            int rc = enqueue(g, false);
            if (rc) return rc;
        }
        return PMC_OK;
    };
    for (int g = 0; g < PMC_MAX_GROUPS; ++g) ctx->pinned[PMC_MAX_GROUPS + g] = 0, ctx->pinned[2 * PMC_MAX_GROUPS + g] = 0;
    if (int rc = drive()) return abandon(rc);
    // the end of the segment (a failure here leaves the segment abandoned like one in the generations)
    auto finish = [&]() -> int {
        // what is left in the groups' statistics logs: the fills that came back with the groups' last generations are final (drive() has
        // waited for every group); the groups' flushes run side by side on their streams
        if (statLogged)
            for (int g = 0; g < G; ++g)
                if (int rc = statFlush(g, ctx->pinned[2 * PMC_MAX_GROUPS + g], ctx->groupStream[g])) return rc;
        // (the last radiation-field logs of the groups are reduced on their streams too)
        if (rfLogged || statLogged)
            for (int g = 0; g < G; ++g) HIP_TRY(hipStreamSynchronize(ctx->groupStream[g]));
        // the segment's statistics: accumulator records -> wifu arrays
        if (D.stat_acc_records) HIP_TRY(pmcLaunchStatMerge(ctx->slot, ctx->numCU * 8, st));
        HIP_TRY(hipEventRecord(ctx->evStop, st));
        HIP_TRY(hipEventSynchronize(ctx->evStop));
        HIP_TRY(hipEventElapsedTime(&ctx->totalMs, ctx->evStart, ctx->evStop));
        return PMC_OK;
    };
    if (int rc = finish()) return abandon(rc);
    ctx->walkMs = walkMs;
    ctx->transitionMs = transMs;
    ctx->peelMs = peelMs;
    ctx->propMs = propMs;
    if (serialWalks && pmcTune("PMC_TIMING_DUMP"))
        fprintf(stderr, "PMC_TIMING peel %.2f ms prop %.2f ms transition+launch %.2f ms segment %.2f ms\n", peelMs, propMs, transMs, ctx->totalMs);
    ctx->generations = generations;
    ctx->timed = true;
    // internal errors counted by the kernels (a sorted peel-off record without a place: see peelTile, pmc_transition.inc)
    {
        unsigned long long tail[3] = {0, 0, 0};  // counters 5 .. 7
        HIP_TRY(hipMemcpy(tail, ctr + 5, sizeof(tail), hipMemcpyDeviceToHost));
        if (tail[2] > ctx->internalErrorsSeen)
        {
            const unsigned long long fresh = tail[2] - ctx->internalErrorsSeen;
            ctx->internalErrorsSeen = tail[2];
            return fail(PMC_ERR_DEVICE, std::to_string(fresh) + " peel-off walks found no place in the sorted records (the two passes of the sort disagree): the segment's results are incomplete");
        }
    }
    // a history with more distinct pixels than the statistics list holds: the statistics arrays are wrong -- say so
    if (D.any_stats)
    {
        unsigned long long overflows = 0;
        HIP_TRY(hipMemcpy(&overflows, ctr + 5, sizeof(overflows), hipMemcpyDeviceToHost));
        if (overflows > ctx->overflowsSeen)
        {
            const unsigned long long fresh = overflows - ctx->overflowsSeen;
            ctx->overflowsSeen = overflows;
            return fail(PMC_ERR_OVERFLOW, std::to_string(fresh) + " photon histories lost contributions to the statistics arrays: the pool of "
                                              + std::to_string(ctx->statPoolBlocks) + " list blocks (" + std::to_string(PMC_STAT_CAP)
                                              + " distinct pixels each) ran out; the statistics arrays of this segment are incomplete.  Raise "
                                                "PMC_STAT_POOL_BLOCKS, or lower PMC_NUM_SLOTS (fewer histories in flight)");
        }
    }
    return PMC_OK;
}

int pmc_set_progress(pmc_ctx* ctx, pmc_progress_fn report, void* user, double interval_seconds)
{
    if (!ctx) return fail(PMC_ERR_INVALID, "null context");
    ctx->progress = report;
    ctx->progressUser = user;
    ctx->progressInterval = interval_seconds > 0. ? interval_seconds : 0.;
    return PMC_OK;
}

int pmc_sync(pmc_ctx* ctx)
{
    if (!ctx) return fail(PMC_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PMC_OK;
}

int pmc_last_kernel_ms(pmc_ctx* ctx, float* ms)
{
    if (!ctx || !ms) return fail(PMC_ERR_INVALID, "null argument");
    if (!ctx->timed) return fail(PMC_ERR_INVALID, "no segment has been run yet");
    *ms = ctx->walkMs;
    return PMC_OK;
}

int pmc_last_timing(pmc_ctx* ctx, float* total_ms, float* walk_ms, float* transition_ms, int32_t* generations)
{
    if (!ctx) return fail(PMC_ERR_INVALID, "null argument");
    if (!ctx->timed) return fail(PMC_ERR_INVALID, "no segment has been run yet");
    if (total_ms) *total_ms = ctx->totalMs;
    if (walk_ms) *walk_ms = ctx->walkMs;
    if (transition_ms) *transition_ms = ctx->transitionMs;
    if (generations) *generations = ctx->generations;
    return PMC_OK;
}

int pmc_last_walk_timing(pmc_ctx* ctx, float* peel_ms, float* prop_ms)
{
    if (!ctx) return fail(PMC_ERR_INVALID, "null argument");
    if (!ctx->timed) return fail(PMC_ERR_INVALID, "no segment has been run yet");
    if (peel_ms) *peel_ms = ctx->peelMs;
    if (prop_ms) *prop_ms = ctx->propMs;
    return PMC_OK;
}

int pmc_walk_work(pmc_ctx* ctx, pmc_walk_work_values* out)
{
    if (!ctx || !out) return fail(PMC_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    unsigned long long host[6];
    HIP_TRY(hipMemcpy(host, ctx->dev.counters + PMC_CTR_WALKWORK, sizeof(host), hipMemcpyDeviceToHost));
    out->peel_wave_steps = host[0], out->peel_lane_steps = host[1], out->peel_rounds = host[2];
    out->prop_wave_steps = host[3], out->prop_lane_steps = host[4], out->prop_rounds = host[5];
    return PMC_OK;
}

int pmc_debug_tables(pmc_ctx* ctx, pmc_debug_table_values* out)
{
    if (!ctx || !out) return fail(PMC_ERR_INVALID, "null argument");
    if (ctx->dev.grid_kind != PMC_GRID_OCTREE) return fail(PMC_ERR_INVALID, "not an octree scene");
    out->cell_table = ctx->dev.cell_tab;
    out->cell_slots = ctx->dev.cell_slots;
    out->loose_base = ctx->dev.cell_slots;
    out->task_cell = ctx->dev.tasks.cell;
    out->num_slots = ctx->allocatedSlots;
    return PMC_OK;
}

int pmc_download(pmc_ctx* ctx, double* host_frames, int64_t num_doubles)
{
    if (!ctx || !host_frames) return fail(PMC_ERR_INVALID, "null argument");
    if (num_doubles != ctx->frameSize) return fail(PMC_ERR_INVALID, "frame buffer size mismatch");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(host_frames, ctx->frames, size_t(num_doubles) * sizeof(double), hipMemcpyDeviceToHost));
    return PMC_OK;
}

double* pmc_frames_device(pmc_ctx* ctx)
{
    return ctx ? ctx->frames : nullptr;
}

int64_t pmc_frames_size(pmc_ctx* ctx)
{
    return ctx ? ctx->frameSize : 0;
}

int64_t pmc_radiation_field_size(pmc_ctx* ctx)
{
    return ctx ? ctx->rfSize : 0;
}

double* pmc_radiation_field_device(pmc_ctx* ctx)
{
    return ctx ? ctx->dev.rf : nullptr;
}

int pmc_bind_radiation_field(pmc_ctx* ctx, double* device_ptr, int64_t num_doubles)
{
    if (!ctx || !device_ptr) return fail(PMC_ERR_INVALID, "null argument");
    if (!ctx->rfSize) return fail(PMC_ERR_INVALID, "the scene does not store the radiation field");
    if (num_doubles != ctx->rfSize) return fail(PMC_ERR_INVALID, "radiation field size mismatch");
    ctx->dev.rf = device_ptr;
    ctx->sceneDirty = true;
    return PMC_OK;
}

int pmc_download_radiation_field(pmc_ctx* ctx, double* host_rf, int64_t num_doubles)
{
    if (!ctx || !host_rf) return fail(PMC_ERR_INVALID, "null argument");
    if (!ctx->rfSize) return fail(PMC_ERR_INVALID, "the scene does not store the radiation field");
    if (num_doubles != ctx->rfSize) return fail(PMC_ERR_INVALID, "radiation field size mismatch");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(host_rf, ctx->dev.rf, size_t(num_doubles) * sizeof(double), hipMemcpyDeviceToHost));
    return PMC_OK;
}

int pmc_clear_radiation_field(pmc_ctx* ctx)
{
    if (!ctx) return fail(PMC_ERR_INVALID, "null context");
    if (!ctx->rfSize) return PMC_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemsetAsync(ctx->dev.rf, 0, size_t(ctx->rfSize) * sizeof(double), ctx->stream));
    return PMC_OK;
}

int pmc_counters(pmc_ctx* ctx, pmc_counter_values* out)
{
    if (!ctx || !out) return fail(PMC_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    unsigned long long host[PMC_NUM_COUNTERS];
    HIP_TRY(hipMemcpy(host, ctx->dev.counters, sizeof(host), hipMemcpyDeviceToHost));
    out->histories = host[0];
    out->paths = host[1];
    out->cell_visits = host[2];
    out->detector_updates = host[3];
    out->scatterings = host[4];
    out->stat_overflows = host[5];
    out->rewalk_visits = host[6];
    if (pmcTune("PMC_PROFILE_DUMP"))
    {
        // work counters of the octree walk kernels: lane utilisation = lane_steps / (64 wave_steps)
        const unsigned long long* w = host + PMC_CTR_WALKWORK;
        fprintf(stderr, "PMC_PROFILE peel: wave_steps %llu lane_steps %llu (%.1f %% of the lanes) service_rounds %llu\n", w[0], w[1],
                w[0] ? 100. * double(w[1]) / (64. * double(w[0])) : 0., w[2]);
        fprintf(stderr, "PMC_PROFILE prop: wave_steps %llu lane_steps %llu (%.1f %% of the lanes) service_rounds %llu\n", w[3], w[4],
                w[3] ? 100. * double(w[4]) / (64. * double(w[3])) : 0., w[5]);
    }
#ifdef PMC_PROFILE
    if (pmcTune("PMC_PROFILE_DUMP"))
    {
        for (int k = 0; k < 2; ++k)
        {
            const unsigned long long* t = host + 208 + 8 * k;
            fprintf(stderr, "PMC_PROFILE %s phases (wave cycles): gather %llu tau+position %llu descent %llu walls %llu inside+exit %llu "
                            "service-finish+claim %llu service-loads %llu loop %llu\n", k ? "prop" : "peel", t[0], t[1], t[2], t[3], t[4], t[5],
                    t[6], t[7]);
        }
        for (int k = 0; k < 2; ++k)
        {
            const unsigned long long* c = host + 224 + 8 * k;
            fprintf(stderr, "PMC_PROFILE %s census (lane events): literal-algorithm steps %llu, edge %llu, descents %llu over %llu levels, octet links %llu, "
                            "hit %llu, exit %llu\n", k ? "prop" : "peel", c[0], c[2], c[3], c[4], c[5], c[6], c[7]);
        }
        fprintf(stderr, "PMC_PROFILE transition (wave cycles): stage %llu mode-load %llu cycle-tail %llu append %llu loads+detect %llu scatter %llu start-cycle %llu flush %llu\n",
                host[192], host[193], host[194], host[195], host[196], host[197], host[198], host[199]);
        fprintf(stderr, "PMC_PROFILE launch (wave cycles): stage %llu list %llu stats-flush %llu start-cycle %llu draw+sample %llu - %llu flush %llu\n",
                host[200], host[201], host[202], host[203], host[204], host[205], host[206]);
    }
#endif
    return PMC_OK;
}

int pmc_reset_counters(pmc_ctx* ctx)
{
    if (!ctx) return fail(PMC_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemsetAsync(ctx->dev.counters, 0, PMC_NUM_COUNTERS * sizeof(unsigned long long), ctx->stream));
    ctx->overflowsSeen = 0;
    ctx->internalErrorsSeen = 0;
    return PMC_OK;
}

int pmc_trace_ray(pmc_ctx* ctx, const double r[3], const double k[3], int32_t* m, double* ds, int32_t cap, int32_t* n)
{
    if (!ctx || !r || !k || !m || !ds || !n || cap < 0) return fail(PMC_ERR_INVALID, "invalid argument");
    HIP_TRY(hipSetDevice(ctx->device));
    // octree: the ray is traced with BOTH flavours of the step (direction in scalar registers as in the peel-off kernel,
    // in vector registers as in the propagation kernel); they must agree bit for bit
    const int flavours = ctx->dev.grid_kind == PMC_GRID_OCTREE ? 2 : 1;
    const size_t room = size_t(std::max(cap, 1));
    int32_t* dm = nullptr;
    double* dds = nullptr;
    int32_t* dn = nullptr;
    double* dk = nullptr;
    HIP_TRY(hipMalloc(&dm, sizeof(int32_t) * room * flavours));
    HIP_TRY(hipMalloc(&dds, sizeof(double) * room * flavours));
    HIP_TRY(hipMalloc(&dn, sizeof(int32_t) * flavours));
    HIP_TRY(hipMalloc(&dk, sizeof(double) * 3));
    hipError_t e = hipMemcpy(dk, k, 3 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && ctx->sceneDirty)
    {
        e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = pmcUploadScene(ctx->slot, &ctx->dev, ctx->stream);
        if (e == hipSuccess) ctx->sceneDirty = false;
    }
    for (int f = 0; f < flavours && e == hipSuccess; ++f)
        e = pmcLaunchTrace(ctx->slot, ctx->dev.grid_kind, ctx->wide, f, r, k, dk, dm + f * room, dds + f * room, cap, dn + f, ctx->walkLds,
                           ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    int rc = PMC_OK;
    if (e != hipSuccess)
        rc = hipFail(e, "trace kernel");
    else
    {
        int32_t count[2] = {0, 0};
        hipMemcpy(count, dn, sizeof(int32_t) * flavours, hipMemcpyDeviceToHost);
        *n = count[0];
        const int32_t got = std::min(*n, cap);
        hipMemcpy(m, dm, sizeof(int32_t) * got, hipMemcpyDeviceToHost);
        hipMemcpy(ds, dds, sizeof(double) * got, hipMemcpyDeviceToHost);
        if (flavours == 2)
        {
            std::vector<int32_t> m2(got);
            std::vector<double> ds2(got);
            hipMemcpy(m2.data(), dm + room, sizeof(int32_t) * got, hipMemcpyDeviceToHost);
            hipMemcpy(ds2.data(), dds + room, sizeof(double) * got, hipMemcpyDeviceToHost);
            if (count[1] != count[0] || std::memcmp(m2.data(), m, sizeof(int32_t) * got) || std::memcmp(ds2.data(), ds, sizeof(double) * got))
                rc = fail(PMC_ERR_DEVICE, "octree traversal: the scalar-direction and vector-direction steps disagree on this ray");
        }
    }
    hipFree(dm);
    hipFree(dds);
    hipFree(dn);
    hipFree(dk);
    return rc;
}

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
