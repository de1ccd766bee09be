// pmc_run.hip -- pmc_run_primary: the pool of packet slots and the generation loop over the slot groups (MonteCarloSimulation::runPrimaryEmission,
// MonteCarloSimulation.cpp:104-138: parallel->call(Npp, performLifeCycle) up to instrumentSystem()->flush()).
#include "pmc_context.h"

namespace
{
    int allocateSlotArrays(pmc_ctx* ctx, int64_t n);
    int allocateSlots(pmc_ctx* ctx, int64_t n);

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
        // (the plan above was held against the memory that was free THEN: contexts of other processes that start on the same device at the same
        // time -- ranks of a job that share a device -- plan against the same free memory.  What is left now must still hold the segment's logs, sort
        // buffers and the kernels' own needs: if an allocation failed, or less than 4 GB is left, the default steps down and tries again)
        bool crowded = rc == PMC_ERR_DEVICE || rc == PMC_ERR_NOMEM;
        if (!rc && hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess && freeBytes < (size_t(4) << 30)) crowded = true;
        if (crowded && !ctx->slotsConfigured && n > (int64_t(1) << 20))
        {
            for (void* p : ctx->slotAllocations) hipFree(p);
            ctx->slotAllocations.clear();
            ctx->allocatedSlots = 0;
            const int64_t less = std::max<int64_t>(int64_t(1) << 20, n / 2);
            fprintf(stderr, "libpmc: device %d is short of memory next to other contexts (%.1f GB left): this segment runs with %lld packet slots instead of %lld\n",
                    ctx->device, freeBytes * 1e-9, (long long)less, (long long)n);
            ctx->steppedDownFree = freeBytes + 1;
            return allocateSlots(ctx, less);
        }
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

int pmcAllocateSlots(pmc_ctx* ctx, int64_t n)
{
    return allocateSlots(ctx, n);
}

extern "C" {

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
    // staggered end of a segment (endedScanKernel): slot group g stops taking histories when fewer than g * drainKeep are left
    uint64_t drainKeep = 0;
    if (const char* env = pmcTune("PMC_DRAIN_KEEP")) drainKeep = (uint64_t)std::max(0.0, atof(env));
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
    const bool voroPeelKernels = D.grid_kind == PMC_GRID_VORONOI && pmcTune("PMC_VORO_NO_PEEL_KERNEL") == nullptr;
    // ... and the propagation walks (every flavour since round 6), on the table of runs with all neighbours (when pmc_create built it)
    const bool voroPropKernel = D.grid_kind == PMC_GRID_VORONOI && D.vgen_run && pmcTune("PMC_VORO_NO_PROP_KERNEL") == nullptr
                                && !(pmcTune("PMC_VORO_PLAIN_PROP_ONLY") && (D.num_media > 1 || D.rf_store || D.explicit_absorption));
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
    // ---- statistics: the pool of list blocks GROWS when a slot group could run out of blocks in its next generation (round 6; rounds 1-5
    // failed the segment with PMC_ERR_OVERFLOW: the reference's list is a std::vector, FluxRecorder.hpp:327-338).  A history takes at most one
    // block per instrument and generation, so a group whose free blocks number at least its live slots x instruments with statistics cannot
    // run out; when they do not, everything in flight is waited for, the pool arrays are allocated anew with room for `add` more blocks (the
    // old contents copied, as a std::vector grows), the new blocks go to the free stack of the group that asked, and the scene constants are
    // uploaded again.  No device memory for it: the segment goes on with the pool it has (and fails loudly if that does run out).
    int statInstruments = 0;
    for (int i = 0; i < D.num_instruments; ++i) statInstruments += D.inst[i].record_stats ? 1 : 0;
    const bool poolGrows = D.any_stats && ctx->statPoolBlocks > 0 && pmcTune("PMC_STAT_POOL_NO_GROWTH") == nullptr;
    bool poolCannotGrow = false;
    auto growStatPool = [&](int g, int64_t need) -> int {
        HIP_TRY(hipDeviceSynchronize());
        unsigned long long freeNow[PMC_MAX_GROUPS] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpy(freeNow, ctr + PMC_CTR_STATFREE(0), sizeof(freeNow), hipMemcpyDeviceToHost));
        const int64_t old = ctx->statPoolBlocks;
        const int64_t add = std::min<int64_t>(std::max<int64_t>(2 * need, old), (int64_t(1) << 30) - old);
        if (add <= 0) return PMC_OK;
        size_t freeBytes = 0, totalBytes = 0;
        const size_t bytes = size_t(old + add) * (PMC_STAT_CAP * 12 + 12);
        if (hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess && bytes + (size_t(1) << 30) > freeBytes)
        {
            static std::atomic<bool> said{false};
            if (!said.exchange(true))
                fprintf(stderr, "libpmc: the statistics lists of the histories in flight need more blocks than the pool of %lld has, and device %d has no room for "
                                "%.1f GB more: the segment goes on and fails if the pool does run out (PMC_NUM_SLOTS lowers the number of histories in flight)\n",
                        (long long)old, ctx->device, bytes * 1e-9);
            return PMC_OK;
        }
        int32_t *bin = nullptr, *next = nullptr, *stack = nullptr, *iota = nullptr;
        double* w = nullptr;
        auto& own = ctx->slotAllocations;
        int rc;
        if ((rc = ctx->allocate<int32_t>(size_t(old + add) * PMC_STAT_CAP, &bin, false, &own))) return rc;
        if ((rc = ctx->allocate<double>(size_t(old + add) * PMC_STAT_CAP, &w, false, &own))) return rc;
        if ((rc = ctx->allocate<int32_t>(size_t(old + add), &next, false, &own))) return rc;
        if ((rc = ctx->allocate<int32_t>(size_t(old + add), &stack, false, &own))) return rc;
        if ((rc = ctx->allocate<int32_t>(size_t(old + add), &iota, false, &own))) return rc;
        HIP_TRY(hipMemcpy(bin, D.stat_pool_bin, size_t(old) * PMC_STAT_CAP * sizeof(int32_t), hipMemcpyDeviceToDevice));
        HIP_TRY(hipMemcpy(w, D.stat_pool_w, size_t(old) * PMC_STAT_CAP * sizeof(double), hipMemcpyDeviceToDevice));
        HIP_TRY(hipMemcpy(next, D.stat_pool_next, size_t(old) * sizeof(int32_t), hipMemcpyDeviceToDevice));
        std::vector<int32_t> ids(static_cast<size_t>(old + add));
        for (size_t i = 0; i < ids.size(); ++i) ids[i] = (int32_t)i;
        HIP_TRY(hipMemcpy(iota, ids.data(), ids.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        // the free stacks: every group keeps its stack (its first freeNow entries), the new ids go on top of group g's
        int32_t first = 0;
        for (int h = 0; h < PMC_MAX_GROUPS; ++h)
        {
            const int32_t count = D.stat_pool_count[h] + (h == g ? int32_t(add) : 0);
            if (D.stat_pool_count[h] > 0 && freeNow[h] > 0)
                HIP_TRY(hipMemcpy(stack + first, D.stat_pool_free + D.stat_pool_first[h], size_t(freeNow[h]) * sizeof(int32_t), hipMemcpyDeviceToDevice));
            if (h == g) HIP_TRY(hipMemcpy(stack + first + freeNow[h], ids.data() + old, size_t(add) * sizeof(int32_t), hipMemcpyHostToDevice));
            D.stat_pool_first[h] = first;
            D.stat_pool_count[h] = count;
            first += count;
        }
        freeNow[g] += (unsigned long long)add;
        HIP_TRY(hipMemcpy(ctr + PMC_CTR_STATFREE(0), freeNow, sizeof(freeNow), hipMemcpyHostToDevice));
        // the old arrays
        for (void* gone : {(void*)D.stat_pool_bin, (void*)D.stat_pool_w, (void*)D.stat_pool_next, (void*)D.stat_pool_free, (void*)ctx->statPoolIota})
        {
            own.erase(std::remove(own.begin(), own.end(), gone), own.end());
            hipFree(gone);
        }
        D.stat_pool_bin = bin, D.stat_pool_w = w, D.stat_pool_next = next, D.stat_pool_free = stack;
        ctx->statPoolIota = iota;
        ctx->statPoolBlocks = old + add;
        ctx->pinned[4 * PMC_MAX_GROUPS + g] = freeNow[g];
        ctx->statPoolGrowths += 1;
        HIP_TRY(pmcUploadScene(ctx->slot, &D, st));
        HIP_TRY(hipStreamSynchronize(st));
        return PMC_OK;
    };
    auto enqueue = [&](int g, bool initial) -> int {
        hipStream_t sg = ctx->groupStream[g];
        // the list of live slots the previous generation left (as many as its live count, which came back with the stream)
        int* const listIn = (!initial && listBuilt[g]) ? D.tasks.liveList + int64_t(listHalf[g]) * D.slots.num_slots + base[g] : nullptr;
        const int listLen = listIn ? int(ctx->pinned[g]) : 0;
        if (!initial && poolGrows)
        {
            // (the group's free blocks came back with its live count: enough for one block per live slot and instrument with statistics?)
            // (only once blocks have been taken at all -- histories of more than 48 distinct pixels exist in this scene --, or when the ski file asks for
            // many scattering events per history: a scene whose lists stay short never touches the pool, however small it is; and not again in a
            // segment in which the device had no room for more)
            const int64_t need = int64_t(ctx->pinned[g]) * statInstruments, freeBlocks = int64_t(ctx->pinned[4 * PMC_MAX_GROUPS + g]);
            const bool inUse = freeBlocks < int64_t(D.stat_pool_count[g]) || D.min_scatt_events > 16;
            if (inUse && freeBlocks < need && !poolCannotGrow)
            {
                const int64_t before = ctx->statPoolBlocks;
                if (int rc = growStatPool(g, need)) return rc;
                poolCannotGrow = ctx->statPoolBlocks == before;
            }
        }
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
                                                  (D.rf_store ? 1 : 0) | (D.explicit_absorption ? 2 : 0) | (D.num_media > 1 ? 4 : 0), ctx->numCU * propBlocks, sg));
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
                                                      cursorSet(g, 1 + k), xcdAffinity ? 8 : 1, D.num_media > 1 ? 1 : 0, ctx->numCU * peelBlocks, sp));
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
            HIP_TRY(pmcLaunchTransition(ctx->slot, base[g], size[g], g, seed, listIn, listLen, transitionBlocks, ctx->transitionLds, &statLog, count,
                                        drainKeep * uint64_t(g), sg));
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
        if (poolGrows)
            HIP_TRY(hipMemcpyAsync(ctx->pinned + 4 * PMC_MAX_GROUPS + g, ctr + PMC_CTR_STATFREE(g), sizeof(unsigned long long), hipMemcpyDeviceToHost, sg));
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
    const auto segmentStart = lastReport;
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
                fprintf(stderr, "PMC_GEN %d group %d live %llu walk_ms %.3f transition_ms %.3f at_ms %.3f\n", generations, g, ctx->pinned[g],
                        haveWalk[g] ? walkOfGen : 0.f, ms, 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - segmentStart).count());
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

}  // extern "C"
