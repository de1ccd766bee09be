// pmc_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) of the primary-emission photon loop.
//
// The life cycle of a photon history (MonteCarloSimulation.cpp:538-613 performLifeCycle) is a chain of grid walks
// separated by short "transitions":
//
//      launch -> [peel-off walk per observer] -> pass-1 walk (tau of the whole path) -> sample tau ->
//      pass-2 walk (same path again, up to the sampled tau) -> interaction -> [peel-off walk per observer] ->
//      scatter -> pass-1 walk ...
//
// >70 % of the work is in the walks: a pointer chase through the cell records in HBM / L2 with one dependent gather and
// about 120-150 f64 instructions per step.  The loop runs over a pool of `num_slots` concurrently live histories whose
// state lives in HBM (struct of arrays, pmc_device.h SlotArrays), divided into slot groups with a stream each; ONE
// generation of a group = one scattering cycle of every live slot of the group, and is five kernels (generation 6,
// DESIGN.md section 4):
//
//   walkPropKernel    (pmc_walk_tree.inc, octree) the propagation walk of every slot: pass 1 over the whole path, the
//                     sampled optical depth between the passes, pass 2 up to it -- from the first recorded segments in
//                     LDS where the interaction lies within them.  Persistent wavefronts, one walk per lane, per-lane
//                     directions; lanes whose walk has ended are served together in bookkeeping rounds.
//   walkPeelKernel2   (pmc_walk_tree.inc, octree; one launch per observer, on a side stream next to the propagation
//                     kernel) the peel-off walks towards one observer: the direction lives in scalar registers, every
//                     wave keeps a queue of task records in LDS from which a lane takes its next walk by itself.
//                     (walkKernel in pmc_walk.inc is the generic form for Cartesian and Voronoi grids: all walks of a
//                     slot one after the other in one lane.)
//   transitionKernel  (pmc_transition.inc) one lane per live slot, every wave working through its own compacted list:
//                     detection of the cycle's peel-off packets (FluxRecorder::detect: privatised SED bins and a
//                     claim-once hot-bin table in LDS in front of f64 atomics, per-history statistics lists), the
//                     interaction the propagation walk found (weights, termination), HG scattering from the slot's
//                     Philox stream keyed by (seed, history index) (include/pmc_philox.h), and the start state of every
//                     walk of the next cycle (task records, pmc_device.h TaskArrays).
//   endedScanKernel   exclusive scan of the per-tile counts of histories that ended: the history index a slot takes up
//                     next is a pure function of the slot order.
//   launchKernel      one lane per ended history: flush of its statistics list, SourceSystem::launch of the next
//                     history, first cycle.
//
// The host enqueues generations until no slot is alive (pmc_api.hip); one 8-byte readback per generation and group tells
// it when a group is done, and the kernels of the other groups keep the device busy meanwhile.
//
// The reference stores the whole path (<= 1000 x 40 B per thread) and binary-searches the interaction point
// (SpatialGridPath.cpp:164-206).  Here the path is walked twice with bit-identical arithmetic instead: pass 1 yields
// tau_path, pass 2 stops in the segment whose cumulative tau exceeds the sampled value and interpolates exactly as
// findInteractionPoint does.
//
// Arithmetic is IEEE double with contraction off (-ffp-contract=off): the reference build has no FMA, and the
// traversal must produce the same (m, ds) sequence bit for bit (pmc_trace_ray).  The only fused operations are the
// explicit ones in exactQuotient (pmc_walk.inc), which reproduce IEEE division.
//
// Octree traversal (TreeSpatialGrid.cpp:132-217): the reference hops through per-wall neighbour lists of heap nodes.
// Here a walk step gathers ONE 32-byte record of the cell it is in (pmc_device.h CellRec: density + the six links
// through its walls; two 16-byte loads of one sector), requested as soon as the cell is known, i.e. before the arithmetic
// that enters the cell; wall coordinates come from a per-axis table staged in LDS (exactly the reference's doubles).  A
// link names the leaf that covers the whole wall (same size or coarser) together with its size exponent, so that the
// box follows from the packed indices of the current cell without a load; or the same-size internal node (finer
// neighbours), from which the child follows from the index bits of the position -- without a load when the node's
// children are all leaves (consecutive cells in depth-first order).  This gives the reference's answer whenever the new
// position lies strictly inside the leaf found; in every other case (position on a shared boundary, corner overshoot,
// rounding, grid boundary) the code falls back to the literal reference algorithm on the neighbour lists kept in HBM in
// the reference's order, followed by the reference's top-down search and next-after escape.

#include "pmc_device.h"
#include "../../include/pmc_philox.h"
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <float.h>
#include <math.h>
#include <algorithm>
#include <mutex>
#include <type_traits>

#ifndef PMC_WALK_REFILL
    #define PMC_WALK_REFILL 40  // waiting (idle or pending) lanes in a wave that trigger a service round (a round costs
                                // several hundred instructions whatever the number of lanes it serves)
#endif
#ifndef PMC_PEEL_REFILL
    #define PMC_PEEL_REFILL 32  // octree peel-off kernel: waiting lanes that trigger a service round (16 / 24 / 32 with the
                                // propagation kernel at 24 / 32 / 40: 413 / 387 / 376 ms per 5e7 packets, profiles/README.md)
#endif
#ifndef PMC_PROP_REFILL
    #define PMC_PROP_REFILL 40  // octree propagation kernel: likewise (its round includes the pass-1 -> pass-2 sampling)
#endif
#ifndef PMC_PEEL_BLOCK
    #define PMC_PEEL_BLOCK 768  // lanes per workgroup of the peel-off kernel (one coordinate table in LDS per workgroup): twelve waves, three per
                                // SIMD (107 registers).  On the sorted records with the round-5 step 512 / 640 / 768 / 896 / 1024 lanes: 27.1 / 26.3 /
                                // 24.5 / 25.1 / 25.9 ms per 2e7 packets in series; with the slot groups overlapped and 24 Mi slots 2.05 / 2.08 / 2.10e8
                                // packets/s for 512 / 640 / 768 (profiles/sweeps/r05_c2_peel_block.txt, r05_d6_peel768_24m.txt; with 8 Mi slots the
                                // larger workgroup gained nothing overlapped)
#endif
#ifndef PMC_PEEL_MIN_WAVES
    #define PMC_PEEL_MIN_WAVES 6  // waves per SIMD the peel-off kernel's register budget must allow (<= 80 VGPRs; it uses 75,
                                  // and must not spill: see treeSlowStep)
#endif
#ifndef PMC_PEEL2_MIN_WAVES
    #define PMC_PEEL2_MIN_WAVES 4  // peel-off kernel with the next task in registers (<= 128 VGPRs)
#endif
#ifndef PMC_PEEL2_REFILL
    #define PMC_PEEL2_REFILL 8  // peel-off kernel with task queues: waiting lanes that trigger a round (a round is cheap: the
                                // records come from LDS)
#endif
#ifndef PMC_PEEL_QCAP
    #define PMC_PEEL_QCAP 64  // task records per wave queue of the peel-off kernel (a power of two >= 64: refilled 64 at a time, when it is empty;
                              // 64 or 128 entries: the same times, and twelve queues of 64 fit next to the table of a 12-level octree)
#endif
#ifndef PMC_PROP_BLOCK
    #define PMC_PROP_BLOCK 768  // lanes per workgroup of the propagation kernel: ONE workgroup of twelve waves per CU -- three per SIMD, what
                                // the kernel's 157 registers allow -- sharing one coordinate table (24.6 KB at ten levels + 144 bytes of pass-1 checkpoints per lane = 135 KB).  With the slot groups overlapped, 1e8
                                // packets: 256 lanes (one to three workgroups per CU) 624-626 ms, 512: 645, 768: 602-608, 1024: 638-642
                                // (profiles/sweeps/r03_batch25_sweep.txt, r03_batch26_sweep.txt)
#endif
#ifndef PMC_PROP_MIN_WAVES
    #define PMC_PROP_MIN_WAVES 3  // likewise for the propagation kernel (<= 168 VGPRs; it uses 135-143)
#endif
#ifndef PMC_WALK_MIN_WAVES
    #define PMC_WALK_MIN_WAVES 3  // waves per SIMD the generic walk kernel's register budget must allow: the Voronoi walk takes 190
                                  // registers left to itself (two waves per SIMD) and runs 3 % faster with three (168 registers, 92
                                  // bytes of scratch): its visits are three dependent round trips
#endif
#ifndef PMC_LAUNCH_MIN_WAVES
    #define PMC_LAUNCH_MIN_WAVES 2  // launch kernel (statistics flush + source sampling)
#endif
#ifndef PMC_CYCLE_MIN_WAVES
    #define PMC_CYCLE_MIN_WAVES 4  // cycle start kernel (first cell and first exit distance of every walk of a cycle): 113 registers
#endif
#ifndef PMC_TRANSITION_MIN_WAVES
    #define PMC_TRANSITION_MIN_WAVES 2  // likewise for the transition and launch kernels (256 registers: two waves per SIMD; left to
                                        // itself the compiler takes 262 and halves the occupancy)
#endif
#ifndef PMC_VORO_PAIRS
    #define PMC_VORO_PAIRS 1  // Voronoi walk reads the per-(cell, neighbour) table DevScene::vpair instead of index -> site
#endif
#ifndef PMC_VORO_UNROLL
    #define PMC_VORO_UNROLL 8  // Voronoi walk: neighbours whose gathers are in flight together (pmc_walk.inc voroEnter)
#endif
#ifndef PMC_VPEEL_MIN_WAVES
    #define PMC_VPEEL_MIN_WAVES 4  // waves per SIMD the Voronoi peel-off kernel's register budget must allow
#endif
#ifndef PMC_VPROP_MIN_WAVES
    #define PMC_VPROP_MIN_WAVES 3  // waves per SIMD the Voronoi propagation kernel's register budget must allow
#endif
#ifndef PMC_VPROP_STEPS
    #define PMC_VPROP_STEPS 4  // scan steps of the Voronoi propagation kernel between two round checks = the tick of its checkpoints (3 / 4 / 6 / 8 / 12 on configs[4]: 514.7 / 514.9 / 516.7 / 518 / 524 ms of walk kernels per 2e7 packets)
#endif
#ifndef PMC_VPROP_ROWS
    #define PMC_VPROP_ROWS 6  // groups of PMC_VORO_RUN_LANES entries that the lanes of a propagation walk request together (one round trip)
#endif
#ifndef PMC_VPROP_REFILL
    #define PMC_VPROP_REFILL 16  // waiting lanes in a wave that trigger a service round of the Voronoi propagation kernel
#endif
#ifndef PMC_VPEEL_ROWS
    #define PMC_VPEEL_ROWS 4  // groups of PMC_VORO_RUN_LANES entries that the lanes of a walk request together (one round trip)
#endif
#ifndef PMC_VPEEL_REFILL
    #define PMC_VPEEL_REFILL 8  // waiting lanes (PMC_VORO_RUN_LANES per walk) in a wave that trigger a service round of the Voronoi peel-off kernel (2 / 4 / 8 / 16 / 32 on configs[4]: 551 / 549 / 549 / 558 / 593 ms of walk kernels per 2e7 packets)
#endif
#ifndef PMC_VORO_RUN_FIRST
    #define PMC_VORO_RUN_FIRST 4  // Voronoi peel-off walk: entries requested together with the header of a cell's run (voroEnterRun)
#endif
#ifndef PMC_VORO_CULL_ROUND
    #define PMC_VORO_CULL_ROUND 6  // Voronoi walk: neighbours per round of the masked exit search
#endif
#ifndef PMC_WALK_STEPS
    #define PMC_WALK_STEPS 8  // steps between two round checks (4 / 8 / 16 with 256 slots per claim: 597 / 592 / 594 ms per 1e8 packets)
#endif
#ifndef PMC_TRANSITION_BLOCK
    #define PMC_TRANSITION_BLOCK 256  // lanes per workgroup of the transition kernel: 256 measured 4 % faster than 512
                                      // (profiles/README.md); small enough to share a CU with the walk kernel
#endif
static_assert(PMC_TRANSITION_BLOCK <= 256 && PMC_TRANSITION_BLOCK % 64 == 0,
              "the per-wave slot lists of the transition and launch kernels are laid out for at most four waves per workgroup");
#ifndef PMC_SCAN_THREADS
    #define PMC_SCAN_THREADS 256  // lanes of the one-workgroup scan kernels (endedScanKernel, rfScanKernel): see endedScanKernel
#endif
#ifndef PMC_TASK_CHUNK
    #define PMC_TASK_CHUNK 256  // slots a wave takes from the global cursor at a time (64 / 128 / 256 / 512 / 1024: 611 / 597 / 592 / 600 /
                                // 623 ms per 1e8 packets, profiles/sweeps/r03_batch32_sweep.txt, r03_batch33_sweep.txt)
#endif

// Builds with an ablation / perturbation / experiment macro (tuning aids: they change results or add work) identify themselves:
// pmc_create prints a warning for such a library (pmc_api.hip)
#if defined(PMC_ABLATE_REFINE) || defined(PMC_ABLATE_SLOW) || defined(PMC_ABLATE_RF_MATH) || defined(PMC_ABLATE_RF_LOG) || defined(PMC_ABLATE_FRAMEADD)      \
    || defined(PMC_ABLATE_HOTBINS) || defined(PMC_ABLATE_DETECT) || defined(PMC_ABLATE_STATS) || defined(PMC_ABLATE_LAUNCH_DUST)                             \
    || defined(PMC_ABLATE_LAUNCH_BINS) || defined(PMC_ABLATE_LAUNCH_FLUSH) || defined(PMC_EXPERIMENT_FOLD_OCTANT) || defined(PMC_EXPERIMENT_PROP_SGN0) || defined(PMC_PERTURB_VALU)               \
    || defined(PMC_PERTURB_GATHER) || defined(PMC_PERTURB_LDS)
extern "C" int pmcExperimentBuild(void) { return 1; }
#else
extern "C" int pmcExperimentBuild(void) { return 0; }
#endif
extern "C" const char* pmcTune(const char* name);  // tuning switches (pmc_api.hip, include/pmc_tuning.h)

// the scene of every live context, in constant memory: all accesses are scalar loads
__constant__ DevScene c_scene[PMC_MAX_CONTEXTS];

namespace
{
    enum Mode : int { MODE_PASS1 = 0, MODE_PASS2 = 1, MODE_PEEL = 2, MODE_NONE = 3 };
    enum Grid : int { GRID_CART = PMC_GRID_CARTESIAN, GRID_TREE = PMC_GRID_OCTREE, GRID_VORO = PMC_GRID_VORONOI };
    constexpr int MODE_ALIVE = 1 << 5;
    constexpr int MODE_ENDED = 1 << 6;  // the history of the slot has ended: the launch kernel takes up the next one

    // ------------------------------------------------------------------------------------------------
    struct Rng
    {
        uint32_t h0, h1, block, have;
        double spare;
    };
    __device__ __forceinline__ double rngUniform(Rng& g, uint64_t seed)
    {
        if (g.have)
        {
            g.have = 0;
            return g.spare;
        }
        uint32_t c[4] = {g.h0, g.h1, g.block, 0x504d4331u};
        g.block += 1;
        pmc_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        g.spare = pmc_bits_to_unit(c[2], c[3]);
        g.have = 1;
        return pmc_bits_to_unit(c[0], c[1]);
    }

    __device__ __forceinline__ void loadRng(const SlotArrays& A, int slot, Rng& rng)
    {
        const uint64_t h = A.history[slot];
        const uint32_t rb = A.rngBlock[slot];
        rng.h0 = (uint32_t)h;
        rng.h1 = (uint32_t)(h >> 32);
        rng.block = rb >> 1;
        rng.have = rb & 1;
        rng.spare = A.rngSpare[slot];
    }
    __device__ __forceinline__ void storeRng(const SlotArrays& A, int slot, const Rng& rng)
    {
        A.rngBlock[slot] = (rng.block << 1) | (rng.have & 1);
        A.rngSpare[slot] = rng.spare;
    }

    __device__ __forceinline__ int locateBasic(const double* xv, double x, int n)
    {
        int jl = -1, ju = n;
        while (ju - jl > 1)
        {
            int jm = (ju + jl) >> 1;
            if (x < xv[jm])
                ju = jm;
            else
                jl = jm;
        }
        return jl;
    }
    __device__ __forceinline__ int locateClip(const double* xv, int n, double x)
    {
        if (x < xv[0]) return 0;
        return locateBasic(xv, x, n - 1);
    }
    __device__ __forceinline__ int locate(const double* xv, int n, double x)
    {
        if (x == xv[n - 1]) return n - 2;
        return locateBasic(xv, x, n);
    }

    __device__ __forceinline__ unsigned long long waveSum(uint32_t value)
    {
        unsigned long long v = value;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        return v;
    }

#include "pmc_walk.inc"
#include "pmc_walk_tree.inc"
#include "pmc_transition.inc"
}

// ---------------------------------------------------------------------------------------------------
// launch wrappers used by pmc_api.hip

extern "C" hipError_t pmcUploadScene(int slot, const DevScene* scene, hipStream_t stream)
{
    return hipMemcpyToSymbolAsync(HIP_SYMBOL(c_scene), scene, sizeof(DevScene), size_t(slot) * sizeof(DevScene),
                                  hipMemcpyHostToDevice, stream);
}

// the dynamic-LDS limit is a property of the kernel, not of a context: it is only ever raised (a small scene created
// after a large one must not lower the limit under the live context)
// LDS bytes of the task queues of one peel-off workgroup (walkPeelKernel2)
static size_t pmcPeelQueueBytes()
{
    return size_t(PMC_PEEL_BLOCK / 64) * PEEL_QBYTES;
}
// the peel-off kernel with task queues: one instantiation per sign octant of the observer's direction (bit a of sgn: k_a < 0)
typedef void (*PeelKernel2)(int, int, int, int, int, int, const int*, PeelSortedArgs);
static PeelKernel2 peelKernel2For(int wide, int sgn)
{
    static const PeelKernel2 narrow[8] = {walkPeelKernel2<false, 0>, walkPeelKernel2<false, 1>, walkPeelKernel2<false, 2>, walkPeelKernel2<false, 3>,
                                          walkPeelKernel2<false, 4>, walkPeelKernel2<false, 5>, walkPeelKernel2<false, 6>, walkPeelKernel2<false, 7>};
    static const PeelKernel2 wider[8] = {walkPeelKernel2<true, 0>, walkPeelKernel2<true, 1>, walkPeelKernel2<true, 2>, walkPeelKernel2<true, 3>,
                                         walkPeelKernel2<true, 4>, walkPeelKernel2<true, 5>, walkPeelKernel2<true, 6>, walkPeelKernel2<true, 7>};
    return wide ? wider[sgn & 7] : narrow[sgn & 7];
}

extern "C" hipError_t pmcConfigureKernels(size_t walkLds, size_t transitionLds)
{
    // (contexts are created concurrently by one host thread per device; the attribute is per device, so every pmc_create
    // sets it on its own current device)
    static std::mutex lock;
    std::lock_guard<std::mutex> guard(lock);
    static size_t walkMax = 0, transitionMax = 0;
    walkMax = std::max(walkMax, walkLds);
    transitionMax = std::max(transitionMax, transitionLds);
    {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rfReduceKernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(sizeof(double) << PMC_RF_BUCKET_BITS));
        if (e != hipSuccess) return e;
        const int scatterLds = (int)(size_t(RF_TILE) * (sizeof(double) + sizeof(uint32_t)) + size_t(2) * RF_MAX_PARTS * sizeof(uint32_t));
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rfScatterKernel<PMC_RF_BUCKET_BITS>), hipFuncAttributeMaxDynamicSharedMemorySize, scatterLds);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rfScatterKernel<PMC_STAT_BUCKET_BITS>), hipFuncAttributeMaxDynamicSharedMemorySize, scatterLds);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&statReduceKernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(5 * sizeof(double) << PMC_STAT_BUCKET_BITS));
        if (e != hipSuccess) return e;
    }
    const struct
    {
        const void* kernel;
        size_t lds;
    } all[] = {{reinterpret_cast<const void*>(&walkPeelKernel<false, false>), walkMax},
               {reinterpret_cast<const void*>(&walkPeelKernel<true, false>), walkMax},
               {reinterpret_cast<const void*>(&walkPeelKernel<false, true>), walkMax},
               {reinterpret_cast<const void*>(&walkPeelKernel<true, true>), walkMax},
               {reinterpret_cast<const void*>(&walkPropKernel<false, false, false, false>), std::min(walkMax + 16 + PROP_CKPT_BYTES, size_t(160) * 1024)},
               {reinterpret_cast<const void*>(&walkPropKernel<false, true, false, false>), std::min(walkMax + 16 + PROP_CKPT_BYTES, size_t(160) * 1024)},
               {reinterpret_cast<const void*>(&walkPropKernel<true, false, false, false>), std::min(walkMax + 16 + PROP_CKPT_BYTES, size_t(160) * 1024)},
               {reinterpret_cast<const void*>(&walkPropKernel<true, true, false, false>), std::min(walkMax + 16 + PROP_CKPT_BYTES, size_t(160) * 1024)},
               {reinterpret_cast<const void*>(&walkPropKernel<false, false, true, false>), walkMax},
               {reinterpret_cast<const void*>(&walkPropKernel<false, true, true, false>), walkMax},
               {reinterpret_cast<const void*>(&walkPropKernel<true, false, true, false>), walkMax},
               {reinterpret_cast<const void*>(&walkPropKernel<true, true, true, false>), walkMax},
               {reinterpret_cast<const void*>(&walkPropKernel<false, false, false, true>), walkMax},
               {reinterpret_cast<const void*>(&walkPropKernel<false, false, true, true>), walkMax},
               {reinterpret_cast<const void*>(&walkPropKernel<false, true, false, true>), walkMax},
               {reinterpret_cast<const void*>(&walkPropKernel<false, true, true, true>), walkMax},
               {reinterpret_cast<const void*>(&walkPropKernel<true, false, false, true>), walkMax},
               {reinterpret_cast<const void*>(&walkPropKernel<true, false, true, true>), walkMax},
               {reinterpret_cast<const void*>(&walkPropKernel<true, true, false, true>), walkMax},
               {reinterpret_cast<const void*>(&walkPropKernel<true, true, true, true>), walkMax},
               {reinterpret_cast<const void*>(&traceTreeKernel<false, false>), walkMax},
               {reinterpret_cast<const void*>(&traceTreeKernel<false, true>), walkMax},
               {reinterpret_cast<const void*>(&traceTreeKernel<true, false>), walkMax},
               {reinterpret_cast<const void*>(&traceTreeKernel<true, true>), walkMax},
               {reinterpret_cast<const void*>(&walkKernel<GRID_CART, false, false, false>), walkMax},
               {reinterpret_cast<const void*>(&walkKernel<GRID_CART, true, false, false>), walkMax},
               {reinterpret_cast<const void*>(&walkKernel<GRID_CART, false, true, false>), walkMax},
               {reinterpret_cast<const void*>(&walkKernel<GRID_CART, true, true, false>), walkMax},
               {reinterpret_cast<const void*>(&walkKernel<GRID_CART, false, false, true>), walkMax},
               {reinterpret_cast<const void*>(&walkKernel<GRID_CART, false, true, true>), walkMax},
               {reinterpret_cast<const void*>(&walkKernel<GRID_CART, true, false, true>), walkMax},
               {reinterpret_cast<const void*>(&walkKernel<GRID_CART, true, true, true>), walkMax},
               {reinterpret_cast<const void*>(&walkKernel<GRID_VORO, false, false, true>), walkMax},
               {reinterpret_cast<const void*>(&walkKernel<GRID_VORO, false, true, true>), walkMax},
               {reinterpret_cast<const void*>(&walkKernel<GRID_VORO, true, false, true>), walkMax},
               {reinterpret_cast<const void*>(&walkKernel<GRID_VORO, true, true, true>), walkMax},
               {reinterpret_cast<const void*>(&traceRayKernel<GRID_CART>), walkMax},
               {reinterpret_cast<const void*>(&transitionKernel), transitionMax},
               {reinterpret_cast<const void*>(&launchKernel), transitionMax},
               {reinterpret_cast<const void*>(&cycleStartKernel<GRID_TREE>), walkMax + 16 + size_t(PMC_SORT_OBS) * PMC_PEEL_TILES * PMC_PEEL_TILES * sizeof(uint32_t)},
               {reinterpret_cast<const void*>(&cycleStartKernel<GRID_CART>), walkMax + 16 + size_t(PMC_SORT_OBS) * PMC_PEEL_TILES * PMC_PEEL_TILES * sizeof(uint32_t)},
               {reinterpret_cast<const void*>(&cycleStartKernel<GRID_VORO>), walkMax + 16 + size_t(PMC_SORT_OBS) * PMC_PEEL_TILES * PMC_PEEL_TILES * sizeof(uint32_t)},
               {reinterpret_cast<const void*>(&walkKernel<GRID_VORO, false, false, false>), walkMax},
               {reinterpret_cast<const void*>(&walkKernel<GRID_VORO, true, false, false>), walkMax},
               {reinterpret_cast<const void*>(&walkKernel<GRID_VORO, false, true, false>), walkMax},
               {reinterpret_cast<const void*>(&walkKernel<GRID_VORO, true, true, false>), walkMax},
               {reinterpret_cast<const void*>(&traceRayKernel<GRID_VORO>), walkMax}};
    for (const auto& k : all)
    {
        hipError_t e = hipFuncSetAttribute(k.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k.lds);
        if (e != hipSuccess) return e;
    }
    for (int wide = 0; wide < 2; ++wide)
        for (int sgn = 0; sgn < 8; ++sgn)
        {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(peelKernel2For(wide, sgn)), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)std::min(walkMax + 16 + pmcPeelQueueBytes(), size_t(160) * 1024));
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}

// resident workgroups per CU of a walk kernel: kind 0 = generic (Cartesian / Voronoi), 1 = octree peel-off, 2 = octree
// propagation
extern "C" int pmcWalkBlocksPerCU(int gridKind, int kind, int wide, int block, size_t ldsBytes)
{
    int n = 0;
    hipError_t e;
    if (gridKind == PMC_GRID_OCTREE && kind == 1)
        e = wide ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&walkPeelKernel<true, false>), block, ldsBytes)
                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&walkPeelKernel<false, false>), block, ldsBytes);
    else if (gridKind == PMC_GRID_OCTREE)
    {
        // (with the pass-1 checkpoints of pmcLaunchProp)
        const size_t trimOffset = (ldsBytes + 15) & ~size_t(15);
        if (!pmcTune("PMC_PROP_NO_CHECKPOINTS") && trimOffset + PROP_CKPT_BYTES <= size_t(160) * 1024) ldsBytes = trimOffset + PROP_CKPT_BYTES;
        e = wide ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&walkPropKernel<true, false, false, false>), block, ldsBytes)
                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&walkPropKernel<false, false, false, false>), block, ldsBytes);
    }
    else if (gridKind == PMC_GRID_VORONOI)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&walkKernel<GRID_VORO, false, false, false>), block, ldsBytes);
    else
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&walkKernel<GRID_CART, false, false, false>), block, ldsBytes);
    return e == hipSuccess ? n : 0;
}

extern "C" int pmcPeelBlock(void)
{
    return PMC_PEEL_BLOCK;
}
extern "C" int pmcPropBlock(void)
{
    return PMC_PROP_BLOCK;
}


// walks of the task records [taskBase, taskBase + numTaskRecords) of one slot group on a Cartesian or Voronoi grid;
// taskCounter = index of the group's (zeroed) cursor
extern "C" hipError_t pmcLaunchWalk(int slot, int gridKind, int storeRf, int taskBase, int numTaskRecords, int taskCounter,
                                    uint64_t seed, int grid, int block, size_t ldsBytes, const WalkStreamArgs* tasks, hipStream_t stream)
{
    WalkStreamArgs ws;
    std::memset(&ws, 0, sizeof(ws));
    if (tasks) ws = *tasks;
    // (the radiation-field and the explicit-absorption flavours are separate instantiations: the plain photon loop pays nothing for
    // them; storeRf: bit 0 = the radiation field is stored, bit 1 = explicit absorption)
    // (bit 2 = several medium components)
    const int flavour = storeRf & 7;
    typedef void (*Kernel)(int, int, int, int, uint64_t, WalkStreamArgs);
    static const Kernel cart[8] = {walkKernel<GRID_CART, false, false, false>, walkKernel<GRID_CART, true, false, false>, walkKernel<GRID_CART, false, true, false>,
                                   walkKernel<GRID_CART, true, true, false>,   walkKernel<GRID_CART, false, false, true>, walkKernel<GRID_CART, true, false, true>,
                                   walkKernel<GRID_CART, false, true, true>,   walkKernel<GRID_CART, true, true, true>};
    static const Kernel voro[8] = {walkKernel<GRID_VORO, false, false, false>, walkKernel<GRID_VORO, true, false, false>, walkKernel<GRID_VORO, false, true, false>,
                                   walkKernel<GRID_VORO, true, true, false>,   walkKernel<GRID_VORO, false, false, true>, walkKernel<GRID_VORO, true, false, true>,
                                   walkKernel<GRID_VORO, false, true, true>,   walkKernel<GRID_VORO, true, true, true>};
    const Kernel kernel = gridKind == PMC_GRID_VORONOI ? voro[flavour] : cart[flavour];
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), ldsBytes, stream, slot, taskBase, numTaskRecords, taskCounter, seed, ws);
    return hipGetLastError();
}

// Voronoi: the peel-off walks of the slots in `list` (their number: *count, device memory) towards the observer of task record `rec`, whose table of
// runs is DevScene::vobs_run[tab]; xcdCursor: `segments` (8 or 1) zeroed cursors over equal parts of the list
extern "C" int pmcVoroPeelWavesPerSimd(void) { return PMC_VPEEL_MIN_WAVES; }
extern "C" hipError_t pmcLaunchVoroPeel(int slot, int rec, int tab, const int32_t* list, const unsigned long long* count, unsigned long long* xcdCursor, int segments,
                                        int severalMedia, int grid, hipStream_t stream)
{
    if (severalMedia)
        hipLaunchKernelGGL(voroPeelKernel<true>, dim3(grid), dim3(256), 0, stream, slot, rec, tab, list, count, xcdCursor, segments);
    else
        hipLaunchKernelGGL(voroPeelKernel<false>, dim3(grid), dim3(256), 0, stream, slot, rec, tab, list, count, xcdCursor, segments);
    return hipGetLastError();
}

// Voronoi: the propagation walks (task record 0) of the slots in `list` on the table of runs DevScene::vgen_run
extern "C" int pmcVoroPropWavesPerSimd(void) { return PMC_VPROP_MIN_WAVES; }
// flavour: bit 0 radiation field, bit 1 explicit absorption, bit 2 several medium components
extern "C" hipError_t pmcLaunchVoroProp(int slot, const int32_t* list, const unsigned long long* count, unsigned long long* xcdCursor, int segments, uint64_t seed,
                                        int flavour, int grid, hipStream_t stream)
{
#define PMC_VPROP_CASE(f, RF, EA, MM) \
    case f: hipLaunchKernelGGL((voroPropKernel<RF, EA, MM>), dim3(grid), dim3(256), 0, stream, slot, list, count, xcdCursor, segments, seed); break;
    switch (flavour & 7)
    {
        PMC_VPROP_CASE(0, false, false, false)
        PMC_VPROP_CASE(1, true, false, false)
        PMC_VPROP_CASE(2, false, true, false)
        PMC_VPROP_CASE(3, true, true, false)
        PMC_VPROP_CASE(4, false, false, true)
        PMC_VPROP_CASE(5, true, false, true)
        PMC_VPROP_CASE(6, false, true, true)
        PMC_VPROP_CASE(7, true, true, true)
    }
#undef PMC_VPROP_CASE
    return hipGetLastError();
}

// octree: the peel-off walks towards observer `obs` of the slots [slotBase, slotBase + numSlots)
// does the peel-off kernel of this octree run with task queues (the form that can take sorted records)?
extern "C" int pmcPeelHasQueues(int wide, size_t ldsBytes)
{
    const bool first = pmcTune("PMC_PEEL_V1") != nullptr;
    return !first && (wide & 2) == 0 && ((ldsBytes + 15) & ~size_t(15)) + pmcPeelQueueBytes() <= size_t(160) * 1024;
}
// sorted peel-off records (pmc_device.h PeelRec): the count pass of the sort over the slots of a group, behind its transition / launch kernels; the
// cycle start kernel, launched with the SAME number of workgroups (returned through `groups`), is the scatter pass.  temp: pmcPeelSortTempBytes();
// the number of sorted records is left at pmcPeelSortedCount(temp)
constexpr int PEEL_SORT_GROUPS = 1024;
constexpr int PEEL_SORT_PARTS = PMC_PEEL_TILES * PMC_PEEL_TILES;
extern "C" size_t pmcPeelSortTempBytes()
{
    // counters (totals, then starts: 2 * parts + 1) followed by the matrix [groups][parts]
    return (size_t(2) * PEEL_SORT_PARTS + 2) * sizeof(unsigned long long) + size_t(PEEL_SORT_GROUPS) * PEEL_SORT_PARTS * sizeof(uint32_t);
}
extern "C" const unsigned long long* pmcPeelSortedCount(void* temp) { return static_cast<const unsigned long long*>(temp) + 2 * PEEL_SORT_PARTS; }
// (ps: numObs, obs, sortIndex, centre, scale set by the caller; sorted[k] / temp[k]: the records and counters of observer k)
extern "C" hipError_t pmcLaunchPeelSortCounts(int slot, int slotBase, int numSlots, PeelSortArgs* ps, PeelRec* const* sorted, int32_t* const* lists, void* const* temp,
                                              int* groups, hipStream_t stream)
{
    ps->numParts = PEEL_SORT_PARTS;
    const int numLists = ps->numObs + (ps->propIndex >= 0 ? 1 : 0);
    for (int k = 0; k < numLists; ++k)
    {
        unsigned long long* totals = static_cast<unsigned long long*>(temp[k]);
        ps->out[k] = sorted ? sorted[k] : nullptr;
        ps->listOut[k] = lists ? lists[k] : nullptr;
        ps->matrix[k] = reinterpret_cast<uint32_t*>(totals + 2 * PEEL_SORT_PARTS + 2);
        ps->start[k] = totals + PEEL_SORT_PARTS;
    }
    const int tiles = (numSlots + PEEL_SORT_TILE - 1) / PEEL_SORT_TILE;
    *groups = std::max(1, std::min(tiles, PEEL_SORT_GROUPS));
    hipLaunchKernelGGL(peelSortCountKernel, dim3(*groups), dim3(256), 0, stream, slot, slotBase, numSlots, *ps);
    for (int k = 0; k < numLists; ++k)
    {
        unsigned long long* totals = static_cast<unsigned long long*>(temp[k]);
        hipLaunchKernelGGL(peelSortOffsetsKernel, dim3((PEEL_SORT_PARTS + 15) / 16), dim3(256), 0, stream, ps->matrix[k], (uint32_t)*groups, (uint32_t)PEEL_SORT_PARTS, totals);
        hipLaunchKernelGGL(rfScanKernel, dim3(1), dim3(PMC_SCAN_THREADS), 0, stream, totals, totals + PEEL_SORT_PARTS, (uint32_t)PEEL_SORT_PARTS);
    }
    return hipGetLastError();
}

extern "C" hipError_t pmcLaunchPeel(int slot, int wide, int slotBase, int numSlots, const int* list, int cursor, int obs, int sgn, int grid, size_t ldsBytes,
                                    const PeelRec* sortedRec, const unsigned long long* sortedCount, unsigned long long* xcdCursor, hipStream_t stream)
{
    const bool first = pmcTune("PMC_PEEL_V1") != nullptr;  // (tuning aid: the form with service rounds)
    // (`wide` bit 1: several medium components -- the form with service rounds)
    const bool mm = (wide & 2) != 0;
    wide &= 1;
    // (an octree of 12 levels leaves no room for the task queues next to its coordinate table: service rounds)
    if (first || mm || ((ldsBytes + 15) & ~size_t(15)) + pmcPeelQueueBytes() > size_t(160) * 1024)
    {
        auto kernel = mm ? (wide ? walkPeelKernel<true, true> : walkPeelKernel<false, true>) : (wide ? walkPeelKernel<true, false> : walkPeelKernel<false, false>);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(PMC_PEEL_BLOCK), ldsBytes, stream, slot, slotBase, numSlots, cursor, obs, list);
    }
    else
    {
        // (the waves' task queues follow the grid tables in LDS)
        // (sgn: the sign octant of the observer's direction, DevInstrument::sgn)
        const PeelKernel2 kernel = peelKernel2For(wide, sgn);
        const size_t queueOffset = (ldsBytes + 15) & ~size_t(15);
        const PeelSortedArgs sorted = {sortedRec, sortedCount, xcdCursor};
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(PMC_PEEL_BLOCK), queueOffset + pmcPeelQueueBytes(), stream, slot, slotBase, numSlots, cursor, obs,
                           (int)queueOffset, list, sorted);
    }
    return hipGetLastError();
}

// octree: the propagation walks of the slots [slotBase, slotBase + numSlots)
extern "C" hipError_t pmcLaunchProp(int slot, int wide, int storeRf, int slotBase, int numSlots, const int* list, int cursor, uint64_t seed, int grid,
                                    size_t ldsBytes, const RfLogArgs* rfLog, hipStream_t stream)
{
    // (storeRf: bit 0 = the radiation field is stored, bit 1 = explicit absorption)
    // (bit 2 = several medium components)
    const bool ea = (storeRf & 2) != 0, mm = (storeRf & 4) != 0;
    typedef void (*Kernel)(int, int, int, int, uint64_t, int, RfLogArgs, const int*);
    static const Kernel narrow[8] = {walkPropKernel<false, false, false, false>, walkPropKernel<false, true, false, false>, walkPropKernel<false, false, true, false>,
                                     walkPropKernel<false, true, true, false>,   walkPropKernel<false, false, false, true>, walkPropKernel<false, true, false, true>,
                                     walkPropKernel<false, false, true, true>,   walkPropKernel<false, true, true, true>};
    static const Kernel wider[8] = {walkPropKernel<true, false, false, false>, walkPropKernel<true, true, false, false>, walkPropKernel<true, false, true, false>,
                                    walkPropKernel<true, true, true, false>,   walkPropKernel<true, false, false, true>, walkPropKernel<true, true, false, true>,
                                    walkPropKernel<true, false, true, true>,   walkPropKernel<true, true, true, true>};
    const Kernel kernel = wide ? wider[storeRf & 7] : narrow[storeRf & 7];
    // (the pass-1 checkpoints follow the grid tables in LDS, if there is room)
    const bool noTrim = pmcTune("PMC_PROP_NO_CHECKPOINTS") != nullptr;  // (tuning aid: pass 2 walks every path from its start)
    const size_t trimOffset = (ldsBytes + 15) & ~size_t(15);
    const bool trim = !noTrim && !ea && !mm && trimOffset + PROP_CKPT_BYTES <= size_t(160) * 1024;
    RfLogArgs none = {nullptr, nullptr, 0ull, 0, 0u};
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(PMC_PROP_BLOCK), trim ? trimOffset + PROP_CKPT_BYTES : ldsBytes, stream, slot, slotBase, numSlots, cursor,
                       seed, trim ? (int)trimOffset : -1, rfLog ? *rfLog : none, list);
    return hipGetLastError();
}

// radiation field: the log of a generation (n entries in whole chunks) partitioned by key range into (sortedKeys, sortedVals) and
// added to the table; temp = 2 * numParts + 1 counters (pmcRfTempBytes)
extern "C" size_t pmcRfTempBytes(int numParts) { return (size_t(2) * size_t(numParts) + 1) * sizeof(unsigned long long); }
extern "C" int pmcRfMaxParts() { return (int)RF_MAX_PARTS; }
// the counting sort of (key, value) pairs on key >> PMC_RF_BUCKET_BITS: n entries in whole tiles of RF_TILE -> (sortedKeys, sortedVals), pad
// keys dropped; temp = 2 * numParts + 1 counters: the partition starts are left at temp + numParts (numParts + 1 of them)
template<int BITS>
static hipError_t launchPartition(const uint32_t* keys, const double* vals, uint32_t* sortedKeys, double* sortedVals, unsigned long long n, int numParts,
                                  void* temp, int numCU, hipStream_t stream, const uint32_t* fills = nullptr)
{
    unsigned long long* cursor = static_cast<unsigned long long*>(temp);
    unsigned long long* start = cursor + numParts;
    hipError_t e = hipMemsetAsync(cursor, 0, size_t(numParts) * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    const unsigned long long tiles = n / RF_TILE;
    const unsigned grid = (unsigned)std::min<unsigned long long>(tiles, (unsigned long long)numCU * 8ull);
    hipLaunchKernelGGL(rfHistKernel<BITS>, dim3(std::max(grid, 1u)), dim3(256), 0, stream, keys, n, (uint32_t)numParts, cursor, fills);
    hipLaunchKernelGGL(rfScanKernel, dim3(1), dim3(PMC_SCAN_THREADS), 0, stream, cursor, start, (uint32_t)numParts);
    const size_t sortLds = size_t(RF_TILE) * (sizeof(double) + sizeof(uint32_t)) + size_t(2) * size_t(numParts) * sizeof(uint32_t);
    const unsigned sortGrid = (unsigned)std::min<unsigned long long>(tiles, (unsigned long long)numCU * 3ull);
    hipLaunchKernelGGL(rfScatterKernel<BITS>, dim3(std::max(sortGrid, 1u)), dim3(RF_SORT_BLOCK), sortLds, stream, keys, vals, n, (uint32_t)numParts, cursor, sortedKeys,
                       sortedVals, fills);
    return hipGetLastError();
}
extern "C" hipError_t pmcLaunchRfFlush(int slot, const uint32_t* keys, const double* vals, uint32_t* sortedKeys, double* sortedVals, unsigned long long n,
                                       int numParts, void* temp, int numCU, hipStream_t stream)
{
    hipError_t e = launchPartition<PMC_RF_BUCKET_BITS>(keys, vals, sortedKeys, sortedVals, n, numParts, temp, numCU, stream);
    if (e != hipSuccess) return e;
    const unsigned long long* start = static_cast<const unsigned long long*>(temp) + numParts;
    const size_t lds = sizeof(double) << PMC_RF_BUCKET_BITS;  // (the limit is raised per device in pmcConfigureKernels)
    const unsigned long long blocks = (n + PMC_RF_REDUCE_SPAN - 1) / PMC_RF_REDUCE_SPAN;
    hipLaunchKernelGGL(rfReduceKernel, dim3((unsigned)blocks), dim3(RF_REDUCE_BLOCK), lds, stream, slot, sortedKeys, sortedVals, start, (uint32_t)numParts);
    return hipGetLastError();
}
// statistics: the log of a slot group (n entries in whole chunks) partitioned by record range into (sortedKeys, sortedVals) and summed into the
// accumulator records; temp = 2 * numParts + 1 counters (pmcRfTempBytes)
extern "C" int pmcStatBucketBits() { return PMC_STAT_BUCKET_BITS; }
extern "C" hipError_t pmcLaunchStatFlush(int slot, const uint32_t* keys, const double* vals, uint32_t* sortedKeys, double* sortedVals, unsigned long long n,
                                         int numParts, void* temp, int numCU, const uint32_t* chunkFill, hipStream_t stream)
{
    hipError_t e = launchPartition<PMC_STAT_BUCKET_BITS>(keys, vals, sortedKeys, sortedVals, n, numParts, temp, numCU, stream, chunkFill);
    if (e != hipSuccess) return e;
    const unsigned long long* start = static_cast<const unsigned long long*>(temp) + numParts;
    const size_t lds = 5 * sizeof(double) << PMC_STAT_BUCKET_BITS;
    const unsigned long long blocks = (n + PMC_RF_REDUCE_SPAN - 1) / PMC_RF_REDUCE_SPAN;
    hipLaunchKernelGGL(statReduceKernel, dim3((unsigned)blocks), dim3(RF_REDUCE_BLOCK), lds, stream, slot, sortedKeys, sortedVals, start, (uint32_t)numParts);
    return hipGetLastError();
}
// end of a segment: statistics accumulator -> wifu arrays of the frames
extern "C" hipError_t pmcLaunchStatMerge(int slot, int blocks, hipStream_t stream)
{
    hipLaunchKernelGGL(statMergeKernel, dim3(blocks), dim3(256), 0, stream, slot);
    return hipGetLastError();
}

// transitions of the slots [slotBase, slotBase + numSlots) of slot group `group`, followed by the scan of the group's
// ended-history counts (the launch kernel's history indices)
extern "C" hipError_t pmcLaunchTransition(int slot, int slotBase, int numSlots, int group, uint64_t seed, const int* list, int listLen, int maxBlocks,
                                          size_t ldsBytes, const StatLogArgs* statLog, uint64_t count, uint64_t keep, hipStream_t stream)
{
    const StatLogArgs none = {nullptr, nullptr, 0ull, 0, nullptr, nullptr, nullptr};
    const int block = PMC_TRANSITION_BLOCK;
    // (a sparse generation: one list entry per lane; otherwise persistent workgroups over runs of 256 slots per wave)
    const int grid = std::max(1, std::min(((list ? listLen : numSlots) + block - 1) / block, maxBlocks));
    hipLaunchKernelGGL(transitionKernel, dim3(grid), dim3(block), ldsBytes, stream, slot, slotBase, numSlots, group, seed, list, listLen, statLog ? *statLog : none);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || list) return e;  // (a sparse generation retires its ended histories in the transition kernel)
    hipLaunchKernelGGL(endedScanKernel, dim3(1), dim3(PMC_SCAN_THREADS), 0, stream, slot, slotBase, numSlots, group, (unsigned long long)count,
                       (unsigned long long)keep);
    return hipGetLastError();
}

// launches of new histories into the slots of the group whose history ended (initial: into all slots of the group)
extern "C" hipError_t pmcLaunchLaunch(int slot, int slotBase, int numSlots, int group, uint64_t first, uint64_t count, uint64_t seed, int initial,
                                      int maxBlocks, size_t ldsBytes, const StatLogArgs* statLog, hipStream_t stream)
{
    const StatLogArgs none = {nullptr, nullptr, 0ull, 0, nullptr, nullptr, nullptr};
    const int grid = std::max(1, std::min((numSlots + 255) / 256, maxBlocks));
    hipLaunchKernelGGL(launchKernel, dim3(grid), dim3(256), ldsBytes, stream, slot, slotBase, numSlots, group, first, count, seed, initial, statLog ? *statLog : none);
    return hipGetLastError();
}

// the walks of the cycle that every live slot of the group is about to start (task records)
extern "C" hipError_t pmcLaunchCycleStart(int slot, int gridKind, int slotBase, int numSlots, int listCounter, int* listOut, const int* listIn,
                                          int listLen, int maxBlocks, size_t ldsBytes, const PeelSortArgs* sort, hipStream_t stream)
{
    PeelSortArgs ps;
    std::memset(&ps, 0, sizeof(ps));
    if (sort) ps = *sort;
    if (ps.numObs > 0)
    {
        // (the sorts' cursors follow the grid tables in LDS; as many workgroups as the sort's count pass had: maxBlocks is that number then)
        ps.ldsOffset = int((ldsBytes + 15) & ~size_t(15));
        ldsBytes = size_t(ps.ldsOffset) + size_t(ps.numObs + (ps.propIndex >= 0 ? 1 : 0)) * PEEL_SORT_PARTS * sizeof(uint32_t);
    }
    const int grid = std::max(1, std::min(((listIn ? listLen : numSlots) + 255) / 256, maxBlocks));
    if (gridKind == PMC_GRID_OCTREE)
        hipLaunchKernelGGL(cycleStartKernel<GRID_TREE>, dim3(grid), dim3(256), ldsBytes, stream, slot, slotBase, numSlots, listCounter, listOut, listIn, listLen, ps);
    else if (gridKind == PMC_GRID_VORONOI)
        hipLaunchKernelGGL(cycleStartKernel<GRID_VORO>, dim3(grid), dim3(256), ldsBytes, stream, slot, slotBase, numSlots, listCounter, listOut, listIn, listLen, ps);
    else
        hipLaunchKernelGGL(cycleStartKernel<GRID_CART>, dim3(grid), dim3(256), ldsBytes, stream, slot, slotBase, numSlots, listCounter, listOut, listIn, listLen, ps);
    return hipGetLastError();
}

// one ray through the grid; octree: `uniform` picks the flavour of the step the ray is traced with (direction in scalar
// registers as in the peel-off kernel, or in vector registers as in the propagation kernel; kdev = the direction in
// device memory)
extern "C" hipError_t pmcLaunchTrace(int slot, int gridKind, int wide, int uniform, const double r[3], const double k[3],
                                     const double* kdev, int32_t* m, double* ds, int32_t cap, int32_t* n, size_t ldsBytes,
                                     hipStream_t stream)
{
    if (gridKind == PMC_GRID_OCTREE)
    {
        auto kernel = wide ? (uniform ? traceTreeKernel<true, true> : traceTreeKernel<true, false>)
                           : (uniform ? traceTreeKernel<false, true> : traceTreeKernel<false, false>);
        hipLaunchKernelGGL(kernel, dim3(1), dim3(64), ldsBytes, stream, slot, r[0], r[1], r[2], k[0], k[1], k[2], kdev, m, ds, cap, n);
    }
    else if (gridKind == PMC_GRID_VORONOI)
        hipLaunchKernelGGL(traceRayKernel<GRID_VORO>, dim3(1), dim3(64), ldsBytes, stream, slot, r[0], r[1], r[2], k[0], k[1],
                           k[2], m, ds, cap, n);
    else
        hipLaunchKernelGGL(traceRayKernel<GRID_CART>, dim3(1), dim3(64), ldsBytes, stream, slot, r[0], r[1], r[2], k[0], k[1],
                           k[2], m, ds, cap, n);
    return hipGetLastError();
}
