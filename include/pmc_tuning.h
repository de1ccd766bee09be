/* pmc_tuning.h -- tuning aids of the MI355X photon-packet engine (libpmc.so): NOT part of the drop-in boundary (pmc.h).
   Nothing here has a counterpart in the reference; the parity tests use the switches to run one scene through two code paths
   of the engine (lists of live slots or not, sorted peel-off records or not, ...), tools/sweep.py and bench.py use them to time
   kernels in series.  The library reads none of this from the environment. */
#ifndef PMC_TUNING_H
#define PMC_TUNING_H

#include "pmc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A process-wide table of named switches, consulted by pmc_create / pmc_run_primary where the names below are listed in
   skirt9_amd/csrc/pmc_api.hip and pmc_kernels.hip (a switch is "set" when it has a value; numeric ones are parsed with atoi):
     PMC_SERIAL_WALKS, PMC_TIMING_DUMP, PMC_GEN_DUMP, PMC_PROFILE_DUMP        measurement: kernels of a group in series, dumps to stderr
     PMC_NO_LIVE_LISTS, PMC_NO_PEEL_SORT, PMC_NO_PROP_SORT, PMC_NO_XCD_AFFINITY, PMC_NO_MONO,
     PMC_RF_ATOMICS, PMC_RF_LOG_PER_SLOT, PMC_STAT_ATOMICS, PMC_STAT_LOG_ENTRIES, PMC_PEEL_V1, PMC_PROP_NO_CHECKPOINTS (octree: at run; Voronoi: at pmc_create),
     PMC_VORO_NO_CULL, PMC_VORO_NO_OBSERVER_LISTS, PMC_VORO_CONE_CULL_ONLY, PMC_VORO_NO_PEEL_KERNEL, PMC_VORO_NO_PROP_KERNEL, PMC_VORO_NO_CONE_TABLES, PMC_VORO_NO_DEFERRED_SCAN, PMC_VORO_LINK_COUNT_MAX, PMC_VPROP_XCD_SEGMENTS, PMC_VORO_WALKS_IN_SERIES, PMC_VPROP_BLOCKS_PER_CU, PMC_VPEEL_BLOCKS_PER_CU,
     PMC_POISON_ALLOCATIONS (test aid: device arrays the engine does not initialise are filled with 0xA5 bytes)                          alternative code paths (cross-checks, A/B)
     PMC_WALK_BLOCKS_PER_CU, PMC_PEEL_BLOCKS_PER_CU, PMC_LAUNCH_BLOCKS_PER_CU,
     PMC_CYCLE_BLOCKS_PER_CU, PMC_TRANSITION_BLOCKS_PER_CU,
     PMC_LIST_TASKS_PER_LANE, PMC_CELL_ORDER, PMC_CELL_SHUFFLE                launch geometry, order of the cell table
   value == NULL removes the switch.  Returns PMC_OK. */
int pmc_tuning_set(const char* name, const char* value);
/* removes every switch */
void pmc_tuning_clear(void);

/* threads per workgroup and workgroups of the persistent walk kernel (0 = default) */
int pmc_set_launch(pmc_ctx* ctx, int32_t block, int32_t grid);

/* counted work of the octree walk kernels since create/reset: a wave-step is one pass of a wavefront through the step
   code, a lane-step one cell visit by one lane (lane_steps / (64 wave_steps) = the fraction of the lanes that held a walk);
   rounds = bookkeeping rounds (finished walks stored, next walks taken up).  Propagation lane-steps include the second
   pass over a forced-scattering path.  No reference counterpart (roofline inputs). */
typedef struct pmc_walk_work_values
{
    uint64_t peel_wave_steps, peel_lane_steps, peel_rounds;
    uint64_t prop_wave_steps, prop_lane_steps, prop_rounds;
} pmc_walk_work_values;
int pmc_walk_work(pmc_ctx* ctx, pmc_walk_work_values* out);
/* Tuning aid: DEVICE addresses of the octree walk's hot table (see skirt9_amd/csrc/pmc_device.h) and of the first cells of the propagation walks the last generations left in
   the task records -- so that profiles/microbench/bridge.hip can replay the walk's memory accesses on the scene's own tables.
   No reference counterpart; nothing in the product reads it. */
typedef struct pmc_debug_table_values
{
    const void* cell_table;   /* [cell_slots] 32-byte records (pmc_device.h CellRec) */
    int64_t cell_slots;
    int64_t loose_base;       /* = cell_slots (the octet-line table of profiles/experiments/r04_octet_line_table.patch: first loose leaf) */
    const int32_t* task_cell; /* [num_slots] first cell of the propagation walk of every slot (stale after the segment's end) */
    int64_t num_slots;
} pmc_debug_table_values;
int pmc_debug_tables(pmc_ctx* ctx, pmc_debug_table_values* out);

#ifdef __cplusplus
}
#endif
#endif
