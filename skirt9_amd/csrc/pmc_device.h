// pmc_device.h -- device-side data layout of the MI355X photon-packet engine (shared by host API and kernels).
//
// HBM layout (read-only during a segment, replicated per GPU):
//   octree cells    CellRec[num_cells]   the hot table of the walk step (32 B per cell: density + the six wall links), cells
//                                        in depth-first order of the tree; LeafRec[num_cells] the cold one (box code, density)
//   octree nodes    NodeRec[num_internal] 64-byte record per non-leaf node: box code + 8 child links (descent only)
//   coord table     double[3][2^Lmax+1]  the reference's wall coordinates per axis and dyadic index; staged in LDS (Lmax <= 12)
//   neighbour CSR   int32                the reference's per-wall neighbour lists of every leaf, in the reference's
//                                        order (cold: only read when a position is not strictly inside a candidate)
//   Cartesian       double xv/yv/zv (staged in LDS), density double[num_cells]
//   dust tables     lambda_border/sigma_ext/sigma_sca/asymmpar double[num_lambda]: the borders are staged in LDS by the launch
//                                        kernel (the binary search of a history's wavelength); the properties of the bin found
//                                        travel with the history in its slot
// HBM, read-write:
//   packet slots    struct-of-arrays over `num_slots` concurrently live photon histories (see SlotArrays)
//   frames          double[frame_size]   detector arrays, accumulated with f64 atomics
#ifndef PMC_DEVICE_H
#define PMC_DEVICE_H

#include "../../include/pmc.h"
#include <stdint.h>

#define PMC_MAX_INSTRUMENTS 16  // (the observers with a peel-off packet of a cycle are flagged in sixteen bits of the slot's mode word;
                                // the per-instrument slot arrays are sized by the scene's own instrument count)
#define PMC_MAX_CONTEXTS 6  // scene slots in constant memory (live contexts per process and device; 6 x sizeof(DevScene) < 64 KB)
#define PMC_MAX_LEVEL 20     // (box codes hold three 20-bit indices of the coordinate table and the size exponent; the packed indices of a walk 21 bits
                             // per axis; link words and task records five bits of size exponent.  Rounds 1-5: 15.  TreePolicy allows maxLevel up to 99:
                             // a tree of 2^20 finest cells per axis resolves 0.04 pc in a box of 40 kpc)
#define PMC_STAT_CAP 48     // entries of a slot's own contribution list per instrument: DISTINCT pixels a history contributes to
                            // (FluxRecorder statistics); a multiple of 4.  A history with more distinct pixels continues its list in
                            // chained blocks of PMC_STAT_CAP entries from the slot group's pool (DevScene::stat_pool_*): the
                            // reference's list is unbounded (FluxRecorder.hpp:327-338)
#define PMC_STAT_POOL_EXHAUSTED 0x40000000  // flag in the length word of a list head: a contribution was lost because the pool had no block left

// link word (uint32): bits 0-4 size exponent e of the target box (its edge spans 2^e finest cells), bits 5-29 index,
// bits 30-31 kind: 0 leaf cell (device index), PMC_LINK_NODE internal node (NodeRec index), PMC_LINK_OCTET internal node
// whose eight children are all leaves (index = device index of child 0; the children are consecutive cells in child order,
// e = exponent of the NODE); PMC_LINK_NONE: outside the grid (test it first: its bit 30 is set)
#define PMC_LINK_NONE 0x7FFFFFFFu
#define PMC_LINK_NODE 0x80000000u
#define PMC_LINK_OCTET 0x40000000u
#define PMC_LINK_EXP_BITS 5
#define PMC_LINK_EXP_MASK 31u
#define PMC_LINK_INDEX_MASK 0x1FFFFFFu
#define PMC_LINK_MAX_INDEX ((1u << 25) - 2u)

// Octree cells.  A walk step leaves its cell through ONE wall and needs the cell's density and the link through that
// wall: the step gathers the cell's HOT record CellRec (32 bytes: two 16-byte loads from one sector, issued as soon as the
// cell is known -- before the arithmetic of the step that enters it -- so that the memory round trip of a step overlaps
// with the arithmetic of the previous one).  Cells are numbered in depth-first order of the tree: siblings are consecutive
// (four records per 128-byte line), and spatially close cells of any level are close in the table.
// A link names the leaf that covers the whole wall (same size or coarser); or, when finer neighbours share the wall, the
// same-size internal node: if its children are all leaves (PMC_LINK_OCTET) the child is picked by the index bits of the
// position without any load, otherwise (PMC_LINK_NODE) the walk descends with one 4-byte gather per level.
// LeafRec is the COLD per-cell record (start of a walk, undecided steps).
struct CellRec
{
    double   density;   // number density n[m]
    uint32_t link[6];   // through wall 2 * axis + side (side 0: lower wall, 1: upper wall)
};
static_assert(sizeof(CellRec) == 32, "CellRec must be 32 bytes");

struct LeafRec
{
    uint64_t code;      // box code: bits 0-19 / 20-39 / 40-59 = fine lower-corner indices fx / fy / fz: the entries of the lower walls in
                        // the coordinate table [3][2^Lmax+1] of their axis; bits 60-63 = size exponent e = Lmax - level (the box spans
                        // 2^e finest cells per axis), or 15 for e >= 15 with e - 15 in the low five bits of fx (zero in such a box);
                        // pmc_walk.inc decodeBoxWords
    double   density;   // number density n[m]
};
static_assert(sizeof(LeafRec) == 16, "LeafRec must be 16 bytes");

struct NodeRec
{
    uint64_t code;
    uint32_t child[8];
    int32_t  pad[6];
};
static_assert(sizeof(NodeRec) == 64, "NodeRec must be one 64-byte record");

struct DevInstrument
{
    double kx, ky, kz;
    double ikx, iky, ikz;   // RN(1/k) of the observer direction, NaN for an ignored axis (|k| <= 1e-15): the peel-off walks
    uint32_t sgn;           // bit a: k_a < 0
    double costheta, sintheta, cosphi, sinphi, cosomega, sinomega;
    double xpmin, xpsiz, ypmin, ypsiz;
    int32_t nxp, nyp;
    int32_t same_observer;
    int32_t include_sed, include_ifu, record_components, num_levels, record_stats;
    double aperture_r2;  // SEDInstrument aperture radius squared (0: none)
    double zp1;          // 1 + redshift of the observer frame: the wavelength bins are looked up at lambda (1 + z) (FluxRecorder.cpp:309-310)
    int32_t num_lambda, num_border;
    int32_t mono_ell;       // DevScene::mono: the wavelength bin of every history in this instrument (-1: outside its grid)
    const double* border;   // device
    const int32_t* ellv;    // device
    // frame layout (doubles)
    int64_t sed_offset, ifu_offset, wsed_offset, wifu_offset, npix;
    int32_t num_components;
    int32_t sed_lds_offset;  // offset (doubles) of this instrument's privatised SED block in LDS: [comp][ell] then [5][ell]
    int64_t stat_acc_offset; // first record of this instrument in DevScene::stat_acc (records of 8 doubles per bin ell * npix + pixel)
};

// One slot = one live photon history.  The walk kernel reads the task fields and writes the result fields; the
// transition kernel owns everything else.  Struct-of-arrays: lane i of a wave touches element slot_i of each array.
struct SlotArrays
{
    // packet (PhotonPacket.hpp:333-363)
    double* rx; double* ry; double* rz;     // position
    double* kx; double* ky; double* kz;     // propagation direction
    double* lambda;                         // wavelength
    double* W;                              // weight (luminosity = W / lambda)
    double* Lthreshold;
    double* ppW;                            // [num_instruments][num_slots] weight of the cycle's peel-off packet towards
                                            // the observer whose instrument group starts at that instrument
    double* taupath;                        // optical depth of the whole path (pass 1), kept for the escape weight
    double* tausample;                      // interaction optical depth sampled between the passes (path-length-bias weight)
    uint64_t* history;
    double*  rngSpare;
    uint32_t* rngBlock;                     // (block << 1) | have
    double* dustExt; double* dustSca;       // extinction and scattering cross section of the dust mix at the history's wavelength
    double* dustAbs;                        // (explicit-absorption cycle only, else null) its absorption cross section
    int32_t* dustIdx;                       // (several medium components, panchromatic: else null) [num_media][num_slots] index of the
                                            // history's wavelength in every component's dust tables
    double* dustAsym;                       // and its asymmetry parameter: DustMix::indexForLambda(lambda) is looked up ONCE, at launch
    int32_t* mode;                          // bit 5 alive, bits 8-23 observers (group leaders) with a peel-off packet this cycle
    int32_t* nscatt;
    int32_t* ell;                           // [num_instruments][num_slots] wavelength bin per instrument
    int32_t* statHead;                      // [num_instruments][num_slots] 64-byte head record of the history's contribution list:
                                            // {int32 bin[4]; double w[4]; int32 n; int32 next; 8 unused}: the first four entries, the
                                            // length of the whole list (head + own entries + chained blocks) and the first chained
                                            // block (-1: none).  A history touches 3.7 distinct pixels on average: one sector, loaded
                                            // with the rest of the slot's state, decides most contributions without a further round trip
    int32_t* rfell;                         // wavelength bin in the radiation field grid, -1 outside (only if rf_store)
    // walk results
    double* ptau;                           // [num_instruments][num_slots] optical depth towards that observer (inf: the
                                            // contribution is zero)
    double* sint;                           // propagation walk: interaction distance
    double* nint;                           //         density of the interaction cell
    int32_t* mint;                          //         interaction cell (-1: no interaction, the history ends); after the scattering
                                            //         there it is the leaf that contains the slot's position: the hint for the first cell
                                            //         of the next cycle's walks (the launch kernel stores -1: no hint)
    // per-history contribution lists, entries 4 .. PMC_STAT_CAP - 1: bin[(inst * num_slots + slot) * PMC_STAT_CAP + e], w likewise
    int32_t* statBin;
    double*  statW;
    int64_t  num_slots;
};

// Walk tasks: the transition and launch kernels start every walk of a cycle (PathSegmentGenerator::moveInside, location
// of the first cell, first exit distance) and leave its start state in a task record of the slot: record 0 = the
// propagation walk, record 1 + g = the peel-off walk towards the observer whose instrument group starts at g; index =
// record * num_slots + slot; bits == PMC_TASK_NONE: no walk.  The walk kernel runs the records of a slot one after the
// other in one lane and touches the slot arrays only to write the results.
struct TaskArrays
{
    double* rx; double* ry; double* rz;     // position inside the grid (after moveInside)
    double* kx; double* ky; double* kz;     // direction
    double* s0;                             // length of the initial segment outside the grid (0 if none)
    double* ds;                             // exit distance of the first cell
    double* target;                         // the walk stops in the first segment with tau > target
    int32_t* cell;                          // first cell
    uint32_t* bits;                         // mode (bits 0-1) | exit axis (2-3) | direction signs (4-6) | size exponent (8-11)
                                            // (octree peel-off records hold position, ds, target, sext, cell, pidx, bits only:
                                            // their direction is the observer's)
    int32_t* cijk;                          // Voronoi: the neighbour (or wall) through which the path leaves the first cell
    uint64_t* pidx;                         // octree: packed fine lower-corner indices of the first cell (Walk::P)
    int32_t* liveList;                      // [2][num_slots] sparse generations (the end of a segment): the live slots of a slot group,
                                            // compacted by the cycle start kernel into the group's own range of one half of this array;
                                            // the kernels of the next generation run over the list instead of the group's slot range
                                            // (and write the list of the generation after it into the other half)
    uint32_t* endedCount;                   // [num_slots / 64 + pad] per wave tile (64 consecutive slots): the histories that
                                            // ended in the tile this generation (transition kernel); endedScanKernel turns the
                                            // counts of a slot group into their exclusive prefix, from which the launch kernel
                                            // derives the index of the history each of those slots takes up next
};
#define PMC_TASK_NONE 0xFFFFFFFFu
#define PMC_TASK_EXP_MASK 31u   // octree: bits 8-12 of TaskArrays::bits: size exponent of the first cell
#define PMC_TASK_MOVED 0x2000u  // octree: bit 13 of TaskArrays::bits: PathSegmentGenerator::moveInside has moved the start of the walk (the
                                // position was outside the grid): position and initial path length are in the record; otherwise the
                                // walk starts at the slot's position (SlotArrays::rx ...) with no initial length, and the record holds
                                // neither (nor, ever, the direction of a propagation walk: it is the slot's)

#define PMC_MAX_SOURCES 16
// one source of the source system: spatial sampling, luminosity per packet, wavelength sampling (pmc.h pmc_source)
struct DevSource
{
    int32_t source_kind;
    double  src_pos[3];
    double  reff;
    int32_t sersic_n;
    const double* sersic_s;
    const double* sersic_M;
    double  src_box[6];
    double  packet_luminosity;
    int32_t lambda_mode;
    int32_t num_oligo;
    const double* oligo_lambda;
    const double* oligo_weight;
    double  lambda_bias;
    int32_t num_sed;
    const double* sed_lambda;
    const double* sed_p;
    const double* sed_P;
    int32_t bias_kind;
    double  bias_min, bias_max;
    int32_t sed_kind;
    double  sed_f1, sed_f2, sed_ltot;
    int32_t angular_kind;        // PMC_ANGULAR_*: emission direction of a point source
    double  angular_axis[3];
    double  angular_cos_delta;
};

// one medium component of a system of several (pmc.h pmc_scene::media): its cell densities in the DEVICE numbering of the cells and its
// dust tables; mono_*: its properties at the one wavelength of an oligochromatic scene (DevScene::mono)
struct DevMedium
{
    const double* density;
    const double* lambda_border;
    const double* sigma_ext;
    const double* sigma_sca;
    const double* sigma_abs;
    const double* asymmpar;
    int32_t num_lambda;
    double mono_ext, mono_sca, mono_abs, mono_asym;
};

struct DevScene
{
    // ---- grid
    int32_t grid_kind;
    double gx0, gy0, gz0, gx1, gy1, gz1, eps;
    int32_t nx, ny, nz;
    const double* xv;
    const double* yv;
    const double* zv;
    const double* cell_density;  // Cartesian
    int32_t lmax;                // octree: finest level; table entries per axis = (1 << lmax) + 1
    double  fine_scale[3];       // octree: 2^lmax / extent per axis (position -> finest-level cell index)
    uint32_t tab_stride_bytes;   // octree: bytes per axis of the coordinate table = 8 * ((1 << lmax) + 1)
    const double* coord_tab;     // [3][(1<<lmax)+1]
    int32_t tab_in_lds;          // octree: the walk kernels stage the coordinate table in LDS (lmax <= 12: up to 98 KB); deeper octrees
                                 // read it from global memory
    int32_t coarse_level;        // octree: level Lc = min(lmax, 6) of the top-down search table
    const uint32_t* coarse_tab;  // [2^Lc][2^Lc][2^Lc] (z, y, x): link of the node at level <= Lc that covers the coarse cell
    const LeafRec* leaves;
    const CellRec* cell_tab;     // [cell_slots]: the walk step's gather
    const NodeRec* nodes;
    uint32_t root_link;
    const int32_t* nbr_start;    // [6*num_cells + 1]
    const int32_t* nbr_list;     // leaf cell indices
    int32_t num_cells;
    // octree: the device tables number the cells in their own order (pmc_api.hip buildTree: depth-first order of the
    // tree); cell_ext[device index] = the caller's cell index m, cell_slots = entries per device table
    const int32_t* cell_ext;
    int32_t cell_slots;
    // Voronoi (pmc_grid::site ...): site positions as double4-aligned records {x, y, z, number density}, so that one
    // 32-byte gather brings everything the traversal needs of a neighbour or of the cell itself
    const double* vsite;          // [num_cells][4]
    const int32_t* vnbr_start;    // [num_cells + 1]
    const int32_t* vnbr_list;
    const double* vpair;          // [vnbr_start[num_cells]][4]: per (cell, neighbour) entry the neighbour's site x, y, z and its
                                  // index (as the bit pattern of an int64), in list order: the hot loop reads two 16-byte
                                  // words per neighbour from consecutive addresses instead of index -> site gathers
    const double* vhead;          // [num_cells][8]: per cell ONE 64-byte record {site x, y, z, number density, list start | list end
                                  // (two int32 in one double), 3 unused}: what a walk reads of the cell it enters, in one sector
    // per observer (up to PMC_SORT_OBS; vobs_of_inst[instrument] = its table or -1): what a peel-off walk towards that observer reads of a cell, in ONE
    // run of 64-byte units: a header {site x, y, z, number density, number of entries (int32)} and behind it the neighbour entries that can be the exit
    // for the observer's direction (n . k_obs > 0, decided at setup by the walk's own test; domain walls as the cone's mask keeps them) in list order,
    // 32 bytes each: {site x, y, z, tag = neighbour index (low 32 bits) | unit at which the neighbour's run starts (high 32 bits)}, stored in groups of
    // PMC_VORO_RUN_LANES entries as {x, y} of every entry of the group, then {z, tag} of every entry (the last group filled up with tag -7): the
    // PMC_VORO_RUN_LANES lanes that share a walk in voroPeelKernel read a group with two loads of adjacent 16-byte words
    const double* vgen_run;         // (or null) ALL neighbours of a cell as a run of the same form: what a propagation walk in voroPropKernel reads
    const uint32_t* vgen_start;     // [num_cells] first unit of the cell's run in vgen_run
    // per main direction cone (48: sign octant x order of |k_x|, |k_y|, |k_z|; or null) the neighbours that the masks of its four sub-cones do not all
    // cull, as a table of runs: a propagation walk keeps its cone, so voroPropKernel reads 65 % of the entries -- fewer lines of the 49 MB that
    // miss L2 -- without a mask (1.7 GB at 10^5 sites: the tables are an acceleration, left out where device memory is short)
    const double* vcone_run[48];
    const uint32_t* vcone_start[48];
    const double* vobs_run[4];
    const uint32_t* vobs_start[4];  // [num_cells] first unit of the cell's run (a walk's first visit)
    int8_t vobs_of_inst[16];
    const uint32_t* vcull;        // [PMC_VORO_CONES][num_cells] (cone-major: a walk keeps its cone, and all peel-off walks towards an observer share
                                  // one: 4 bytes per cell of a 400 KB slice instead of one line per visit): per cell and direction cone (sign pattern of k x order of |k_x|, |k_y|,
                                  // |k_z|) bit j set: the j-th neighbour of the list (j < 32) lies behind every direction of the
                                  // cone (n . k < 0 with a margin far above rounding), or is a domain wall the cone moves away
                                  // from: the reference skips it (ndotk > 0 fails), and the walk does not even read it
    int32_t vblock_n;
    const int32_t* vblock_start;  // [vblock_n^3 + 1]
    const int32_t* vblock_list;
    // ---- medium
    int32_t num_lambda;
    const double* lambda_border;
    const double* sigma_ext;
    const double* sigma_sca;
    const double* sigma_abs;     // (explicit-absorption cycle only, else null)
    const double* asymmpar;
    // every source emits at ONE wavelength, the same one (oligochromatic with a single wavelength: BASELINE configs[0] and [1]): the
    // wavelength of a history and the dust mix's properties at it are constants of the scene, and no kernel stores or loads them per
    // slot (SlotArrays::lambda, dustExt, dustSca, dustAsym stay unused)
    int32_t mono;
    double mono_lambda, mono_ext, mono_sca, mono_asym, mono_abs;
    // ---- options
    int32_t force_scattering;
    int32_t voro_prop_checkpoints;  // (Voronoi) voroPropKernel keeps two pass-1 walker states per walk in LDS for pass 2 (off: tuning switch PMC_PROP_NO_CHECKPOINTS)
    int32_t voro_defer_scan;      // (Voronoi) the cycle start kernel locates the first cell of a walk and leaves its exit to the walk kernels, whose first step
                                  // scans it like every other cell (TaskArrays::cijk = PMC_VORO_FIRST_SCAN)
    int32_t explicit_absorption;  // PhotonPacketOptions::explicitAbsorption (pmc.h pmc_options)
    // ---- several medium components (num_media > 1; else the members above describe the only one): MediumSystem.cpp:874-887, 678-730,
    // 796-817.  Component 0 is ALSO what the members above describe (the hot cell records carry its density).
    int32_t num_media;
    DevMedium med[PMC_MAX_MEDIA];
    double  min_weight_reduction;
    int32_t min_scatt_events;
    double  path_length_bias;
    // ---- sources (SourceSystem.cpp:100-107: history index h belongs to source i with src_first[i] <= h < src_first[i+1])
    int32_t num_sources;
    DevSource src[PMC_MAX_SOURCES];
    const double* netzer_cos;    // NetzerAngularDistribution: cos theta grid and cumulative distribution (PMC_NETZER_POINTS + 1 each), or null
    const double* netzer_X;
    uint64_t src_first[PMC_MAX_SOURCES + 1];
    // ---- instruments
    int32_t num_instruments;
    DevInstrument inst[PMC_MAX_INSTRUMENTS];
    int32_t any_stats;
    // ---- radiation field (MonteCarloSimulation::storeRadiationField): rf[m * rf_num_lambda + ell]
    int32_t rf_store, rf_num_lambda, rf_num_border;
    const double* rf_border;
    const int32_t* rf_ellv;
    double* rf;
    // ---- outputs and work state
    double* frames;
    // statistics accumulator: per instrument with statistics and per bin q = ell * npix + pixel ONE 64-byte record
    // {sum w^0 .. sum w^4, 3 unused}, so that the five sums of a history's contribution to a bin go out with one atomic
    // instruction whose lanes share a sector; merged into the wifu arrays of `frames` (FluxRecorder's layout) and cleared
    // at the end of every segment (statMergeKernel)
    double* stat_acc;
    int64_t stat_acc_records;
    // continuation blocks of the per-history contribution lists (instruments with statistics): block b holds PMC_STAT_CAP
    // entries (stat_pool_bin / stat_pool_w [b * PMC_STAT_CAP + q]) and the id of the block that follows it (stat_pool_next[b],
    // -1: none).  Every slot group owns the ids [stat_pool_first[g], stat_pool_first[g] + stat_pool_count[g]) and keeps its free
    // ids as a stack in stat_pool_free over the same index range, filled up to counters[PMC_CTR_STATFREE(g)]: the transition
    // kernel of a group only takes blocks, its launch kernel only returns them, and the two never overlap (one stream).
    int32_t* stat_pool_bin;
    double*  stat_pool_w;
    int32_t* stat_pool_next;
    int32_t* stat_pool_free;
    int32_t  stat_pool_first[4];
    int32_t  stat_pool_count[4];
    unsigned long long* counters;  // [PMC_NUM_COUNTERS]: [0..6] pmc_counter_values, [7] internal errors (pmc_run_primary fails), [8] next history offset,
                                   // [16..21] walk work, [32 + 4 g ..] work counters of slot group g, [120 + g] history bases,
                                   // [128 + 16 g ..] task cursors (PMC_CTR_*), [192..239] section timers of profiling builds
    SlotArrays slots;
    TaskArrays tasks;
    // ---- LDS carve-up (in doubles from the start of dynamic LDS; no kernel has static LDS, so that the octree
    //      coordinate table sits at LDS address 0 in all of them)
    int32_t lds_grid_len;          // all kernels: grid tables at offset 0
    //      transition AND launch kernels: privatised SED blocks, hot-bin table, integer scratch (lds_total_transition doubles);
    //      launch kernel only, behind them: Sersic tables of a single source, wavelength borders of the dust mix
    //      (lds_total_launch doubles)
    int32_t lds_dust_off;
    int32_t lds_src_off, lds_sed_off, lds_sed_len, lds_hot_off, lds_sort_off, lds_total_transition, lds_total_launch, lds_total_walk;
    int32_t dust_in_lds;           // the launch kernel stages the index borders of the dust mix (DustMix::_lambdav) in LDS
};

// The radiation-field contributions of the propagation walks of one slot group and generation (octree): (key, value)
// pairs, key = m * num_lambda + ell, appended by the walk kernel in chunks of PMC_RF_LOG_CHUNK entries that a wave claims
// from the cursor; partitioned by key range and added to the table after the generation (pmc_api.hip rfFlush)
struct RfLogArgs
{
    uint32_t* keys;
    double* vals;
    unsigned long long cap;  // entries; 0: no log, every contribution is an atomic add into the table
    int32_t cursor;          // index of the log's cursor in DevScene::counters
    uint32_t padKey;         // key of the entries that fill up a wave's last chunk (beyond every table index)
};
// Statistics (FluxRecorder::recordContributions, FluxRecorder.cpp:962-1014): when a history ends, every entry of its contribution list --
// (bin, summed weight w) -- adds w^0 .. w^4 to the bin's record of DevScene::stat_acc.  As atomics these are one sector request per
// entry (3.7 per history on configs[1]: 33 ms of the launch kernel per 1e8 packets at the chip's 2.4e10 sector atomics/s); instead
// the launch kernel (and the transition kernel of a sparse generation) appends the entries to a log per slot group -- key = record
// index, value = w -- in chunks of PMC_RF_LOG_CHUNK entries that a WAVE claims from the log's cursor and keeps over the generations
// (its place in its open chunk is saved when the kernel ends, by wave index, and taken up again by the next launch: a claim per
// entry, or per wave and call, is a same-address atomic -- measured: the launch kernel twice as slow); chunkFill[chunk] tells the
// flush how many entries of a chunk are there.  The log is partitioned by key range (the counting sort of the radiation-field log,
// pmc_walk_tree.inc) and summed per bin in LDS (statReduceKernel) at the end of the segment, or when it is half full: 28 bytes of
// streaming traffic per entry instead of a sector atomic, and no kernel more in the chain of a generation (every small kernel there
// waits for a free CU next to the other groups' walk kernels: flushed every few generations the log was SLOWER than the atomics).
struct StatLogArgs
{
    uint32_t* keys;
    double* vals;
    unsigned long long cap;        // entries; 0: no log, the sums are added atomically
    int cursor;                    // index of the log's cursor in DevScene::counters
    unsigned long long* waveBase;  // [PMC_STAT_LOG_WAVES] first entry of the open chunk of wave w of the kernels' grids
    uint32_t* waveFill;            // [PMC_STAT_LOG_WAVES] entries of it in use (PMC_STAT_NO_CHUNK: the wave holds no chunk)
    uint32_t* chunkFill;           // [cap / PMC_RF_LOG_CHUNK] entries of every chunk (PMC_RF_LOG_CHUNK unless a wave has left it open or short)
};
#define PMC_STAT_NO_CHUNK 0xFFFFFFFFu
#define PMC_STAT_LOG_WAVES 16384  // waves of a launch / transition kernel grid that keep a chunk (1024 workgroups of four waves; beyond: atomics)
#define PMC_STAT_BUCKET_BITS 11  // records per partition of the statistics log: 2^11 x five sums = 80 KB of LDS in statReduceKernel

// Octree, up to PMC_SORT_OBS observers: the peel-off walks of a generation towards an observer as 64-byte records SORTED by the tile of the detector plane their start
// position projects to: a counting sort on the tile in which the cycle start kernel is the scatter pass.  peelSortCountKernel counts the
// slots that will have a walk per tile (from the slot's mode word and position), workgroup by workgroup; the cycle start kernel -- same
// workgroups, same slots -- writes every walk's start state straight to its place in tile order (one LDS atomic away).  The peel-off
// kernel takes the records in that order: its task loads are coalesced, and the walks in flight on the chip at any moment run through
// one slab of the grid (parallel lines of sight from neighbouring tiles) -- the L2s hold it.
struct PeelRec
{
    double rx, ry, rz;   // start of the walk (inside the grid)
    double ds;           // exit distance of the first cell
    double target;       // the walk stops in the first segment with tau > target
    uint64_t pidx;       // packed fine lower-corner indices of the first cell
    uint32_t cell, bits; // first cell; exit axis (bits 2-3), size exponent (8-11) as in TaskArrays::bits
    int32_t slot;        // where the optical depth goes (SlotArrays::ptau)
    uint32_t pad;
};
#define PMC_SORT_OBS 4  // observers whose peel-off walks are sorted (scenes with more observers: task arrays, slot order)
struct PeelSortArgs  // sort-count kernel and cycle start kernel; numObs == 0: no sort, the walks' start states go to TaskArrays
{
    int32_t numObs;                             // sorted observers
    int32_t propIndex;                          // (Cartesian, Voronoi) index into the arrays below of the list of the PROPAGATION walks, sorted by the sign
                                                // octant of their direction (partition = octant), or -1: the propagation walks run in slot order
    int32_t obs[PMC_SORT_OBS];                  // their instruments (the first of each observer group)
    int8_t sortIndex[16];                       // instrument -> index into the arrays below, or -1
    PeelRec* out[PMC_SORT_OBS];                 // (octree) the group's records in tile order, per observer
    int32_t* listOut[PMC_SORT_OBS];             // (Cartesian, Voronoi) the group's slots in tile order, per observer: the walks' start states stay in TaskArrays
    uint32_t* matrix[PMC_SORT_OBS];             // [workgroups][numParts]: entries of workgroup b's sort tiles per partition, then their prefix over the workgroups
    const unsigned long long* start[PMC_SORT_OBS];  // [numParts + 1] first record of every partition (start[numParts] = number of records)
    uint32_t numParts;                          // PMC_PEEL_TILES^2
    uint32_t cap;                               // records (list entries) allocated per observer: a place beyond it is refused and counted (counters[7])
    int32_t ldsOffset;                          // cycle start kernel: where its cursors live in LDS (behind the grid tables): numObs x numParts
    double centre[3];                           // of the grid
    double scale;                               // PMC_PEEL_TILES / the grid's diagonal
};
// Cartesian / Voronoi walk kernel: the walks of a generation as ONE stream of tasks -- the propagation walks of the group's slots in slot
// order (record 0), then for every sorted observer its peel-off walks in tile order (lists of slots written by the cycle start kernel).
// numLists == 0: a task is a slot with all its walks, in slot order (scenes with more than PMC_SORT_OBS observers)
struct WalkStreamArgs
{
    int32_t numLists;
    int32_t rec[PMC_SORT_OBS];                      // task record of the walks of list k (1 + instrument)
    const int32_t* list[PMC_SORT_OBS];
    const unsigned long long* count[PMC_SORT_OBS];  // entries of list k (device memory)
    unsigned long long* xcdCursor;                  // [8] (or null) one cursor per eighth of the stream: an XCD's workgroups take their own eighth first
    const int32_t* propList;                        // (or null) the slots with a propagation walk, sorted by the sign octant of their direction: with the eighths
    const unsigned long long* propCount;            // of the XCD affinity an XCD's L2 sees the walks of about one octant (its entries: device memory)
};
struct PeelSortedArgs  // peel-off kernel; rec == nullptr: task records from TaskArrays
{
    const PeelRec* rec;
    const unsigned long long* count;  // number of sorted records (device memory)
    unsigned long long* xcdCursor;    // [8] one cursor per eighth of the records: the workgroups of an XCD take the walks of their own eighth
                                      // of the tile order first (then the others'), so that the L2 of an XCD sees an eighth of the slab the
                                      // walks in flight run through
};
#ifndef PMC_PEEL_TILES
#define PMC_PEEL_TILES 32  // tiles per axis of the detector plane (PMC_PEEL_TILES^2 sort partitions)
#endif
// a link to a run of a table of runs (an entry's tag, high 32 bits; DevScene::v*_start): first unit of the run | number of its entries << 27 (31: more than
// 30, the header knows) -- a walk requests exactly the groups the next cell has, together with its header
#define PMC_VORO_RUN_UNIT_BITS 27
#define PMC_VORO_RUN_UNIT_MASK 0x07FFFFFFu
#define PMC_VORO_RUN_COUNT_UNKNOWN 31u
#define PMC_VORO_FIRST_SCAN (-1000)  // TaskArrays::cijk of a walk whose first cell has not been scanned for its exit yet
#define PMC_VORO_RUN_PAD 16   // units of zeros behind the last run of DevScene::vobs_run
#ifndef PMC_VORO_RUN_LANES
#define PMC_VORO_RUN_LANES 2  // lanes that share a walk in voroPeelKernel = entries per group of a run (2 or 4)
#endif
#ifndef PMC_VORO_CONES
#define PMC_VORO_CONES 192  // direction cones of DevScene::vcull: 48, or 192 (every cone divided at the midpoints of its edges)
#endif
#define PMC_RF_LOG_CHUNK 4096
#define PMC_RF_BUCKET_BITS 13  // keys per partition of the log: 2^13 doubles = 64 KB of LDS in rfReduceKernel

#define PMC_NUM_COUNTERS 512
#define PMC_CTR_HISTORY 8
// [16..21] work of the octree walk kernels: peel-off wave-steps, lane-steps, service rounds; propagation likewise
#define PMC_CTR_WALKWORK 16
// per slot group g: live slots; cursors of the walk kernels over the slots of the group (k = 0: the generic / propagation
// kernel, k = 1 + observer: the peel-off kernel of that observer)
#define PMC_CTR_LIVE(g) (35 + 4 * (g))
#define PMC_CTR_TASK(g, k) (256 + 32 * (g) + (k))  // k = 0: propagation walks, 1 + i: peel-off walks towards instrument i (< 16)
#define PMC_CTR_TASKS_PER_GROUP 32
#define PMC_CTR_LIST(g) PMC_CTR_TASK(g, 30)        // entries of the group's list of live slots (TaskArrays::liveList)
#define PMC_CTR_RFLOG(g) PMC_CTR_TASK(g, 31)       // entries of the group's radiation-field log claimed so far
#define PMC_CTR_STATLOG(g) (56 + (g))              // entries of the group's statistics log claimed so far (kept over the generations: the log is
                                                   // flushed when it has filled up, pmc_api.hip)
// per slot group g: offset of the first history the group's ended slots take up this generation (endedScanKernel)
#define PMC_CTR_HBASE(g) (120 + (g))
// ... and the first history index the group may not take (the staggered end of a segment: endedScanKernel)
#define PMC_CTR_HLIMIT(g) (124 + (g))
#define PMC_MAX_GROUPS 4
// per slot group g: free blocks of the group's share of the statistics pool (DevScene::stat_pool_free)
#define PMC_CTR_STATFREE(g) (48 + (g))
#define PMC_TRANSITION_ALIGN 1024  // slot groups start at multiples of the transition kernel's workgroup size

#endif
