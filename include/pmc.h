/* pmc.h -- C ABI of the MI355X photon-packet Monte Carlo engine ("pmc" = primary Monte Carlo).
 *
 * Drop-in boundary for SKIRT 9's primary-emission photon loop.  The reference has no FFI: the loop sits
 * behind C++ virtuals in one process.  This header defines the seam a SKIRT maintainer would bind to; each
 * entry point cites the reference code whose work it takes over (paths relative to the SKIRT 9 tree):
 *
 *   pmc_create        <- the read-only state the loop consumes after Simulation::setupSimulation():
 *                        SpatialGrid (SKIRT/core/TreeSpatialGrid.cpp:23-78, CartesianSpatialGrid.cpp:14-28),
 *                        MediumState number densities (MediumSystem.cpp:292-390), DustMix tables
 *                        (DustMix.cpp:47-162), Configuration photon options (Configuration.cpp:105-109),
 *                        SourceSystem launch data (SourceSystem.cpp:75-97), instruments
 *                        (FrameInstrument.cpp:12-33) and FluxRecorder array shapes (FluxRecorder.cpp:185-300)
 *   pmc_run_primary   <- MonteCarloSimulation::runPrimaryEmission's parallel->call(Npp, performLifeCycle)
 *                        (MonteCarloSimulation.cpp:126-129, 538-613) + instrumentSystem()->flush() (:129)
 *   pmc_download      <- the detector arrays FluxRecorder::calibrateAndWrite reads (FluxRecorder.cpp:484-493)
 *   pmc_frames_device <- same arrays, as a device pointer, so that the caller can run the counterpart of
 *                        ProcessManager::sumToRoot (SKIRT/mpi/ProcessManager.cpp:223-255) as ONE RCCL reduce
 *   pmc_trace_ray     <- PathSegmentGenerator::start()/next() (TreeSpatialGrid.cpp:132-217,
 *                        CartesianSpatialGrid.cpp:87-163): the (m, ds) sequence of one ray, for the bit-exact check
 *   pmc_download_radiation_field <- the primary radiation field table MediumSystem::_rf1 that
 *                        MonteCarloSimulation::storeRadiationField fills (MonteCarloSimulation.cpp:638-692,
 *                        MediumSystem.cpp:1294-1300) and MediumSystem::meanIntensity reads (:1370-1380);
 *   pmc_radiation_field_device <- same table as a device pointer, for the counterpart of
 *                        MediumSystem::communicateRadiationField (MediumSystem.cpp:1304-1313: sumToAll) as one RCCL all-reduce
 *   pmc_sampler_*     <- ParticleSnapshot::density(Position) evaluated for the sample positions of the setup phase
 *                        (ParticleSnapshot.cpp:233-243; DensityTreePolicy.cpp:141, MediumSystem.cpp:91-96)
 *   pmc_counters      <- no reference counterpart: counted cell visits / detector updates for the roofline
 *
 * Conventions: plain C, no exceptions cross the boundary; every function returns PMC_OK (0) or a negative
 * status, and pmc_last_error() returns a human-readable message for the calling thread.  All input tables are
 * host pointers owned by the caller; pmc_create copies what it needs to the device before returning.  One
 * context per device; calls on one context must be serialised by the caller (the reference forbids recursive
 * Parallel::call, SKIRT/core/Parallel.hpp).  All physical quantities are SI doubles, indices are int32.
 */
#ifndef PMC_H
#define PMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMC_ABI_VERSION 9

enum { PMC_OK = 0, PMC_ERR_INVALID = -1, PMC_ERR_UNSUPPORTED = -2, PMC_ERR_DEVICE = -3, PMC_ERR_NOMEM = -4,
       PMC_ERR_OVERFLOW = -5 /* the pool of statistics-list blocks ran out and could not grow (see pmc_run_primary) */ };

/* ---------------------------------------------------------------- spatial grid ---- */

enum { PMC_GRID_CARTESIAN = 1, PMC_GRID_OCTREE = 2, PMC_GRID_VORONOI = 3 };

/* walls in the reference's order (SKIRT/core/TreeNode.hpp enum Wall): BACK=-x, FRONT=+x, LEFT=-y, RIGHT=+y,
   BOTTOM=-z, TOP=+z */
enum { PMC_WALL_BACK = 0, PMC_WALL_FRONT, PMC_WALL_LEFT, PMC_WALL_RIGHT, PMC_WALL_BOTTOM, PMC_WALL_TOP };

typedef struct pmc_grid
{
    int32_t kind;                 /* PMC_GRID_* */
    double  xmin, ymin, zmin;     /* extent of the spatial domain (BoxSpatialGrid) */
    double  xmax, ymax, zmax;
    double  eps;                  /* 1e-12 * diagonal (TreeSpatialGrid.cpp:28, CartesianSpatialGrid.cpp:95) */
    int32_t num_cells;

    /* --- Cartesian (CartesianSpatialGrid.cpp:19-24): border arrays, m = k + nz*j + nz*ny*i */
    int32_t nx, ny, nz;
    const double* xv;             /* nx+1 */
    const double* yv;             /* ny+1 */
    const double* zv;             /* nz+1 */

    /* --- Octree: the reference's node list in node-id order (TreeSpatialGrid.cpp:38-49), flattened.
       node_box[6*id + {0..5}] = xmin,ymin,zmin,xmax,ymax,zmax exactly as the reference's TreeNode holds them;
       node_first_child[id] = id of child 0 (children are 8 consecutive ids, OctTreeNode.cpp:22-33) or -1;
       node_cell[id] = cell index m of a leaf or -1 (TreeSpatialGrid::_cellindexv);
       neighbours of node id through wall w, in the reference's (sorted) list order (TreeNode.cpp:200-207):
       nbr_list[ nbr_start[6*id+w] .. nbr_start[6*id+w+1] )  -- a CSR over 6*num_nodes rows. */
    int32_t        num_nodes;
    const double*  node_box;
    const int32_t* node_level;
    const int32_t* node_first_child;
    const int32_t* node_cell;
    const int32_t* nbr_start;     /* 6*num_nodes + 1 */
    const int32_t* nbr_list;

    /* --- Voronoi (VoronoiMeshSnapshot.cpp:1058-1188): cell m = Voronoi cell of site m inside the domain box.
       site[3*m + {0,1,2}] = site position (the reference's order: sorted by x, :509);
       neighbours of cell m: vnbr_list[ vnbr_start[m] .. vnbr_start[m+1] ) = site indices, or -1..-6 for the domain walls
       xmin, xmax, ymin, ymax, zmin, zmax (the Voro++ convention, :1137-1147);
       nearest-site search (VoronoiMeshSnapshot::cellIndex, :1006-1040): a grid of vblock_n^3 blocks over the domain
       (Box::cellIndices), block b = (i*n + j)*n + k lists the cells whose bounding box overlaps it in
       vblock_list[ vblock_start[b] .. vblock_start[b+1] ). */
    const double*  site;
    const int32_t* vnbr_start;    /* num_cells + 1 */
    const int32_t* vnbr_list;
    int32_t        vblock_n;
    const int32_t* vblock_start;  /* vblock_n^3 + 1 */
    const int32_t* vblock_list;
} pmc_grid;

/* ---------------------------------------------------------------- medium ---- */

#define PMC_MAX_MEDIA 4
typedef struct pmc_medium
{
    /* one dust medium component with spatially constant cross sections (Configuration::hasSingleConstantSectionMedium; several of
       them: hasMultipleConstantSectionMedia, see pmc_scene::media) */
    const double* number_density;   /* n[m], num_cells (MediumState, MediumSystem.cpp:868) */
    /* DustMix tables (DustMix.cpp:93-98,112-162): index = locateClip(lambda_border, lambda) */
    int32_t       num_lambda;
    const double* lambda_border;    /* DustMix::_lambdav, num_lambda */
    const double* sigma_ext;        /* num_lambda */
    const double* sigma_sca;        /* num_lambda */
    const double* asymmpar;         /* num_lambda, already clamped to +-0.999999 */
    const double* sigma_abs;        /* num_lambda (DustMix::sectionAbs; sigma_ext = sigma_abs + sigma_sca was formed from it): read only by the
                                       explicit-absorption photon cycle */
} pmc_medium;

typedef struct pmc_options
{
    int32_t force_scattering;       /* PhotonPacketOptions::forceScattering */
    double  min_weight_reduction;   /* default 1e4 */
    int32_t min_scatt_events;       /* default 0 */
    double  path_length_bias;       /* default 0.5 */
    int32_t explicit_absorption;    /* PhotonPacketOptions::explicitAbsorption: scattering and absorption optical depths apart, the packet
                                       weight carries exp(-tau_abs) instead of the albedo (MonteCarloSimulation.cpp:568-569, 729-733, 751-766;
                                       MediumSystem.cpp:905-932, 1075-1110) */
} pmc_options;

/* ---------------------------------------------------------------- source ---- */

enum { PMC_SOURCE_POINT = 1, PMC_SOURCE_SERSIC = 2, PMC_SOURCE_UNIFORM_BOX = 3, PMC_SOURCE_EXP_DISK = 4, PMC_SOURCE_PLUMMER = 5 };
enum { PMC_LAMBDA_OLIGO = 1, PMC_LAMBDA_TABULATED = 2 };
enum { PMC_BIAS_NONE = 0, PMC_BIAS_LOG = 1, PMC_BIAS_LIN = 2 };
enum { PMC_SED_TABULATED = 0, PMC_SED_BLACKBODY = 1 };
enum { PMC_ANGULAR_ISOTROPIC = 0, PMC_ANGULAR_LASER = 1, PMC_ANGULAR_CONICAL = 2, PMC_ANGULAR_NETZER = 3 };
#define PMC_NETZER_POINTS 400 /* bins of the cumulative table of NetzerAngularDistribution (NetzerAngularDistribution.cpp:17) */

typedef struct pmc_source
{
    int32_t kind;                   /* PMC_SOURCE_* (single source; SourceSystem with Ns = 1) */
    double  position[3];            /* point source position; geometric sources: offset added to the sampled position
                                       (OffsetGeometryDecorator.cpp:33-39; zeros: none) */
    double  reff;                   /* Sersic: effective radius; tables of SersicFunction (SersicFunction.cpp:13-77) */
    int32_t sersic_n;               /*   number of table points (101) */
    const double* sersic_s;         /*   _sv */
    const double* sersic_M;         /*   _Mv (cumulative mass, normalised) */
    double  box[6];                 /* uniform box source: xmin,ymin,zmin,xmax,ymax,zmax;
                                       exponential disk (ExpDiskGeometry.cpp:46-68, SepAxGeometry.cpp:11-19): scale length,
                                       scale height, min radius, max radius (0: none), max |z| (0: none);
                                       Plummer sphere (PlummerGeometry.cpp:29-33): scale length;
                                       Sersic and Plummer: box[5] = flattening q of a SpheroidalGeometryDecorator
                                       (SpheroidalGeometryDecorator.cpp:19-25: z -> q z; 0: none) */

    double  packet_luminosity;      /* L/Npp * Lv[h]/Wv[h]  (SourceSystem.cpp:96,105-106), before the lambda weight */

    /* wavelength sampling (NormalizedSource.cpp:73-110) */
    int32_t lambda_mode;            /* PMC_LAMBDA_* */
    /* oligochromatic: xi = 1; 2 uniforms per packet; lambda = oligo_lambda[int(u*n)], weight oligo_weight[i] = s/b */
    int32_t       num_oligo;
    const double* oligo_lambda;
    const double* oligo_weight;
    /* panchromatic: SED tabulated for Random::cdfLogLog (Random.cpp:209-216): lambda, p (normalised), P (cumulative) */
    double        lambda_bias;      /* xi of NormalizedSource (wavelengthBias, default 0.5); 0 => SED only */
    int32_t       num_sed;
    const double* sed_lambda;
    const double* sed_p;
    const double* sed_P;
    int32_t       bias_kind;        /* PMC_BIAS_*: LogWavelengthDistribution / LinWavelengthDistribution */
    double        bias_min, bias_max;
    /* SED::specificLuminosity for the bias weight (NormalizedSource.cpp:91-105).  PMC_SED_TABULATED: log-log
       interpolation of (sed_lambda, sed_p).  PMC_SED_BLACKBODY: f2 / lambda^5 / (exp(f1 / lambda) - 1) / ltot
       (BlackBodySED.cpp:36-39, PlanckFunction.cpp:24-27) */
    int32_t       sed_kind;         /* PMC_SED_* */
    double        sed_f1, sed_f2, sed_ltot;
    /* emission direction of a point source (PointSource.cpp:32-43): isotropic, or an axisymmetric distribution about the unit
       vector angular_axis (AxAngularDistribution.cpp:27-41: the direction is Random::direction(axis, cos theta) with cos theta
       from the distribution; an emission peel-off packet towards k_obs carries the weight probabilityForDirection(k_obs),
       PhotonPacket.cpp:78): LaserAngularDistribution.cpp:11-23, ConicalAngularDistribution.cpp:11-37 (angular_cos_delta =
       cos(openingAngle)), NetzerAngularDistribution.cpp:12-47 (its 401-point cumulative table is rebuilt by the engine) */
    int32_t       angular_kind;     /* PMC_ANGULAR_* */
    double        angular_axis[3];
    double        angular_cos_delta;
} pmc_source;

/* ---------------------------------------------------------------- instruments ---- */

typedef struct pmc_instrument
{
    /* DistantInstrument / FrameInstrument (DistantInstrument.cpp:13-51, FrameInstrument.cpp:12-33) */
    double  kobs[3];
    double  costheta, sintheta, cosphi, sinphi, cosomega, sinomega;
    int32_t nxp, nyp;
    double  xpmin, xpsiz, ypmin, ypsiz;
    int32_t same_observer_as_preceding;
    /* FluxRecorder configuration (FluxRecorder.cpp:185-300) */
    int32_t include_flux_density;        /* SED arrays */
    int32_t include_surface_brightness;  /* IFU arrays */
    int32_t record_components;           /* with a medium: Transparent, PrimaryDirect, PrimaryScattered (+levels) */
    int32_t num_scattering_levels;
    int32_t record_statistics;           /* sum of w^k, k = 0..4 */
    double  redshift;
    /* instrument wavelength grid (DisjointWavelengthGrid.cpp:320-345): bin = ellv[upper_bound(border, lambda)] */
    int32_t        num_lambda;           /* number of bins */
    int32_t        num_border;
    const double*  border;
    const int32_t* ellv;                 /* num_border + 1 */
    /* ApertureInstrument::isInsideAperture (ApertureInstrument.cpp:22-43) for an SEDInstrument with a finite aperture:
       a packet whose position projects farther than the radius from the line of sight is not detected (0: no aperture) */
    double         aperture_radius2;
} pmc_instrument;

/* Layout of one instrument's detector arrays inside the frame buffer (doubles).  Components c:
   record_components ? {0:Transparent, 1:PrimaryDirect, 2:PrimaryScattered, 3+i: level i+1} : {0:Total}.
   sed[c][ell]            at sed_offset  + c*num_lambda + ell
   ifu[c][l + ell*npix]   at ifu_offset  + c*npix*num_lambda + l + ell*npix        (FluxRecorder.cpp:433)
   wsed[k][ell]           at wsed_offset + k*num_lambda + ell                       k = 0..4
   wifu[k][l + ell*npix]  at wifu_offset + k*npix*num_lambda + ...                  (FluxRecorder.cpp:962-1014)
   an offset of -1 means "array not present". */
typedef struct pmc_frame_layout
{
    int64_t num_components;
    int64_t npix;
    int64_t num_lambda;
    int64_t sed_offset, ifu_offset, wsed_offset, wifu_offset;
    int64_t end_offset;
} pmc_frame_layout;

/* ---------------------------------------------------------------- radiation field ---- */

/* RadiationFieldOptions::storeRadiationField (forced scattering only, Configuration.cpp:476-482): every segment of every
   forced-scattering path adds  L * lnmean(e^-tau0, e^-tau1) * ds  to rf[m * num_lambda + ell], with ell the bin of the
   packet's wavelength in the radiation field wavelength grid (MonteCarloSimulation.cpp:641-662; constant perceived
   wavelength, i.e. no kinematics).  Bin lookup as for an instrument grid: ell = ellv[upper_bound(border, lambda)]. */
typedef struct pmc_radiation_field
{
    int32_t        store;         /* 0: off (no table is allocated) */
    int32_t        num_lambda;    /* bins of Configuration::radiationFieldWLG() */
    int32_t        num_border;
    const double*  border;
    const int32_t* ellv;          /* num_border + 1 */
} pmc_radiation_field;

typedef struct pmc_scene
{
    int32_t abi_version;            /* PMC_ABI_VERSION */
    pmc_grid    grid;
    pmc_medium  medium;
    pmc_options options;
    pmc_source  source;
    int32_t     num_instruments;
    const pmc_instrument* instruments;
    pmc_radiation_field radiation_field;
    /* a source system with more than one source (SourceSystem.cpp:75-107).  num_sources <= 1: `source` above is the
       only source.  num_sources > 1: sources[0..num_sources) replace it (each with its own packet_luminosity =
       L/Npp * Lv[h]/Wv[h]), and history index h is launched by source i with source_first[i] <= h < source_first[i+1]
       (SourceSystem::_Iv for the segment's number of packets; num_sources + 1 entries).  At most 16 sources (and 16 instruments). */
    int32_t         num_sources;
    const pmc_source* sources;
    const uint64_t*  source_first;
    /* a medium system with more than one component (MediumSystem.cpp:874-887 optical depths summed over the components in order,
       :678-730 albedo and scattering weights from the components' scattering opacities in the interaction cell, :796-817 the
       scattering component drawn from their cumulative distribution with ONE uniform deviate, :734-767 consolidated peel-off).
       num_media <= 1: `medium` above is the only component.  num_media > 1: media[0..num_media) replace it (each with its own cell
       densities and dust tables).  At most PMC_MAX_MEDIA components. */
    int32_t           num_media;
    const pmc_medium* media;
} pmc_scene;

/* counted work, accumulated over all pmc_run_primary calls since create/reset (roofline inputs) */
typedef struct pmc_counter_values
{
    uint64_t histories;        /* packets launched */
    uint64_t paths;            /* grid walks started (forced-scattering paths + peel-off paths) */
    uint64_t cell_visits;      /* V: segments with m >= 0 over every path walked */
    uint64_t detector_updates; /* U: f64 atomic adds into flux/statistics arrays */
    uint64_t scatterings;      /* scattering events simulated */
    uint64_t stat_overflows;   /* histories whose per-history contribution list overflowed (should be 0) */
    uint64_t rewalk_visits;    /* engine only: cell visits of the second pass over a forced-scattering path (the
                                  reference stores the path instead); NOT part of V */
} pmc_counter_values;

typedef struct pmc_ctx pmc_ctx;

/* --- pure host helpers (no device needed) */
int  pmc_abi_version(void);
const char* pmc_last_error(void);
/* the compiler and the flags this binary was built with (one line).  Results are bit-compatible with the reference only when the device
 * code was compiled with -ffp-contract=off (the reference build has no fused multiply-add): pmc_create probes that on the device and
 * fails with PMC_ERR_DEVICE for a binary built otherwise */
const char* pmc_build_info(void);
/* computes the layout of instrument i and returns the total number of doubles of the whole frame buffer */
int64_t pmc_frame_layout_of(const pmc_scene* scene, int32_t instrument, pmc_frame_layout* out);

/* --- device API */
int pmc_create(const pmc_scene* scene, int32_t device, pmc_ctx** out);
void pmc_destroy(pmc_ctx* ctx);
/* Use caller-owned DEVICE memory (num_doubles f64, zero-initialised by the caller) for the frames instead of the
   context's own allocation -- e.g. a torch tensor, so that torch.distributed can reduce it over RCCL. */
int pmc_bind_frames(pmc_ctx* ctx, double* device_ptr, int64_t num_doubles);
int pmc_clear_frames(pmc_ctx* ctx);
/* Run histories [first, first+count) on the context's stream; results are ACCUMULATED into the frames.  The host
   thread drives the generations of the two device kernels and returns when the segment is complete (the frames
   stay on the device; pmc_download copies them).  The RNG stream of a history depends only on (seed, history
   index), so any partition of [0,Npp) over calls and devices gives the same result up to the summation order of
   the floating-point atomics.
   FluxRecorder::recordContributions keeps every contribution of a history (FluxRecorder.cpp:962-1014, FluxRecorder.hpp:327-338:
   an unbounded list).  The engine keeps one entry per DISTINCT pixel of a history and instrument: four in a head record of the
   history's slot, 44 more in the slot's own list, the rest in chained blocks of 48 entries from a device pool (default: one block per
   four slots, more for a ski file with a large minScattEvents; environment PMC_STAT_POOL_BLOCKS) that GROWS between the generations of
   the segment whenever the histories in flight could use it up.  Only on a device that has no memory left for that are statistics
   lost: the call then returns PMC_ERR_OVERFLOW (the flux arrays are unaffected, the statistics arrays of the segment are incomplete,
   the next segment starts with a full pool). */
int pmc_run_primary(pmc_ctx* ctx, uint64_t first, uint64_t count, uint64_t seed);
/* Progress of a running segment (MonteCarloSimulation::logProgress, MonteCarloSimulation.cpp:522-526,609 -> Log::infoIfElapsed): while
   pmc_run_primary drives its generations it calls `report(user, launched, count)` from the calling thread -- launched = histories of the
   segment that SourceSystem has handed out so far -- at most once per `interval_seconds` (the reference logs every 3 s).  report == NULL
   switches it off (the default). */
typedef void (*pmc_progress_fn)(void* user, uint64_t launched, uint64_t count);
int pmc_set_progress(pmc_ctx* ctx, pmc_progress_fn report, void* user, double interval_seconds);
int pmc_sync(pmc_ctx* ctx);
int pmc_download(pmc_ctx* ctx, double* host_frames, int64_t num_doubles);
double* pmc_frames_device(pmc_ctx* ctx);
int64_t pmc_frames_size(pmc_ctx* ctx);
/* Radiation field table rf[m * num_lambda + ell] (doubles, W m): size (0 if the scene does not store it), device
   pointer, copy to the host, reset to zero.  Like the frames it is ACCUMULATED into by every pmc_run_primary; it is
   complete when the call returns.  (On an octree the engine keeps device logs of a generation's contributions, 24 bytes x
   128 entries per packet slot, instead of adding each one atomically; environment PMC_RF_ATOMICS selects the atomics.) */
int64_t pmc_radiation_field_size(pmc_ctx* ctx);
double* pmc_radiation_field_device(pmc_ctx* ctx);
/* caller-owned DEVICE memory for the table (num_doubles f64, zero-initialised by the caller), e.g. a torch tensor */
int pmc_bind_radiation_field(pmc_ctx* ctx, double* device_ptr, int64_t num_doubles);
int pmc_download_radiation_field(pmc_ctx* ctx, double* host_rf, int64_t num_doubles);
int pmc_clear_radiation_field(pmc_ctx* ctx);
/* milliseconds spent in the walk kernel (the dominant kernel) during the most recent pmc_run_primary, summed over
   its launches and measured with HIP events on the streams they run on (the launches of different slot groups
   overlap, so the sum can exceed the segment time reported by pmc_last_timing) */
int pmc_last_kernel_ms(pmc_ctx* ctx, float* ms);
int pmc_counters(pmc_ctx* ctx, pmc_counter_values* out);
int pmc_reset_counters(pmc_ctx* ctx);
/* Walk one ray on the device with the same traversal code the photon loop uses; k is normalised by the caller.
   Writes up to cap segments (cell index m or -1, length ds) and the number found to *n. */
int pmc_trace_ray(pmc_ctx* ctx, const double r[3], const double k[3], int32_t* m, double* ds, int32_t cap, int32_t* n);
/* number of photon histories kept in flight on the device (default 24 Mi, fewer where the device memory is short; environment PMC_NUM_SLOTS).  The slots are
   divided into slot groups (default 3; environment PMC_NUM_GROUPS) whose generations run on separate streams */
int pmc_set_num_slots(pmc_ctx* ctx, int64_t num_slots);
/* HIP-event timing of the most recent pmc_run_primary: whole segment, sum over its walk-kernel launches, sum over
   its transition + launch kernel launches, and the number of generations (walk, transition, launch kernel triples,
   counted over all slot groups) */
int pmc_last_timing(pmc_ctx* ctx, float* total_ms, float* walk_ms, float* transition_ms, int32_t* generations);
/* octree grids: the two kinds of walk kernel of the most recent pmc_run_primary apart -- the spans of the peel-off kernels
   (MediumSystem::getExtinctionOpticalDepth, all observers) and of the propagation kernel
   (MediumSystem::setExtinctionOpticalDepths + the interaction point), HIP events, summed over the generations; the two run
   side by side on two streams of a slot group, so the spans overlap.  0 on other grids. */
int pmc_last_walk_timing(pmc_ctx* ctx, float* peel_ms, float* prop_ms);
/* (Tuning aids -- launch geometry, counted walk work, device addresses of the hot tables for the microbenchmarks, and the
   switches that select alternative code paths for A/B measurements and cross-checks -- are declared in pmc_tuning.h; a caller that
   replaces runPrimaryEmission needs none of them.  From the ENVIRONMENT the library reads three settings only: PMC_NUM_SLOTS,
   PMC_NUM_GROUPS and PMC_STAT_POOL_BLOCKS.) */

/* ---------------------------------------------------------------- several GPUs: one segment over RCCL ---- */

/* The histories of a segment shard by index over the devices: rank g of G runs [floor(g N / G), floor((g+1) N / G)) --
   the static counterpart of the reference's chunk server (SKIRT/core/MultiHybridParallel.cpp:26-104); the random stream
   of a history depends only on (seed, history index), so the partition does not change the result. */
void pmc_history_range(uint64_t num_packets, int32_t rank, int32_t num_ranks, uint64_t* first, uint64_t* count);

/* RCCL communicator handles cross this ABI as opaque pointers (ncclComm_t of <rccl/rccl.h>).
   pmc_comm_init_all: one communicator per listed device for ONE process that drives them all (ncclCommInitAll);
   pmc_comm_init_rank: the communicator of this process in a job of one process per device (ncclCommInitRank; unique_id
   = the 128 bytes of an ncclUniqueId that rank 0 obtained with pmc_comm_unique_id and handed to the other processes). */
#define PMC_COMM_ID_BYTES 128
int  pmc_comm_init_all(int32_t num_devices, const int32_t* devices, void** comms);
int  pmc_comm_unique_id(void* unique_id);
int  pmc_comm_init_rank(int32_t device, int32_t num_ranks, int32_t rank, const void* unique_id, void** comm);
/* ranks of the communicator and the rank of this handle in it, as RCCL reports them (ncclCommCount, ncclCommUserRank) */
int  pmc_comm_size(void* comm, int32_t* num_ranks, int32_t* rank);
void pmc_comm_destroy(void* comm);

/* End of a segment: the detector arrays of all ranks are summed onto `root` with ONE ncclReduce (f64, sum) on the
   context's stream, in place -- FluxRecorder::flush -> ProcessManager::sumToRoot (SKIRT/core/FluxRecorder.cpp:487-493,
   SKIRT/mpi/ProcessManager.cpp:223-255).  The other ranks' arrays are cleared afterwards (they start the next segment
   from zero, as the reference's do after the sum).  Every rank of the communicator must call it. */
int pmc_reduce_frames(pmc_ctx* ctx, void* comm, int32_t root);
/* The radiation field table of all ranks summed onto ALL ranks (ncclAllReduce): MediumSystem::communicateRadiationField
   -> ProcessManager::sumToAll (SKIRT/core/MediumSystem.cpp:1304-1313). */
int pmc_allreduce_radiation_field(pmc_ctx* ctx, void* comm);

/* ---------------------------------------------------------------- setup: density of a smoothed-particle medium ---- */

/* ParticleSnapshot::density(Position) (SKIRT/core/ParticleSnapshot.cpp:233-243) for MANY positions at once: the setup
   phase of an imported medium evaluates it 100 times per tree node and per cell (DensityTreePolicy.cpp:141,
   MediumSystem.cpp:91-96).  density = sum over the particles listed for the block of the position, IN LIST ORDER, of
   kernel(|r - r_i| / h_i) * rho_i with rho_i = M_i / h_i^3; block lookup as BoxSearch::entitiesFor(Vec)
   (SKIRT/utils/BoxSearch.cpp:14-52,124-135: NR::locateClip on the three separation arrays).  With the cubic-spline
   and uniform kernels the result equals the host evaluation bit for bit (no transcendental function on the way). */
enum { PMC_KERNEL_CUBIC_SPLINE = 1, PMC_KERNEL_UNIFORM = 2 };

typedef struct pmc_particles
{
    int32_t        kernel;         /* PMC_KERNEL_* */
    int64_t        num_particles;
    const double*  particle;       /* [num_particles][5]: x, y, z, h, rho = M / h^3 */
    int32_t        num_blocks;     /* blocks per axis of the search grid */
    const double*  xgrid;          /* num_blocks + 1 separation points per axis (first -inf, last +inf) */
    const double*  ygrid;
    const double*  zgrid;
    const int64_t* block_start;    /* num_blocks^3 + 1; block b = (i*n + j)*n + k */
    const int32_t* block_list;     /* particle indices */
} pmc_particles;

typedef struct pmc_sampler pmc_sampler;
int  pmc_sampler_create(const pmc_particles* particles, int32_t device, pmc_sampler** out);
/* positions: host array [n][3]; density: host array [n] (mass or number density as the particle weights imply) */
int  pmc_sampler_density(pmc_sampler* sampler, const double* positions, int64_t n, double* density);
void pmc_sampler_destroy(pmc_sampler* sampler);

#ifdef __cplusplus
}
#endif
#endif
