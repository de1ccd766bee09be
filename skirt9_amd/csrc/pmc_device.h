// pmc_device.h -- device-side data layout of the MI355X photon-packet engine (shared by host API and kernels).
//
// HBM layout (all read-only during a segment, replicated per GPU):
//   octree leaves   LeafRec[num_cells]   64-B aligned record per cell m: packed dyadic box code, number density,
//                                        6 neighbour links (one per wall) -> ONE cache line per cell visit
//   octree nodes    NodeRec[num_internal] 64-B record per non-leaf node: box code + 8 child links (descent only)
//   coord table     double[3][2^Lmax+1]  the reference's wall coordinates per axis and dyadic index; staged in LDS
//   neighbour CSR   int32                the reference's per-wall neighbour lists of every leaf, in the reference's
//                                        order (cold: only read when a position lies exactly on a cell boundary)
//   Cartesian       double xv/yv/zv (staged in LDS), density double[num_cells]
//   dust tables     lambda_border/sigma_ext/sigma_sca/asymmpar double[num_lambda] (staged in LDS when small)
//   frames          double[frame_size]   detector arrays, accumulated with f64 atomics
#ifndef PMC_DEVICE_H
#define PMC_DEVICE_H

#include "../../include/pmc.h"
#include <stdint.h>

#define PMC_MAX_INSTRUMENTS 4
#define PMC_MAX_CONTEXTS 8   // scene slots in constant memory (live contexts per process and device)
#define PMC_MAX_LEVEL 12
#define PMC_STAT_CAP 48  // per-history contribution list capacity per instrument (FluxRecorder statistics)

// link encoding: >= 0 leaf cell index m; -1 none (outside the grid); <= -2 internal node index = -2 - link
#define PMC_LINK_NONE (-1)

struct LeafRec
{
    uint64_t code;     // level (bits 48..51) | fx (bits 32..47) | fy (16..31) | fz (0..15): fine lower-corner indices
    double   density;  // number density n[m]
    int32_t  link[6];  // neighbour through wall w (same size or coarser leaf, or the internal node that covers finer ones)
    int32_t  pad[2];
    int32_t  pad2[4];
};
static_assert(sizeof(LeafRec) == 64, "LeafRec must be one 64-byte record");

struct NodeRec
{
    uint64_t code;
    int32_t  child[8];
    int32_t  pad[6];
};
static_assert(sizeof(NodeRec) == 64, "NodeRec must be one 64-byte record");

struct DevInstrument
{
    double kx, ky, kz;
    double costheta, sintheta, cosphi, sinphi, cosomega, sinomega;
    double xpmin, xpsiz, ypmin, ypsiz;
    int32_t nxp, nyp;
    int32_t same_observer;
    int32_t include_sed, include_ifu, record_components, num_levels, record_stats;
    int32_t num_lambda, num_border;
    const double* border;   // device
    const int32_t* ellv;    // device
    // frame layout (doubles)
    int64_t sed_offset, ifu_offset, wsed_offset, wifu_offset, npix;
    int32_t num_components;
    int32_t sed_lds_offset;  // offset (doubles) of this instrument's privatised SED block in LDS: [comp][ell] then [5][ell]
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
    const double* coord_tab;     // [3][(1<<lmax)+1]
    const LeafRec* leaves;
    const NodeRec* nodes;
    int32_t root_link;
    const int32_t* nbr_start;    // [6*num_cells + 1]
    const int32_t* nbr_list;     // leaf cell indices
    int32_t num_cells;
    // ---- medium
    int32_t num_lambda;
    const double* lambda_border;
    const double* sigma_ext;
    const double* sigma_sca;
    const double* asymmpar;
    // ---- options
    int32_t force_scattering;
    double  min_weight_reduction;
    int32_t min_scatt_events;
    double  path_length_bias;
    // ---- source
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
    // ---- instruments
    int32_t num_instruments;
    DevInstrument inst[PMC_MAX_INSTRUMENTS];
    int32_t any_stats;
    // ---- outputs
    double* frames;
    unsigned long long* counters;         // pmc_counter_values as 6 x u64
    unsigned long long* history_counter;  // next history offset to hand out
    // per-lane contribution lists: bin[(inst*CAP + e)*lanes + lane], w likewise
    int32_t* stat_bin;
    double*  stat_w;
    int64_t  stat_lanes;
    // ---- LDS carve-up (in doubles from the start of dynamic LDS)
    int32_t lds_grid_off, lds_dust_off, lds_src_off, lds_sed_off, lds_sed_len, lds_total;
    int32_t dust_in_lds;
};

#endif
