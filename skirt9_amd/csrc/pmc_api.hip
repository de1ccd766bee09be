// pmc_api.hip -- the extern "C" ABI of include/pmc.h: scene upload, launches, downloads.
//
// pmc_create turns the reference-shaped tables of pmc_scene into the device layout of pmc_device.h.  For the
// octree this means: verify that every node box is consistent with one per-axis dyadic coordinate table (true for
// any tree built by recursive midpoint subdivision, OctTreeNode.cpp:22-33), build that table from the reference's
// own doubles, and replace the per-wall neighbour lists by one link per wall (the same-size-or-coarser neighbour
// leaf, or the internal node that covers the finer neighbours).  The reference lists themselves are uploaded too,
// re-indexed by cell, for the exact fallback path.

#include "pmc_device.h"
#include "../../include/pmc_layout.h"
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

extern "C" hipError_t pmcUploadScene(int slot, const DevScene* scene, hipStream_t stream);
extern "C" hipError_t pmcLaunchPrimary(int slot, int gridKind, uint64_t first, uint64_t count, uint64_t seed, int grid,
                                       int block, size_t ldsBytes, hipStream_t stream);
extern "C" hipError_t pmcLaunchTrace(int slot, int gridKind, const double r[3], const double k[3], int32_t* m, double* ds,
                                     int32_t cap, int32_t* n, size_t ldsBytes, hipStream_t stream);

namespace
{
    thread_local std::string t_error;
    // constant-memory scene slots (pmc_kernels.hip c_scene): one per live context
    bool g_slotUsed[PMC_MAX_CONTEXTS] = {false};

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
    hipEvent_t evStart{nullptr}, evStop{nullptr};
    bool timed{false};
    DevScene dev{};
    std::vector<void*> allocations;
    double* frames{nullptr};
    bool ownFrames{true};
    int64_t frameSize{0};
    size_t ldsBytes{0};
    int block{256};
    int grid{0};
    int64_t statLanes{0};

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
    template<typename T> int allocate(size_t count, T** out, bool zero)
    {
        *out = nullptr;
        if (!count) return PMC_OK;
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, count * sizeof(T));
        if (e != hipSuccess) return hipFail(e, "hipMalloc");
        allocations.push_back(d);
        if (zero)
        {
            e = hipMemset(d, 0, count * sizeof(T));
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
        std::vector<NodeRec> internals;  // by internal index
        std::vector<int32_t> nbrStart, nbrList;
        int32_t rootLink{0};
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
        auto code = [&](int id) -> uint64_t {
            return ((uint64_t)g.node_level[id] << 48) | ((uint64_t)fx[id] << 32) | ((uint64_t)fy[id] << 16) | (uint64_t)fz[id];
        };
        auto linkOf = [&](int id) -> int32_t {
            if (id < 0) return PMC_LINK_NONE;
            return g.node_first_child[id] < 0 ? g.node_cell[id] : (-2 - internalIndex[id]);
        };
        T.rootLink = linkOf(0);

        // node lookup by (level, fine coordinates) for the same-size neighbour search: walk up/down the tree
        // neighbour of node `id` through `wall`: the deepest node at level <= level(id) whose box covers the region
        // adjacent to the wall; found by descending from the root towards a point just across the wall centre
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
        T.leaves.assign(numCells, LeafRec{});
        T.internals.assign(numInternal, NodeRec{});
        T.nbrStart.assign(6 * size_t(numCells) + 1, 0);
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
        {
            int id = nodeOfCell[m];
            if (id < 0) return fail(PMC_ERR_INVALID, "cell without a leaf node");
            LeafRec& rec = T.leaves[m];
            rec.code = code(id);
            rec.density = density[m];
            for (int wall = 0; wall < 6; ++wall)
            {
                rec.link[wall] = linkOf(covering(id, wall));
                // the reference's neighbour list of this leaf, re-indexed by cell
                T.nbrStart[6 * size_t(m) + wall] = (int32_t)T.nbrList.size();
                for (int q = g.nbr_start[6 * size_t(id) + wall]; q < g.nbr_start[6 * size_t(id) + wall + 1]; ++q)
                {
                    int nb = g.nbr_list[q];
                    if (g.node_first_child[nb] >= 0)
                        return fail(PMC_ERR_INVALID, "neighbour list of a leaf contains a non-leaf node");
                    T.nbrList.push_back(g.node_cell[nb]);
                }
            }
        }
        T.nbrStart[6 * size_t(numCells)] = (int32_t)T.nbrList.size();
        return PMC_OK;
    }
}

extern "C" {

int pmc_abi_version(void)
{
    return PMC_ABI_VERSION;
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
    for (void* p : ctx->allocations) hipFree(p);
    if (ctx->evStart) hipEventDestroy(ctx->evStart);
    if (ctx->evStop) hipEventDestroy(ctx->evStop);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    if (ctx->slot >= 0) g_slotUsed[ctx->slot] = false;
    delete ctx;
}

int pmc_create(const pmc_scene* scene, int32_t device, pmc_ctx** out)
{
    if (!scene || !out) return fail(PMC_ERR_INVALID, "null argument");
    *out = nullptr;
    if (scene->abi_version != PMC_ABI_VERSION) return fail(PMC_ERR_INVALID, "pmc_scene ABI version mismatch");
    if (scene->num_instruments < 1 || scene->num_instruments > PMC_MAX_INSTRUMENTS)
        return fail(PMC_ERR_UNSUPPORTED, "between 1 and " + std::to_string(PMC_MAX_INSTRUMENTS) + " instruments are supported");
    if (scene->grid.kind != PMC_GRID_CARTESIAN && scene->grid.kind != PMC_GRID_OCTREE)
        return fail(PMC_ERR_UNSUPPORTED, "unsupported grid kind");
    if (scene->instruments[0].same_observer_as_preceding) return fail(PMC_ERR_INVALID, "first instrument cannot share an observer");

    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count < 1)
        return fail(PMC_ERR_DEVICE, "no HIP device available: the MI355X engine cannot run (there is no CPU fallback)");
    if (device < 0 || device >= count) return fail(PMC_ERR_INVALID, "invalid device index");
    HIP_TRY(hipSetDevice(device));

    pmc_ctx* ctx = new pmc_ctx();
    ctx->device = device;
    for (int sl = 0; sl < PMC_MAX_CONTEXTS && ctx->slot < 0; ++sl)
        if (!g_slotUsed[sl])
        {
            g_slotUsed[sl] = true;
            ctx->slot = sl;
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
    if (hipEventCreate(&ctx->evStart) != hipSuccess || hipEventCreate(&ctx->evStop) != hipSuccess)
        return bail(fail(PMC_ERR_DEVICE, "hipEventCreate failed"));

    DevScene& D = ctx->dev;
    const pmc_grid& g = scene->grid;
    D.grid_kind = g.kind;
    D.gx0 = g.xmin, D.gy0 = g.ymin, D.gz0 = g.zmin;
    D.gx1 = g.xmax, D.gy1 = g.ymax, D.gz1 = g.zmax;
    D.eps = g.eps;
    D.num_cells = g.num_cells;
    int ldsDoubles = 0;
    D.lds_grid_off = 0;
    if (g.kind == PMC_GRID_CARTESIAN)
    {
        D.nx = g.nx, D.ny = g.ny, D.nz = g.nz;
        if ((rc = ctx->upload(g.xv, g.nx + 1, &D.xv))) return bail(rc);
        if ((rc = ctx->upload(g.yv, g.ny + 1, &D.yv))) return bail(rc);
        if ((rc = ctx->upload(g.zv, g.nz + 1, &D.zv))) return bail(rc);
        if ((rc = ctx->upload(scene->medium.number_density, g.num_cells, &D.cell_density))) return bail(rc);
        ldsDoubles += (g.nx + 1) + (g.ny + 1) + (g.nz + 1);
        D.lmax = 0;
    }
    else
    {
        TreeBuild T;
        if ((rc = buildTree(g, scene->medium.number_density, T))) return bail(rc);
        D.lmax = T.lmax;
        D.root_link = T.rootLink;
        if ((rc = ctx->upload(T.table.data(), T.table.size(), &D.coord_tab))) return bail(rc);
        if ((rc = ctx->upload(T.leaves.data(), T.leaves.size(), &D.leaves))) return bail(rc);
        if ((rc = ctx->upload(T.internals.data(), T.internals.size(), &D.nodes))) return bail(rc);
        if ((rc = ctx->upload(T.nbrStart.data(), T.nbrStart.size(), &D.nbr_start))) return bail(rc);
        if ((rc = ctx->upload(T.nbrList.data(), T.nbrList.size(), &D.nbr_list))) return bail(rc);
        ldsDoubles += 3 * T.tabn;
    }

    // ---- medium
    const pmc_medium& med = scene->medium;
    D.num_lambda = med.num_lambda;
    if ((rc = ctx->upload(med.lambda_border, med.num_lambda, &D.lambda_border))) return bail(rc);
    if ((rc = ctx->upload(med.sigma_ext, med.num_lambda, &D.sigma_ext))) return bail(rc);
    if ((rc = ctx->upload(med.sigma_sca, med.num_lambda, &D.sigma_sca))) return bail(rc);
    if ((rc = ctx->upload(med.asymmpar, med.num_lambda, &D.asymmpar))) return bail(rc);
    D.lds_dust_off = ldsDoubles;
    D.dust_in_lds = med.num_lambda <= 2048;  // <= 64 KiB for the four tables
    if (D.dust_in_lds) ldsDoubles += 4 * med.num_lambda;

    D.force_scattering = scene->options.force_scattering;
    D.min_weight_reduction = scene->options.min_weight_reduction;
    D.min_scatt_events = scene->options.min_scatt_events;
    D.path_length_bias = scene->options.path_length_bias;

    // ---- source
    const pmc_source& src = scene->source;
    D.source_kind = src.kind;
    std::memcpy(D.src_pos, src.position, sizeof(D.src_pos));
    D.reff = src.reff;
    D.sersic_n = src.sersic_n;
    std::memcpy(D.src_box, src.box, sizeof(D.src_box));
    D.packet_luminosity = src.packet_luminosity;
    D.lambda_mode = src.lambda_mode;
    D.num_oligo = src.num_oligo;
    D.lambda_bias = src.lambda_bias;
    D.num_sed = src.num_sed;
    D.bias_kind = src.bias_kind;
    D.bias_min = src.bias_min;
    D.bias_max = src.bias_max;
    D.lds_src_off = ldsDoubles;
    if (src.kind == PMC_SOURCE_SERSIC)
    {
        if (src.sersic_n < 2) return bail(fail(PMC_ERR_INVALID, "Sersic source without tables"));
        if ((rc = ctx->upload(src.sersic_s, src.sersic_n, &D.sersic_s))) return bail(rc);
        if ((rc = ctx->upload(src.sersic_M, src.sersic_n, &D.sersic_M))) return bail(rc);
        ldsDoubles += 2 * src.sersic_n;
    }
    if (src.lambda_mode == PMC_LAMBDA_OLIGO)
    {
        if (src.num_oligo < 1) return bail(fail(PMC_ERR_INVALID, "oligochromatic source without wavelengths"));
        if ((rc = ctx->upload(src.oligo_lambda, src.num_oligo, &D.oligo_lambda))) return bail(rc);
        if ((rc = ctx->upload(src.oligo_weight, src.num_oligo, &D.oligo_weight))) return bail(rc);
    }
    else if (src.lambda_mode == PMC_LAMBDA_TABULATED)
    {
        if (src.num_sed < 2) return bail(fail(PMC_ERR_INVALID, "tabulated source without SED table"));
        if ((rc = ctx->upload(src.sed_lambda, src.num_sed, &D.sed_lambda))) return bail(rc);
        if ((rc = ctx->upload(src.sed_p, src.num_sed, &D.sed_p))) return bail(rc);
        if ((rc = ctx->upload(src.sed_P, src.num_sed, &D.sed_P))) return bail(rc);
    }
    else
        return bail(fail(PMC_ERR_UNSUPPORTED, "unsupported wavelength sampling mode"));

    // ---- instruments and frame layout
    D.num_instruments = scene->num_instruments;
    D.lds_sed_off = ldsDoubles;
    int sedDoubles = 0;
    D.any_stats = 0;
    for (int i = 0; i < scene->num_instruments; ++i)
    {
        const pmc_instrument& I = scene->instruments[i];
        DevInstrument& d = D.inst[i];
        d.kx = I.kobs[0], d.ky = I.kobs[1], d.kz = I.kobs[2];
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
        if (I.redshift != 0.) return bail(fail(PMC_ERR_UNSUPPORTED, "instrument redshift is not supported"));
        d.num_lambda = I.num_lambda;
        d.num_border = I.num_border;
        if ((rc = ctx->upload(I.border, I.num_border, &d.border))) return bail(rc);
        if ((rc = ctx->upload(I.ellv, I.num_border + 1, &d.ellv))) return bail(rc);
        pmc_frame_layout L;
        ctx->frameSize = pmc_layout_compute(scene, i, &L);
        d.sed_offset = L.sed_offset, d.ifu_offset = L.ifu_offset, d.wsed_offset = L.wsed_offset, d.wifu_offset = L.wifu_offset;
        d.npix = L.npix;
        d.num_components = (int32_t)L.num_components;
        d.sed_lds_offset = sedDoubles;
        if (d.include_sed) sedDoubles += (d.num_components + (d.record_stats ? 5 : 0)) * d.num_lambda;
        if (d.record_stats) D.any_stats = 1;
    }
    D.lds_sed_len = sedDoubles;
    ldsDoubles += sedDoubles;
    D.lds_total = ldsDoubles;
    ctx->ldsBytes = size_t(ldsDoubles) * sizeof(double);
    if (ctx->ldsBytes > 160 * 1024)
        return bail(fail(PMC_ERR_UNSUPPORTED, "scene tables need " + std::to_string(ctx->ldsBytes) + " bytes of LDS (> 160 KiB)"));

    // ---- launch geometry: persistent workgroups, as many as stay resident
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return bail(fail(PMC_ERR_DEVICE, "hipGetDeviceProperties failed"));
    int perCU = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / std::max<size_t>(ctx->ldsBytes, 1)));
    perCU = std::min(perCU, 4);
    ctx->grid = prop.multiProcessorCount * perCU;
    ctx->block = 256;
    ctx->statLanes = 0;

    // ---- outputs
    if ((rc = ctx->allocate<double>(ctx->frameSize, &ctx->frames, true))) return bail(rc);
    D.frames = ctx->frames;
    if ((rc = ctx->allocate<unsigned long long>(16, &D.counters, true))) return bail(rc);
    D.history_counter = D.counters + 15;
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

int pmc_bind_frames(pmc_ctx* ctx, double* device_ptr, int64_t num_doubles)
{
    if (!ctx || !device_ptr) return fail(PMC_ERR_INVALID, "null argument");
    if (num_doubles != ctx->frameSize) return fail(PMC_ERR_INVALID, "frame buffer size mismatch");
    ctx->frames = device_ptr;
    ctx->ownFrames = false;
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
    // per-lane contribution lists for the statistics
    const int64_t lanes = int64_t(ctx->grid) * ctx->block;
    if (D.any_stats && lanes != ctx->statLanes)
    {
        size_t entries = size_t(D.num_instruments) * PMC_STAT_CAP * size_t(lanes);
        int rc;
        if ((rc = ctx->allocate<int32_t>(entries, &D.stat_bin, false))) return rc;
        if ((rc = ctx->allocate<double>(entries, &D.stat_w, false))) return rc;
        ctx->statLanes = lanes;
        ctx->sceneDirty = true;
    }
    if (D.stat_lanes != lanes) ctx->sceneDirty = true;
    D.stat_lanes = lanes;
    if (ctx->sceneDirty)
    {
        // make sure no earlier launch of this context still reads the slot, then refresh it
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(pmcUploadScene(ctx->slot, &D, ctx->stream));
        ctx->sceneDirty = false;
    }
    HIP_TRY(hipMemsetAsync(D.history_counter, 0, sizeof(unsigned long long), ctx->stream));
    HIP_TRY(hipEventRecord(ctx->evStart, ctx->stream));
    HIP_TRY(pmcLaunchPrimary(ctx->slot, D.grid_kind, first, count, seed, ctx->grid, ctx->block, ctx->ldsBytes, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->evStop, ctx->stream));
    ctx->timed = true;
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
    if (!ctx->timed) return fail(PMC_ERR_INVALID, "no kernel has been launched yet");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventSynchronize(ctx->evStop));
    HIP_TRY(hipEventElapsedTime(ms, ctx->evStart, ctx->evStop));
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

int pmc_counters(pmc_ctx* ctx, pmc_counter_values* out)
{
    if (!ctx || !out) return fail(PMC_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    unsigned long long host[16];
    HIP_TRY(hipMemcpy(host, ctx->dev.counters, sizeof(host), hipMemcpyDeviceToHost));
    out->histories = host[0];
    out->paths = host[1];
    out->cell_visits = host[2];
    out->detector_updates = host[3];
    out->scatterings = host[4];
    out->stat_overflows = host[5];
    out->rewalk_visits = host[6];
    return PMC_OK;
}

int pmc_reset_counters(pmc_ctx* ctx)
{
    if (!ctx) return fail(PMC_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemsetAsync(ctx->dev.counters, 0, 15 * sizeof(unsigned long long), ctx->stream));
    return PMC_OK;
}

int pmc_trace_ray(pmc_ctx* ctx, const double r[3], const double k[3], int32_t* m, double* ds, int32_t cap, int32_t* n)
{
    if (!ctx || !r || !k || !m || !ds || !n || cap < 0) return fail(PMC_ERR_INVALID, "invalid argument");
    HIP_TRY(hipSetDevice(ctx->device));
    int32_t* dm = nullptr;
    double* dds = nullptr;
    int32_t* dn = nullptr;
    HIP_TRY(hipMalloc(&dm, sizeof(int32_t) * std::max(cap, 1)));
    HIP_TRY(hipMalloc(&dds, sizeof(double) * std::max(cap, 1)));
    HIP_TRY(hipMalloc(&dn, sizeof(int32_t)));
    size_t gridLds = size_t(ctx->dev.lds_dust_off) * sizeof(double);
    hipError_t e = hipSuccess;
    if (ctx->sceneDirty)
    {
        e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = pmcUploadScene(ctx->slot, &ctx->dev, ctx->stream);
        if (e == hipSuccess) ctx->sceneDirty = false;
    }
    if (e == hipSuccess)
        e = pmcLaunchTrace(ctx->slot, ctx->dev.grid_kind, r, k, dm, dds, cap, dn, gridLds, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    int rc = PMC_OK;
    if (e != hipSuccess)
        rc = hipFail(e, "trace kernel");
    else
    {
        hipMemcpy(n, dn, sizeof(int32_t), hipMemcpyDeviceToHost);
        int32_t got = std::min(*n, cap);
        hipMemcpy(m, dm, sizeof(int32_t) * got, hipMemcpyDeviceToHost);
        hipMemcpy(ds, dds, sizeof(double) * got, hipMemcpyDeviceToHost);
    }
    hipFree(dm);
    hipFree(dds);
    hipFree(dn);
    return rc;
}
}
