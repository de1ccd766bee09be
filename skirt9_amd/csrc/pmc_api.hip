// pmc_api.hip -- the extern "C" ABI of include/pmc.h: context life time and scene upload (pmc_create: validation, medium, source and
// instrument tables, LDS plans, launch geometry), frames, radiation field, counters, the single-ray tracer.  The grid tables are built in
// pmc_tables.hip, the generation loop (pmc_run_primary) lives in pmc_run.hip, the RCCL calls in pmc_comm.hip, the switch table of
// include/pmc_tuning.h in pmc_tuning.hip; pmc_context.h is what they share.
#include "pmc_context.h"

namespace
{
    thread_local std::string t_error;
    // constant-memory scene slots (pmc_kernels.hip c_scene): every device has its own copy of the symbol, so the table
    // of live contexts is per device; guarded, because contexts may be created from one host thread per GPU
    constexpr int MAX_DEVICES = 64;
    bool g_slotUsed[MAX_DEVICES][PMC_MAX_CONTEXTS] = {{false}};
    std::mutex g_slotMutex;
}

// error text of the calling thread, for the other translation units of the library (pmc_sampler.hip)
void pmcSetError(const std::string& message)
{
    t_error = message;
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
    // (tuning aid PMC_STREAM_PRIORITY: "prop" = the group streams, which carry the propagation kernel -- the longest kernel of a generation -- and the
    // transition side, at the highest priority and the peel-off side streams at the lowest; "peel" = the other way round)
    int prioGroup = 0, prioPeel = 0;
    if (const char* v = pmcTune("PMC_STREAM_PRIORITY"))
    {
        int least = 0, greatest = 0;
        hipDeviceGetStreamPriorityRange(&least, &greatest);
        const bool prop = std::strcmp(v, "prop") == 0;
        prioGroup = prop ? greatest : least, prioPeel = prop ? least : greatest;
    }
    auto makeStream = [&](hipStream_t* out, int priority) { return hipStreamCreateWithPriority(out, hipStreamDefault, priority); };
    if (makeStream(&ctx->stream, prioGroup) != hipSuccess) return bail(fail(PMC_ERR_DEVICE, "hipStreamCreate failed"));
    ctx->groupStream[0] = ctx->stream;
    for (int g = 1; g < PMC_MAX_GROUPS; ++g)
        if (makeStream(&ctx->groupStream[g], prioGroup) != hipSuccess) return bail(fail(PMC_ERR_DEVICE, "hipStreamCreate failed"));
    for (int g = 0; g < PMC_MAX_GROUPS; ++g)
        if (makeStream(&ctx->peelStream[g], prioPeel) != hipSuccess) return bail(fail(PMC_ERR_DEVICE, "hipStreamCreate failed"));
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
    if (hipHostMalloc(reinterpret_cast<void**>(&ctx->pinned), 24 * sizeof(unsigned long long)) != hipSuccess)
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
        if ((rc = pmcUploadVoronoiGrid(ctx, scene, med))) return bail(rc);
    }
    else
    {
        if ((rc = pmcUploadOctreeGrid(ctx, scene, med, devToCell))) return bail(rc);
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
    D.voro_defer_scan = (scene->grid.kind == PMC_GRID_VORONOI && pmcTune("PMC_VORO_NO_PEEL_KERNEL") == nullptr
                         && pmcTune("PMC_VORO_NO_DEFERRED_SCAN") == nullptr) ? 1 : 0;
    if (scene->grid.kind == PMC_GRID_VORONOI && D.vgen_run && pmcTune("PMC_VORO_NO_DEFERRED_SCAN") == nullptr) D.voro_defer_scan |= 2;
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

}  // extern "C"
