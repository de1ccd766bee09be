// pmc_sampler.hip -- setup-time density of a smoothed-particle medium on the MI355X (include/pmc.h, pmc_sampler_*).
//
// One lane per sample position: locate the block of the search grid (three binary searches over the separation arrays,
// staged in LDS), then walk the block's particle list IN ORDER and accumulate kernel(u) * rho -- exactly the loop of
// ParticleSnapshot::density (SKIRT/core/ParticleSnapshot.cpp:233-243), so that the sums come out bit-identical to the
// host's (IEEE f64 add / mul / div / sqrt, -ffp-contract=off, no transcendental function in the cubic-spline and
// uniform kernels).  Compute-bound f64 work with gathers from the particle table (40 B per particle, L2 / Infinity-Cache
// resident for 10^6 particles); positions and results stream over PCIe in batches.
#include "../../include/pmc.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

extern "C" const char* pmc_last_error(void);
void pmcSetError(const std::string& message);  // pmc_api.hip

namespace
{
    struct DevParticles
    {
        int32_t kernel;
        int32_t nb;
        const double* particle;
        const double* grid;  // xgrid | ygrid | zgrid, (nb + 1) each
        const long long* block_start;
        const int32_t* block_list;
    };

    // NR::locateClip on a separation array (NR.hpp:152-159)
    __device__ __forceinline__ int locateClip(const double* xv, int n, double x)
    {
        if (x < xv[0]) return 0;
        int jl = -1, ju = n - 1;
        while (ju - jl > 1)
        {
            const int jm = (ju + jl) >> 1;
            if (x < xv[jm])
                ju = jm;
            else
                jl = jm;
        }
        return jl;
    }

    // CubicSplineSmoothingKernel::density (CubicSplineSmoothingKernel.cpp:40-50), UniformSmoothingKernel::density
    template<int KERNEL> __device__ __forceinline__ double kernelDensity(double u)
    {
        if (KERNEL == PMC_KERNEL_CUBIC_SPLINE)
        {
            if (u < 0.0 || u >= 1.0)
                return 0.0;
            else if (u < 0.5)
                return 8.0 / M_PI * (1.0 - 6.0 * u * u * (1.0 - u));
            else
                return 8.0 / M_PI * 2.0 * (1.0 - u) * (1.0 - u) * (1.0 - u);
        }
        if (u < 0.0 || u > 1.0) return 0.0;
        return 0.75 / M_PI;
    }

    template<int KERNEL>
    __global__ __launch_bounds__(256) void sampleDensityKernel(DevParticles P, const double* __restrict__ positions, long long n,
                                                               double* __restrict__ density)
    {
        extern __shared__ double grid[];  // 3 (nb + 1) separation points
        const int ng = 3 * (P.nb + 1);
        for (int i = threadIdx.x; i < ng; i += blockDim.x) grid[i] = P.grid[i];
        __syncthreads();
        const double* xg = grid;
        const double* yg = grid + (P.nb + 1);
        const double* zg = yg + (P.nb + 1);
        for (long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (long long)gridDim.x * blockDim.x)
        {
            const double x = positions[3 * s], y = positions[3 * s + 1], z = positions[3 * s + 2];
            const int i = locateClip(xg, P.nb + 1, x);
            const int j = locateClip(yg, P.nb + 1, y);
            const int k = locateClip(zg, P.nb + 1, z);
            const long long b = ((long long)i * P.nb + j) * P.nb + k;
            double sum = 0.;
            const long long e = P.block_start[b + 1];
            for (long long q = P.block_start[b]; q < e; ++q)
            {
                const double* p = P.particle + 5 * (long long)P.block_list[q];
                const double dx = x - p[0], dy = y - p[1], dz = z - p[2];
                const double u = sqrt(dx * dx + dy * dy + dz * dz) / p[3];
                sum += kernelDensity<KERNEL>(u) * p[4];
            }
            density[s] = sum > 0. ? sum : 0.;
        }
    }
}

struct pmc_sampler
{
    int device{0};
    DevParticles dev{};
    std::vector<void*> allocations;
    double* dpos{nullptr};
    double* dout{nullptr};
    long long batch{0};
    hipStream_t stream{nullptr};
};

namespace
{
    int failSampler(int code, const std::string& message)
    {
        pmcSetError(message);
        return code;
    }
    template<typename T> bool uploadTo(pmc_sampler* s, const T* host, size_t count, const T** out)
    {
        void* d = nullptr;
        if (hipMalloc(&d, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return false;
        s->allocations.push_back(d);
        if (count && hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return false;
        *out = static_cast<const T*>(d);
        return true;
    }
}

extern "C" {

int pmc_sampler_create(const pmc_particles* P, int32_t device, pmc_sampler** out)
{
    if (!P || !out) return failSampler(PMC_ERR_INVALID, "null argument");
    *out = nullptr;
    if (P->kernel != PMC_KERNEL_CUBIC_SPLINE && P->kernel != PMC_KERNEL_UNIFORM)
        return failSampler(PMC_ERR_UNSUPPORTED, "smoothing kernel not supported by the device sampler");
    if (P->num_particles < 1 || P->num_blocks < 1 || !P->particle || !P->xgrid || !P->ygrid || !P->zgrid || !P->block_start
        || !P->block_list)
        return failSampler(PMC_ERR_INVALID, "particle tables are missing");
    if (3 * size_t(P->num_blocks + 1) * sizeof(double) > 64 * 1024)
        return failSampler(PMC_ERR_UNSUPPORTED, "search grid with more than 2729 blocks per axis");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1)
        return failSampler(PMC_ERR_DEVICE, "no HIP device available: the device sampler cannot run");
    if (device < 0 || device >= count) return failSampler(PMC_ERR_INVALID, "invalid device index");
    if (hipSetDevice(device) != hipSuccess) return failSampler(PMC_ERR_DEVICE, "hipSetDevice failed");
    pmc_sampler* s = new pmc_sampler();
    s->device = device;
    const size_t nb = size_t(P->num_blocks), nb3 = nb * nb * nb;
    std::vector<double> grid;
    grid.insert(grid.end(), P->xgrid, P->xgrid + nb + 1);
    grid.insert(grid.end(), P->ygrid, P->ygrid + nb + 1);
    grid.insert(grid.end(), P->zgrid, P->zgrid + nb + 1);
    std::vector<long long> starts(P->block_start, P->block_start + nb3 + 1);
    bool ok = uploadTo(s, P->particle, 5 * size_t(P->num_particles), &s->dev.particle) && uploadTo(s, grid.data(), grid.size(), &s->dev.grid)
              && uploadTo(s, starts.data(), starts.size(), &s->dev.block_start)
              && uploadTo(s, P->block_list, size_t(P->block_start[nb3]), &s->dev.block_list);
    s->dev.kernel = P->kernel;
    s->dev.nb = P->num_blocks;
    s->batch = 1 << 22;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&s->dpos), size_t(s->batch) * 3 * sizeof(double)) == hipSuccess
         && hipMalloc(reinterpret_cast<void**>(&s->dout), size_t(s->batch) * sizeof(double)) == hipSuccess
         && hipStreamCreate(&s->stream) == hipSuccess;
    if (!ok)
    {
        pmc_sampler_destroy(s);
        return failSampler(PMC_ERR_NOMEM, "device allocation for the particle sampler failed");
    }
    *out = s;
    return PMC_OK;
}

int pmc_sampler_density(pmc_sampler* s, const double* positions, int64_t n, double* density)
{
    if (!s || n < 0) return failSampler(PMC_ERR_INVALID, "invalid argument");
    if (n == 0) return PMC_OK;
    if (!positions || !density) return failSampler(PMC_ERR_INVALID, "null buffer");
    if (hipSetDevice(s->device) != hipSuccess) return failSampler(PMC_ERR_DEVICE, "hipSetDevice failed");
    const size_t lds = 3 * size_t(s->dev.nb + 1) * sizeof(double);
    for (int64_t first = 0; first < n; first += s->batch)
    {
        const long long m = std::min<long long>(s->batch, n - first);
        if (hipMemcpyAsync(s->dpos, positions + 3 * first, size_t(m) * 3 * sizeof(double), hipMemcpyHostToDevice, s->stream) != hipSuccess)
            return failSampler(PMC_ERR_DEVICE, "hipMemcpy of the sample positions failed");
        const int blocks = int(std::min<long long>((m + 255) / 256, 256 * 16));
        if (s->dev.kernel == PMC_KERNEL_CUBIC_SPLINE)
            hipLaunchKernelGGL(sampleDensityKernel<PMC_KERNEL_CUBIC_SPLINE>, dim3(blocks), dim3(256), lds, s->stream, s->dev, s->dpos, m, s->dout);
        else
            hipLaunchKernelGGL(sampleDensityKernel<PMC_KERNEL_UNIFORM>, dim3(blocks), dim3(256), lds, s->stream, s->dev, s->dpos, m, s->dout);
        if (hipGetLastError() != hipSuccess) return failSampler(PMC_ERR_DEVICE, "launch of the density sampling kernel failed");
        if (hipMemcpyAsync(density + first, s->dout, size_t(m) * sizeof(double), hipMemcpyDeviceToHost, s->stream) != hipSuccess
            || hipStreamSynchronize(s->stream) != hipSuccess)
            return failSampler(PMC_ERR_DEVICE, "density sampling kernel failed");
    }
    return PMC_OK;
}

void pmc_sampler_destroy(pmc_sampler* s)
{
    if (!s) return;
    hipSetDevice(s->device);
    if (s->stream) hipStreamDestroy(s->stream);
    for (void* p : s->allocations) hipFree(p);
    if (s->dpos) hipFree(s->dpos);
    if (s->dout) hipFree(s->dout);
    delete s;
}
}
