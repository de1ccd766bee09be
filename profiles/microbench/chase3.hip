// chase3.hip -- microbenchmark: the pointer chase of chase2.hip (8-byte + 4-byte load from one 128-byte record per
// step, 100 dependent f64 FMAs) with B divergent branches per step (each taken by about half of the lanes).  Separates
// the cost of exec-mask control flow from memory and arithmetic.
//   hipcc --offload-arch=gfx950 -O3 chase3.hip -o chase3 && ./chase3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

template<int BRANCHES> __global__ void chase(const char* table, unsigned mask, int steps, double* out)
{
    unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u & mask;
    double acc = 1.0;
    for (int i = 0; i < steps; ++i)
    {
        const char* rec = table + (size_t(idx) << 7);
        const double d = *reinterpret_cast<const double*>(rec + 8);
        const unsigned wq = (idx >> 3) % 24u;
        const unsigned link = *reinterpret_cast<const unsigned*>(rec + 16 + 4 * wq);
        double x = d;
        constexpr int CHUNK = 100 / (BRANCHES > 0 ? BRANCHES : 1);
#pragma unroll
        for (int b = 0; b < (BRANCHES > 0 ? BRANCHES : 1); ++b)
        {
            // the same total number of FMAs, but each chunk behind a data-dependent branch (when BRANCHES > 0)
            if (BRANCHES == 0 || ((link >> b) & 1u) || __builtin_amdgcn_readfirstlane(i) < 0)
            {
#pragma unroll
                for (int k = 0; k < CHUNK; ++k) x = __builtin_fma(x, 0.999999, 1e-9);
            }
            else
            {
#pragma unroll
                for (int k = 0; k < CHUNK; ++k) x = __builtin_fma(x, 0.999998, 2e-9);
            }
        }
        acc += x;
        idx = link & mask;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template<int BRANCHES> void run(const char* dev, unsigned mask, size_t mb, int steps)
{
    for (int wavesPerSimd : {1, 2, 6})
    {
        int blocks = 256 * wavesPerSimd;
        double* out;
        hipMalloc(&out, size_t(blocks) * 256 * 8);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        chase<BRANCHES><<<blocks, 256>>>(dev, mask, 10, out);
        hipEventRecord(a);
        chase<BRANCHES><<<blocks, 256>>>(dev, mask, steps, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        double n = double(blocks) * 256 * steps;
        printf("table %4zu MB  branches/step %2d  waves/SIMD %d : %.3e steps/s, %.0f ns per dependent step\n", mb, BRANCHES, wavesPerSimd,
               n / (ms * 1e-3), ms * 1e6 / steps);
        hipFree(out);
    }
}

int main()
{
    const int steps = 1000;
    for (size_t mb : {16, 128})
    {
        size_t records = mb * 1024 * 1024 / 128;
        unsigned pow2 = 1;
        while (size_t(pow2) * 2 <= records) pow2 *= 2;
        unsigned mask = pow2 - 1;
        std::vector<unsigned> host(size_t(pow2) * 32);
        std::mt19937 rng(1);
        for (size_t r = 0; r < pow2; ++r)
        {
            for (int q = 0; q < 24; ++q) host[r * 32 + 4 + q] = rng();
            double one = 1.0;
            memcpy(&host[r * 32 + 2], &one, 8);
        }
        char* dev;
        hipMalloc(&dev, host.size() * 4);
        hipMemcpy(dev, host.data(), host.size() * 4, hipMemcpyHostToDevice);
        size_t actual = size_t(pow2) * 128 >> 20;
        run<0>(dev, mask, actual, steps);
        run<4>(dev, mask, actual, steps);
        run<10>(dev, mask, actual, steps);
        run<20>(dev, mask, actual, steps);
        hipFree(dev);
    }
    return 0;
}
