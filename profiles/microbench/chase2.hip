// chase2.hip -- microbenchmark: dependent pointer chase through 128-byte records with the walk kernel's access shape:
// per step an 8-byte and a 4-byte load from the SAME record (issued together), then `work` dependent f64 FMAs, the next
// record index coming from the 4-byte load.  Shows what the memory system sustains for this shape at several
// occupancies, with and without the arithmetic of a traversal step.
//   hipcc --offload-arch=gfx950 -O3 chase2.hip -o chase2 && ./chase2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

template<int WORK> __global__ void chase(const char* table, unsigned mask, int steps, double* out)
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
#pragma unroll
        for (int k = 0; k < WORK; ++k) x = __builtin_fma(x, 0.999999, 1e-9);
        acc += x;
        idx = link & mask;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template<int WORK> void run(const char* dev, unsigned mask, size_t mb, int steps)
{
    for (int wavesPerSimd : {2, 4, 6, 8})
    {
        int blocks = 256 * wavesPerSimd;
        double* out;
        hipMalloc(&out, size_t(blocks) * 256 * 8);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        chase<WORK><<<blocks, 256>>>(dev, mask, 10, out);
        hipEventRecord(a);
        chase<WORK><<<blocks, 256>>>(dev, mask, steps, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        double n = double(blocks) * 256 * steps;
        printf("table %4zu MB  fma/step %3d  waves/SIMD %d : %.3e steps/s, %.0f ns per dependent step\n", mb, WORK, wavesPerSimd,
               n / (ms * 1e-3), ms * 1e6 / steps);
        hipFree(out);
    }
}

int main()
{
    const int steps = 1000;
    for (size_t mb : {16, 122, 488, 2048})
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
        run<100>(dev, mask, actual, steps);
        hipFree(dev);
    }
    return 0;
}
