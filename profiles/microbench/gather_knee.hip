#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
template<int MODE> __global__ __launch_bounds__(1024) void chase(const uint4* __restrict__ table, unsigned mask, int steps, unsigned* out)
{
    unsigned idx = ((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u) & mask;
    unsigned acc = 0;
#pragma unroll 1
    for (int i = 0; i < steps; ++i)
    {
        if (MODE == 0) { const uint4 v = table[idx]; acc += v.y; idx = (v.x + i * 0x9E3779B1u) & mask; }
        else { const uint4 a = table[2 * idx], b = table[2 * idx + 1]; acc += a.y + b.w; idx = ((a.x ^ b.z) + i * 0x9E3779B1u) & mask; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + idx;
}
int main()
{
    const size_t maxRecords = size_t(1) << 21;
    std::vector<uint4> host(maxRecords);
    std::mt19937 rng(1);
    for (auto& r : host) r.x = rng(), r.y = rng(), r.z = rng(), r.w = rng();
    uint4* dev; unsigned* out;
    hipMalloc(&dev, maxRecords * sizeof(uint4));
    hipMalloc(&out, size_t(512) * 1024 * sizeof(unsigned));
    hipMemcpy(dev, host.data(), maxRecords * sizeof(uint4), hipMemcpyHostToDevice);
    const int steps = 2000;
    for (int mode = 0; mode < 2; ++mode)
        for (int threads : {256, 384, 512, 640, 768, 896, 1024})
            for (int blocks : {256, 512})
            {
                const unsigned mask = unsigned(32768 * 1024 / (mode ? 32 : 16) - 1);
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                if (mode == 0) hipLaunchKernelGGL(chase<0>, dim3(blocks), dim3(threads), 0, 0, dev, mask, 10, out); else hipLaunchKernelGGL(chase<1>, dim3(blocks), dim3(threads), 0, 0, dev, mask, 10, out);
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(chase<0>, dim3(blocks), dim3(threads), 0, 0, dev, mask, steps, out); else hipLaunchKernelGGL(chase<1>, dim3(blocks), dim3(threads), 0, 0, dev, mask, steps, out);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                printf("%s  %2d waves per CU (%d x %d): %.3e steps/s %5.0f ns per step\n", mode ? "two 16 B loads of a 32 B record" : "one 16 B load                  ", threads / 64 * blocks / 256, blocks / 256, threads,
                       double(blocks) * threads * steps / (ms * 1e-3), ms * 1e6 / steps);
            }
    return 0;
}
