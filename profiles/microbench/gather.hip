// gather.hip -- microbenchmark: throughput of DEPENDENT random gathers (pointer chase) from an HBM-resident table,
// the access pattern of the octree walk (one record per step, next address known only after the load returns).
//   hipcc --offload-arch=gfx950 -O3 gather.hip -o gather && ./gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

__global__ void chase16(const uint4* table, size_t strideU4, unsigned mask, int steps, unsigned* out)
{
    unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u & mask;
    unsigned acc = 0;
    for (int i = 0; i < steps; ++i)
    {
        uint4 v = table[size_t(idx) * strideU4];
        acc += v.y;
        idx = v.x & mask;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + idx;
}

int main(int argc, char** argv)
{
    int steps = 2000;
    for (size_t mb : {8, 61, 122, 488})
        for (int stride : {16, 64, 128})
        {
            size_t records = mb * 1024 * 1024 / stride;
            unsigned pow2 = 1;
            while (pow2 * 2 <= records) pow2 *= 2;
            unsigned mask = pow2 - 1;
            std::vector<uint4> host(size_t(pow2) * stride / 16);
            std::mt19937 rng(1);
            for (size_t r = 0; r < pow2; ++r)
            {
                host[r * (stride / 16)].x = rng();
                host[r * (stride / 16)].y = 1;
            }
            uint4* dev;
            hipMalloc(&dev, host.size() * sizeof(uint4));
            hipMemcpy(dev, host.data(), host.size() * sizeof(uint4), hipMemcpyHostToDevice);
            for (int wavesPerSimd : {2, 4, 8})
            {
                int blocks = 256 * wavesPerSimd;  // 256-thread blocks: 4 waves -> one per SIMD
                unsigned* out;
                hipMalloc(&out, size_t(blocks) * 256 * 4);
                hipEvent_t a, b;
                hipEventCreate(&a);
                hipEventCreate(&b);
                chase16<<<blocks, 256>>>(dev, stride / 16, mask, 10, out);
                hipEventRecord(a);
                chase16<<<blocks, 256>>>(dev, stride / 16, mask, steps, out);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                double loads = double(blocks) * 256 * steps;
                printf("table %4zu MB stride %3d B waves/SIMD %d : %.3e gathers/s, %.1f ns per dependent step\n", size_t(pow2) * stride >> 20,
                       stride, wavesPerSimd, loads / (ms * 1e-3), ms * 1e6 / steps);
                hipFree(out);
            }
            hipFree(dev);
        }
    return 0;
}
