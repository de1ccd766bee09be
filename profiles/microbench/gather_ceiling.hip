// gather_ceiling.hip -- microbenchmark: rate of dependent random 16-byte gathers (the walk step's memory operation) by
// table size (L1-, L2-, Infinity-Cache-resident) and by the number of CUs in use: is the ceiling per CU or chip-wide?
// FLAWED (kept for the record of round 2's first half): a pure table chase is a random mapping whose walks fall into the
// same few cycles of ~sqrt(N) entries and then hit in L1: the rates of long runs are too high.  See gather_modes.hip /
// gather_knee.hip, which mix the step number into the index.
//   hipcc --offload-arch=gfx950 -O3 gather_ceiling.hip -o gather_ceiling && ./gather_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>

__global__ __launch_bounds__(256) void chase(const uint4* __restrict__ table, unsigned mask, int steps, unsigned* out)
{
    unsigned idx = ((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u) & mask;
    unsigned acc = 0;
#pragma unroll 1
    for (int i = 0; i < steps; ++i)
    {
        const uint4 v = table[idx];
        acc += v.y;
        idx = v.x & mask;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + idx;
}

int main()
{
    const size_t maxRecords = size_t(1) << 22;
    std::vector<uint4> host(maxRecords);
    std::mt19937 rng(1);
    for (auto& r : host) r.x = rng(), r.y = rng(), r.z = rng(), r.w = rng();
    uint4* dev;
    unsigned* out;
    hipMalloc(&dev, maxRecords * sizeof(uint4));
    hipMalloc(&out, size_t(256) * 8 * 256 * sizeof(unsigned));
    hipMemcpy(dev, host.data(), maxRecords * sizeof(uint4), hipMemcpyHostToDevice);
    const int steps = 2000;
    for (size_t kb : {16, 256, 2048, 16384, 65536})
        for (int blocks : {64, 256, 512, 1024})
        {
            const unsigned mask = unsigned(kb * 1024 / 16 - 1);
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            hipLaunchKernelGGL(chase, dim3(blocks), dim3(256), 0, 0, dev, mask, 10, out);
            hipEventRecord(a);
            hipLaunchKernelGGL(chase, dim3(blocks), dim3(256), 0, 0, dev, mask, steps, out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            printf("table %6zu KB  workgroups %4d (%.2f per CU): %.3e gathers/s  %.3e per CU in use  %5.0f ns per step\n", kb, blocks, blocks / 256.0,
                   double(blocks) * 256 * steps / (ms * 1e-3), double(blocks) * 256 * steps / (ms * 1e-3) / (blocks < 256 ? blocks : 256), ms * 1e6 / steps);
        }
    return 0;
}
