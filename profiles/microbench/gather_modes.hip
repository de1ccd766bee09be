// gather_modes.hip -- microbenchmark: what a walk step's memory operation costs by SHAPE, for a dependent random chase
// through a table of fixed byte size: one 16-byte load per step; the two 16-byte halves of one 32-byte record (the
// octree walk's CellRec); two 16-byte loads of two different records; one 8-byte load.  Is the chip's gather ceiling
// counted in lane-loads (TA / TCP work) or in records (L2 requests, sectors)?
// The next index mixes in the step number: a pure table chase is a random mapping, whose walks all end in the same
// few cycles of ~sqrt(records) entries after ~sqrt(records) steps and then hit in L1 (gather_ceiling.hip and synth_walk.hip
// have that flaw: their figures for long runs are too high).
//   hipcc --offload-arch=gfx950 -O3 gather_modes.hip -o gather_modes && ./gather_modes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>

// MODE 0: one uint4 of a 16-byte record; 1: both uint4 of a 32-byte record; 2: one uint4 each of two 16-byte records
// (the second index from the first record's y); 3: one uint2 of a 16-byte record; 4: as 1 but only the first half
// of the 32-byte record is read (same table, half the lane-loads)
template<int MODE> __global__ __launch_bounds__(256) void chase(const uint4* __restrict__ table, unsigned mask, int steps, unsigned* out)
{
    unsigned idx = ((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u) & mask;
    unsigned acc = 0;
#pragma unroll 1
    for (int i = 0; i < steps; ++i)
    {
        if (MODE == 0)
        {
            const uint4 v = table[idx];
            acc += v.y;
            idx = (v.x + i * 0x9E3779B1u) & mask;
        }
        else if (MODE == 1)
        {
            const uint4 a = table[2 * idx], b = table[2 * idx + 1];
            acc += a.y + b.w;
            idx = ((a.x ^ b.z) + i * 0x9E3779B1u) & mask;
        }
        else if (MODE == 2)
        {
            const uint4 a = table[idx];
            const uint4 b = table[(idx * 2654435761u + 12345u) & mask];
            acc += a.y + b.w;
            idx = ((a.x ^ b.z) + i * 0x9E3779B1u) & mask;
        }
        else if (MODE == 3)
        {
            const uint2 v = *reinterpret_cast<const uint2*>(table + idx);
            acc += v.y;
            idx = (v.x + i * 0x9E3779B1u) & mask;
        }
        else
        {
            const uint4 a = table[2 * idx];
            acc += a.y;
            idx = (a.x + i * 0x9E3779B1u) & mask;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + idx;
}

template<int MODE> void run(const char* name, const uint4* dev, size_t kb, unsigned* out)
{
    const int steps = 2000;
    const size_t recBytes = (MODE == 1 || MODE == 4) ? 32 : 16;
    const unsigned mask = unsigned(kb * 1024 / recBytes - 1);
    for (int blocks : {512, 1024, 2048})
    {
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipLaunchKernelGGL(chase<MODE>, dim3(blocks), dim3(256), 0, 0, dev, mask, 10, out);
        hipEventRecord(a);
        hipLaunchKernelGGL(chase<MODE>, dim3(blocks), dim3(256), 0, 0, dev, mask, steps, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("table %6zu KB  %-44s workgroups %4d: %.3e steps/s  %5.0f ns per step\n", kb, name, blocks, double(blocks) * 256 * steps / (ms * 1e-3),
               ms * 1e6 / steps);
    }
}

int main()
{
    const size_t maxRecords = size_t(1) << 22;  // 64 MB of uint4
    std::vector<uint4> host(maxRecords);
    std::mt19937 rng(1);
    for (auto& r : host) r.x = rng(), r.y = rng(), r.z = rng(), r.w = rng();
    uint4* dev;
    unsigned* out;
    hipMalloc(&dev, maxRecords * sizeof(uint4));
    hipMalloc(&out, size_t(2048) * 256 * sizeof(unsigned));
    hipMemcpy(dev, host.data(), maxRecords * sizeof(uint4), hipMemcpyHostToDevice);
    for (size_t kb : {2048, 32768})
    {
        run<0>("one 16 B load of a 16 B record", dev, kb, out);
        run<1>("two 16 B loads of one 32 B record", dev, kb, out);
        run<4>("first 16 B of a 32 B record", dev, kb, out);
        run<2>("two 16 B loads of two 16 B records", dev, kb, out);
        run<3>("one 8 B load of a 16 B record", dev, kb, out);
    }
    return 0;
}
