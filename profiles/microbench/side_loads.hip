// side_loads.hip -- microbenchmark: a dependent random chase through a 32 MB table of 32-byte records (the octree walk's
// gather) with, every PERIOD steps, a batch of nine coalesced 8-byte loads from a large streaming array (the task record a
// walk kernel reads for its next walk; HBM latency) and one 8-byte store (the result of a walk).  Loads return in order:
// how much does a slow load in a wave's stream delay the gathers behind it?
//   hipcc --offload-arch=gfx950 -O3 side_loads.hip -o side_loads && ./side_loads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>

template<int PERIOD, bool STORE> __global__ __launch_bounds__(256) void chase(const uint4* __restrict__ table, unsigned mask, int steps,
                                                                          const double* __restrict__ stream, size_t streamLen, double* sink, unsigned* out)
{
    const size_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned idx = (unsigned(gid) * 2654435761u) & mask;
    unsigned acc = 0;
    double side = 0.;
    size_t at = gid;
    const size_t plane = streamLen / 9;
#pragma unroll 1
    for (int i = 0; i < steps; ++i)
    {
        const uint4 a = table[2 * idx], b = table[2 * idx + 1];
        if (PERIOD > 0 && (i % PERIOD) == 0)
        {
            at = (at + size_t(gridDim.x) * blockDim.x) % plane;
#pragma unroll
            for (int f = 0; f < 9; ++f) side += stream[f * plane + at];
            if (STORE) sink[at] = side;
        }
        acc += a.y + b.w;
        idx = ((a.x ^ b.z) + i * 0x9E3779B1u) & mask;
    }
    out[gid] = acc + idx + unsigned(side);
}

template<int PERIOD, bool STORE> void run(const char* name, const uint4* dev, const double* stream, size_t streamLen, double* sink, unsigned* out)
{
    const int steps = 2000, blocks = 512;
    const unsigned mask = unsigned(32768 * 1024 / 32 - 1);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((chase<PERIOD, STORE>), dim3(blocks), dim3(256), 0, 0, dev, mask, 10, stream, streamLen, sink, out);
    hipEventRecord(a);
    hipLaunchKernelGGL((chase<PERIOD, STORE>), dim3(blocks), dim3(256), 0, 0, dev, mask, steps, stream, streamLen, sink, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-58s %.3e steps/s  %5.0f ns per step\n", name, double(blocks) * 256 * steps / (ms * 1e-3), ms * 1e6 / steps);
}

int main()
{
    const size_t records = size_t(1) << 21;  // 32 MB of 16-byte halves
    std::vector<uint4> host(records);
    std::mt19937 rng(1);
    for (auto& r : host) r.x = rng(), r.y = rng(), r.z = rng(), r.w = rng();
    uint4* dev;
    unsigned* out;
    double *stream, *sink;
    const size_t streamLen = size_t(9) << 27;  // 9 planes of 1 GB
    hipMalloc(&dev, records * sizeof(uint4));
    hipMalloc(&out, size_t(512) * 256 * sizeof(unsigned));
    hipMalloc(&stream, streamLen * sizeof(double));
    hipMalloc(&sink, (streamLen / 9) * sizeof(double));
    hipMemset(stream, 0, streamLen * sizeof(double));
    hipMemcpy(dev, host.data(), records * sizeof(uint4), hipMemcpyHostToDevice);
    run<0, false>("chase alone (8 waves per CU)", dev, stream, streamLen, sink, out);
    run<64, false>("+ 9 streaming loads every 64 steps", dev, stream, streamLen, sink, out);
    run<16, false>("+ 9 streaming loads every 16 steps", dev, stream, streamLen, sink, out);
    run<8, false>("+ 9 streaming loads every 8 steps", dev, stream, streamLen, sink, out);
    run<4, false>("+ 9 streaming loads every 4 steps", dev, stream, streamLen, sink, out);
    run<8, true>("+ 9 streaming loads and a store every 8 steps", dev, stream, streamLen, sink, out);
    run<1, false>("+ 9 streaming loads every step", dev, stream, streamLen, sink, out);
    return 0;
}
