// atomic_kinds.hip -- microbenchmark: rate of no-return global atomics into a table by operand type (f64, f32, u32, u64),
// by address pattern (random per lane; the 64 lanes of a wave on 64 consecutive entries at a random base) and by table size.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_kinds.hip -o atomic_kinds && ./atomic_kinds
#include <hip/hip_runtime.h>
#include <cstdio>

template<typename T, int PATTERN> __global__ __launch_bounds__(256) void scatter(T* table, unsigned mask, int perLane)
{
    unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
    const unsigned lane = threadIdx.x & 63u;
    for (int i = 0; i < perLane; ++i)
    {
        idx = idx * 1664525u + 1013904223u;
        unsigned at = (idx >> 8) & mask;
        if (PATTERN == 1) at = ((__shfl(at, 0, 64) & ~63u) + lane) & mask;
        __hip_atomic_fetch_add(table + at, T(1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template<typename T, int PATTERN> void run(const char* name, void* dev, size_t kb)
{
    const unsigned mask = unsigned(kb * 1024 / sizeof(T) - 1);
    const int blocks = 2048, perLane = 1000;
    hipMemset(dev, 0, kb * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((scatter<T, PATTERN>), dim3(blocks), dim3(256), 0, 0, (T*)dev, mask, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL((scatter<T, PATTERN>), dim3(blocks), dim3(256), 0, 0, (T*)dev, mask, perLane);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("table %6zu KB  %-34s %.3e atomics/s\n", kb, name, double(blocks) * 256 * perLane / (ms * 1e-3));
}

int main()
{
    void* dev;
    hipMalloc(&dev, size_t(64) << 20);
    for (size_t kb : {2048, 65536})
    {
        run<double, 0>("f64 random", dev, kb);
        run<float, 0>("f32 random", dev, kb);
        run<unsigned, 0>("u32 random", dev, kb);
        run<unsigned long long, 0>("u64 random", dev, kb);
        run<double, 1>("f64 wave on 64 consecutive", dev, kb);
        run<float, 1>("f32 wave on 64 consecutive", dev, kb);
        run<unsigned, 1>("u32 wave on 64 consecutive", dev, kb);
    }
    return 0;
}
