// xcd_atomics.hip -- microbenchmark: f64 atomic adds into a table, device scope (executed memory-side: the XCDs' L2s are
// not coherent) against WORKGROUP scope into one private copy of the table per XCD (executed in that XCD's L2: every
// writer of a copy sits on the XCD that owns it), followed by a merge of the eight copies.  Checks the sums.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics xcd_atomics.hip -o xcd_atomics && ./xcd_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xccId()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

// MODE 0: device-scope atomics into one table; 1: workgroup-scope atomics into copy[xcc]; 2: device scope into copy[xcc]
template<int MODE> __global__ __launch_bounds__(256) void scatter(double* table, size_t entries, unsigned mask, int perLane, unsigned* xccSeen)
{
    const unsigned xcc = xccId();
    if (threadIdx.x == 0) atomicOr(xccSeen + xcc, 1u);
    double* dst = MODE == 0 ? table : table + size_t(xcc) * entries;
    unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
    for (int i = 0; i < perLane; ++i)
    {
        idx = idx * 1664525u + 1013904223u;
        double* p = dst + ((idx >> 8) & mask);
        if (MODE == 1)
            __hip_atomic_fetch_add(p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else
            __hip_atomic_fetch_add(p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void merge(double* table, size_t entries, double* out)
{
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < entries; i += size_t(gridDim.x) * blockDim.x)
    {
        double s = 0;
        for (int x = 0; x < 8; ++x) s += table[size_t(x) * entries + i];
        out[i] = s;
    }
}

int main()
{
    for (size_t kb : {64, 1024, 8192, 65536})
    {
        const size_t entries = kb * 1024 / 8;
        const unsigned mask = unsigned(entries - 1);
        double *table, *out;
        unsigned* seen;
        hipMalloc(&table, 8 * entries * sizeof(double));
        hipMalloc(&out, entries * sizeof(double));
        hipMalloc(&seen, 8 * sizeof(unsigned));
        const int blocks = 2048, perLane = 2000;
        const double total = double(blocks) * 256 * perLane;
        for (int mode = 0; mode < 3; ++mode)
        {
            hipMemset(table, 0, 8 * entries * sizeof(double));
            hipMemset(seen, 0, 8 * sizeof(unsigned));
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(scatter<0>, dim3(blocks), dim3(256), 0, 0, table, entries, mask, perLane, seen);
            if (mode == 1) hipLaunchKernelGGL(scatter<1>, dim3(blocks), dim3(256), 0, 0, table, entries, mask, perLane, seen);
            if (mode == 2) hipLaunchKernelGGL(scatter<2>, dim3(blocks), dim3(256), 0, 0, table, entries, mask, perLane, seen);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            hipLaunchKernelGGL(merge, dim3(1024), dim3(256), 0, 0, table, entries, out);
            std::vector<double> h(entries);
            hipMemcpy(h.data(), mode == 0 ? table : out, entries * sizeof(double), hipMemcpyDeviceToHost);
            double sum = 0;
            for (double v : h) sum += v;
            unsigned hs[8];
            hipMemcpy(hs, seen, sizeof(hs), hipMemcpyDeviceToHost);
            int nx = 0;
            for (unsigned v : hs) nx += v ? 1 : 0;
            printf("table %6zu KB  %-44s %.3e atomics/s  sum %s (%.0f of %.0f)  XCDs seen %d\n", kb,
                   mode == 0 ? "device scope, one table" : mode == 1 ? "workgroup scope, one copy per XCD" : "device scope, one copy per XCD",
                   total / (ms * 1e-3), sum == total ? "exact" : "WRONG", sum, total, nx);
        }
        hipFree(table);
        hipFree(out);
        hipFree(seen);
    }
    return 0;
}
