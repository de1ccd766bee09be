// fetch_calib.hip -- round 6: known byte counts for the calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE (tools/fetch_calibration.sh):
//   calibStreamRead    every lane reads 16 bytes, coalesced, over `bytes` bytes (a wide streaming read)
//   calibStreamWrite   every lane writes 16 bytes, coalesced, over `bytes` bytes
//   calibScatterWrite  every lane writes ONE 8-byte word into a 64-byte sector of its own (sectors in hashed order): what the result
//                      stores of a walk kernel do when its walks leave slot order
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC profiles/microbench/fetch_calib.hip -o profiles/microbench/libfetchcalib.so
#include <hip/hip_runtime.h>
#include <cstdint>

namespace
{
    __global__ __launch_bounds__(256) void calibStreamRead(const uint4* __restrict__ in, size_t n, unsigned* out)
    {
        unsigned acc = 0;
        for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
        {
            const uint4 v = in[i];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
        if (acc == 0x12345678u) out[0] = acc;
    }
    __global__ __launch_bounds__(256) void calibStreamWrite(uint4* __restrict__ out, size_t n)
    {
        for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
            out[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
    }
    __global__ __launch_bounds__(256) void calibScatterWrite(double* __restrict__ out, size_t sectors)
    {
        // (sectors is a power of two; an odd multiplier permutes the sector indices)
        for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < sectors; i += size_t(gridDim.x) * blockDim.x)
        {
            const size_t at = (i * 0x9E3779B97F4A7C15ull) & (sectors - 1);
            out[at * 8 + (i & 7)] = (double)i;
        }
    }
}

// runs the three kernels once each over `bytes` bytes of a buffer allocated here; returns the milliseconds of each in ms[3]
extern "C" int fetch_calib_run(size_t bytes, float* ms)
{
    void* buf = nullptr;
    unsigned* sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    hipEvent_t e[4];
    for (auto& x : e) hipEventCreate(&x);
    hipEventRecord(e[0]);
    hipLaunchKernelGGL(calibStreamRead, dim3(4096), dim3(256), 0, 0, static_cast<const uint4*>(buf), bytes / 16, sink);
    hipEventRecord(e[1]);
    hipLaunchKernelGGL(calibStreamWrite, dim3(4096), dim3(256), 0, 0, static_cast<uint4*>(buf), bytes / 16);
    hipEventRecord(e[2]);
    hipLaunchKernelGGL(calibScatterWrite, dim3(4096), dim3(256), 0, 0, static_cast<double*>(buf), bytes / 64);
    hipEventRecord(e[3]);
    if (hipEventSynchronize(e[3]) != hipSuccess) return 2;
    for (int i = 0; i < 3; ++i) hipEventElapsedTime(ms + i, e[i], e[i + 1]);
    hipFree(buf), hipFree(sink);
    return 0;
}
