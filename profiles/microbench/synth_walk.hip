// synth_walk.hip -- microbenchmark: what does the SHAPE of the octree walk step cost, feature by feature?
// A dependent chase through a 46 MB table of 16-byte records (one gather per step, as the walk kernels), to which the
// features of the real step are added one at a time: f64 arithmetic that depends on the gathered record, six LDS reads
// at record-dependent offsets, partially filled waves, a second dependent load for a fraction of the lanes.
// FLAWED in the same way as gather_ceiling.hip (the chase falls into short cycles): see gather_modes.hip / gather_knee.hip.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off synth_walk.hip -o synth_walk && ./synth_walk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

// load flavours of the gather: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0
template<int FL> __device__ __forceinline__ uint4 gather16(const uint4* p)
{
    uint4 v;
    if (FL == 0) asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (FL == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (FL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (FL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (FL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (FL == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template<int NVALU, int NLDS, int DESC, int FL = 0>
__global__ __launch_bounds__(256) void synth(const uint4* __restrict__ table, const unsigned* __restrict__ nodes, unsigned mask, int steps,
                                             int lanes, double kx, double ky, double kz, double* out)
{
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 3 * 1025; i += blockDim.x) lds[i] = 1.0 + i * 1e-3;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned idx = ((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u) & mask;
    double rx = 0.1 + lane * 1e-3, ry = 0.2, rz = 0.3, tau = 0., ds = 1e-3;
    if (lane < lanes)
    {
#pragma unroll 1
        for (int i = 0; i < steps; ++i)
        {
            const uint4 v = gather16<FL>(table + idx);
            double dens = __longlong_as_double(((long long)(v.y & 0x000FFFFFu | 0x3FF00000u) << 32) | v.z);
            unsigned next = v.x;
            if (DESC > 0)
            {
                // a second dependent load for ~DESC/256 of the lanes
                if ((v.w & 255u) < (unsigned)DESC) next = nodes[(v.x >> 3) & mask];
            }
            if (NVALU > 0)
            {
                tau += dens * ds * 1e-9;
                const double step = ds + 1e-12;
                rx += kx * step, ry += ky * step, rz += kz * step;
            }
            double w[6] = {1.5, 2.5, 1.5, 2.5, 1.5, 2.5};
            if (NLDS > 0)
            {
                const unsigned o = v.w;
                w[0] = lds[o & 1023u], w[1] = lds[(o & 1023u) + 1];
                w[2] = lds[1025 + ((o >> 10) & 1023u)], w[3] = lds[1026 + ((o >> 10) & 1023u)];
                w[4] = lds[2050 + ((o >> 20) & 1023u)], w[5] = lds[2051 + ((o >> 20) & 1023u)];
            }
            if (NVALU > 0)
            {
                // the shape of the step's arithmetic: six differences, minimum, three refined quotients, minimum
                const double x0 = w[0] - rx, x1 = w[1] - rx, y0 = w[2] - ry, y1 = w[3] - ry, z0 = w[4] - rz, z1 = w[5] - rz;
                const double clear = fmin(fmin(fmin(x1, -x0), fmin(y1, -y0)), fmin(z1, -z0));
                double q[3];
                const double a[3] = {x1, y1, z1}, k[3] = {kx, ky, kz};
#pragma unroll
                for (int j = 0; j < 3; ++j)
                {
                    const double y = 1.0 / 3.0;
                    double t = a[j] * y;
#pragma unroll
                    for (int r = 0; r < NVALU / 50; ++r)
                    {
                        double e = __builtin_fma(-k[j], t, a[j]);
                        t = __builtin_fma(e, y, t);
                    }
                    q[j] = t;
                }
                ds = fmin(q[0], fmin(q[1], q[2])) * 1e-6 + (clear > 0. ? 1e-3 : 2e-3);
            }
            idx = next & mask;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = rx + ry + rz + tau + ds + idx;
}

template<int NVALU, int NLDS, int DESC, int FL = 0> void run(const char* name, const uint4* dev, const unsigned* nodes, unsigned mask, double* out)
{
    const int steps = 1000;
    for (int lanes : {48})
        for (int wavesPerSimd : {2, 3, 6})
        {
            const int blocks = 256 * wavesPerSimd;
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            hipLaunchKernelGGL((synth<NVALU, NLDS, DESC, FL>), dim3(blocks), dim3(256), 3 * 1025 * 8 + 64, 0, dev, nodes, mask, 10, lanes, 0.3, 0.5, 0.8, out);
            hipEventRecord(a);
            hipLaunchKernelGGL((synth<NVALU, NLDS, DESC, FL>), dim3(blocks), dim3(256), 3 * 1025 * 8 + 64, 0, dev, nodes, mask, steps, lanes, 0.3, 0.5, 0.8, out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double laneSteps = double(blocks) * 4 * lanes * steps;
            printf("%-34s lanes %2d waves/SIMD %d : %.3e lane-steps/s, %6.0f ns per wave-step\n", name, lanes, wavesPerSimd, laneSteps / (ms * 1e-3),
                   ms * 1e6 / steps);
        }
}

int main()
{
    const size_t records = size_t(1) << 22;  // 4 Mi records of 16 bytes = 64 MB (the walk's per-axis table: 46 MB)
    const unsigned mask = unsigned(records - 1);
    std::vector<uint4> host(records);
    std::vector<unsigned> hnodes(records);
    std::mt19937 rng(1);
    for (size_t r = 0; r < records; ++r)
    {
        host[r].x = rng();
        host[r].y = rng();
        host[r].z = rng();
        host[r].w = rng();
        hnodes[r] = rng();
    }
    uint4* dev;
    unsigned* nodes;
    double* out;
    hipMalloc(&dev, records * sizeof(uint4));
    hipMalloc(&nodes, records * sizeof(unsigned));
    hipMalloc(&out, size_t(256) * 8 * 256 * sizeof(double));
    hipMemcpy(dev, host.data(), records * sizeof(uint4), hipMemcpyHostToDevice);
    hipMemcpy(nodes, hnodes.data(), records * sizeof(unsigned), hipMemcpyHostToDevice);
    run<0, 0, 0>("chase only", dev, nodes, mask, out);
    run<0, 0, 0, 1>("chase only, nt loads", dev, nodes, mask, out);
    run<0, 0, 0, 2>("chase only, sc1 loads", dev, nodes, mask, out);
    run<0, 0, 0, 3>("chase only, sc0 sc1 loads", dev, nodes, mask, out);
    run<0, 0, 0, 4>("chase only, sc0 loads", dev, nodes, mask, out);
    run<0, 0, 0, 5>("chase only, sc0 sc1 nt loads", dev, nodes, mask, out);
    run<100, 6, 0, 1>("chase + arithmetic + LDS, nt", dev, nodes, mask, out);
    run<100, 6, 0, 2>("chase + arithmetic + LDS, sc1", dev, nodes, mask, out);
    run<0, 6, 0>("chase + 6 LDS reads", dev, nodes, mask, out);
    run<50, 0, 0>("chase + step arithmetic (1 refine)", dev, nodes, mask, out);
    run<100, 0, 0>("chase + step arithmetic (2 refine)", dev, nodes, mask, out);
    run<100, 6, 0>("chase + arithmetic + LDS", dev, nodes, mask, out);
    run<100, 6, 4>("  + 2nd load for 1.6 % of lanes", dev, nodes, mask, out);
    run<100, 6, 64>("  + 2nd load for 25 % of lanes", dev, nodes, mask, out);
    return 0;
}
