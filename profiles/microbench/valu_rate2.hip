// valu_rate2.hip -- like valu_rate.hip, at 1, 2, 4 and 8 waves per SIMD (aggregate issue rate of a SIMD): is the step's
// cost its instruction count at any occupancy?
// valu_rate.hip -- issue cost (SIMD cycles per wave64 instruction) of the VALU operations the walk step is made of.
// One wave per SIMD (grid = 4 waves per CU, one workgroup of 256 per CU), 8 independent chains per operation so that
// the dependent-issue latency does not limit the rate; cycles from s_memtime (100 MHz constant clock is NOT used:
// clock64() = s_memtime counts shader clocks on gfx950).   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 512
#define CHAINS 8

template<int OP> __device__ __forceinline__ void op(double& a, double b, double c, int& i, int j)
{
    if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == 3) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == 4) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
    if (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i) : "v"(j) : "vcc");
    if (OP == 6) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i) : "v"(a));
    if (OP == 7) asm volatile("v_and_b32 %0, %0, %1" : "+v"(i) : "v"(j));
    if (OP == 8) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(i) : "v"(j));
    if (OP == 9) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(i) : "v"(j));
    if (OP == 10) asm volatile("v_rcp_f64 %0, %0" : "+v"(a));
    if (OP == 11) asm volatile("v_add_u32 %0, %0, %1" : "+v"(i) : "v"(j));
    if (OP == 12) asm volatile("v_cmp_lt_f64 s[20:21], %0, %1" : : "v"(a), "v"(b) : "s20", "s21");
    if (OP == 13) asm volatile("v_bfe_u32 %0, %0, %1, 1" : "+v"(i) : "v"(j));
    if (OP == 14) asm volatile("v_mov_b32 %0, %1" : "=v"(i) : "v"(j));
    if (OP == 15) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a) : "v"(j));
    if (OP == 16) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(i) : "v"(j));
    if (OP == 17) asm volatile("v_fract_f64 %0, %0" : "+v"(a));
    if (OP == 18) asm volatile("v_floor_f64 %0, %0" : "+v"(a));
    if (OP == 19) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i) : "v"(j));
}

template<int OP> __global__ __launch_bounds__(256) void rate(double* out, long long* cycles, double b, double c, int j)
{
    double a[CHAINS];
    int i[CHAINS];
    for (int q = 0; q < CHAINS; ++q) a[q] = 1.0 + threadIdx.x * 1e-9 + q, i[q] = threadIdx.x + q;
    long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < ITER; ++it)
    {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < CHAINS; ++q) op<OP>(a[q], b, c, i[q], j);
    }
    long long t1 = clock64();
    double s = 0;
    for (int q = 0; q < CHAINS; ++q) s += a[q] + i[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template<int OP> void run(const char* name, double* out, long long* cyc, int cus)
{
    printf("%-18s", name);
    for (int w : {1, 2, 4, 8})
    {
        const int blocks = cus * w;  // w workgroups of 256 per CU = w waves per SIMD
        hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0000001, 1e-9, 3);
        hipDeviceSynchronize();
        std::vector<long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += v;
        mean /= blocks;
        // SIMD cycles per wave64 instruction = elapsed cycles of a wave / (instructions per wave * waves per SIMD)
        printf("  w=%d: %5.2f", w, mean / (ITER * 4.0 * CHAINS * w));
    }
    printf("\n");
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int blocks = p.multiProcessorCount;
    double* out;
    long long* cyc;
    hipMalloc(&out, 8 * blocks * 256 * sizeof(double));
    hipMalloc(&cyc, 8 * blocks * sizeof(long long));
    printf("%s, %d CUs; SIMD cycles (clock64 ticks) per wave64 instruction at w waves per SIMD\n", p.name, blocks);
    run<9>("v_fma_f32", out, cyc, blocks);
    run<0>("v_fma_f64", out, cyc, blocks);
    run<1>("v_mul_f64", out, cyc, blocks);
    run<2>("v_add_f64", out, cyc, blocks);
    run<3>("v_min_f64", out, cyc, blocks);
    run<4>("v_cmp_lt_f64 vcc", out, cyc, blocks);
    run<12>("v_cmp_lt_f64 sgpr", out, cyc, blocks);
    run<5>("v_cndmask_b32", out, cyc, blocks);
    run<6>("v_cvt_i32_f64", out, cyc, blocks);
    run<15>("v_cvt_f64_i32", out, cyc, blocks);
    run<10>("v_rcp_f64", out, cyc, blocks);
    run<17>("v_fract_f64", out, cyc, blocks);
    run<18>("v_floor_f64", out, cyc, blocks);
    run<7>("v_and_b32", out, cyc, blocks);
    run<8>("v_lshlrev_b32", out, cyc, blocks);
    run<11>("v_add_u32", out, cyc, blocks);
    run<13>("v_bfe_u32", out, cyc, blocks);
    run<14>("v_mov_b32", out, cyc, blocks);
    run<16>("v_lshl_add_u32", out, cyc, blocks);
    run<19>("v_mul_lo_u32", out, cyc, blocks);
    return 0;
}
