// bridge.hip -- round 4: from the gather microbenchmark (gather_knee.hip: 2.1e11 random 32-byte records/s) to the octree walk
// kernels (1.0e11 lane-steps/s, propagation; 1.4e11, peel-off), ONE ingredient of the real step at a time.  A synthetic walker
// that keeps the shape of walkPropKernel (pmc_walk_tree.inc): one walk per lane, the record of the next cell requested as soon as
// the link is known (two 16-byte loads of one 32-byte record), then the arithmetic that "enters" the cell; ingredients:
//   REAL    the scene's own CellRec table and its links (the walk goes from neighbour to neighbour through real links: the
//           table's own locality) instead of a random chase through a random table; walks start at the first cells the engine's
//           task records hold (the source's distribution) and end at the grid boundary
//   BODY    the f64 arithmetic of treeEnterBox (six wall differences, strict test, three correctly rounded quotients, tie order)
//           and of the step (position, optical depth), -ffp-contract=off; the exit axis it finds selects the next link
//   LDS     the six lane-divergent 8-byte reads of the coordinate table per step
//   ROUNDS  lanes whose walk has ended wait for a bookkeeping round (when `refill` lanes of a wave wait; checked every 8 steps)
//   IO      a round loads a task record (12 x 8 bytes from struct-of-arrays "slot" storage, one atomic per 256 slots) and stores
//           three results per finished walk; it also does the sampling arithmetic between the passes (exp, log) and 1 / k
//   TRIM    the first five segments of a walk leave 20 bytes each in LDS
// Built as a shared library (hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC) and driven by bridge.py, which
// hands over the device addresses of the engine's tables (pmc_debug_tables).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <algorithm>

namespace
{
    constexpr uint32_t LINK_NONE = 0x7FFFFFFFu, LINK_NODE = 0x80000000u, LINK_OCTET = 0x40000000u, LINK_INDEX = 0x1FFFFFFu;  // (link word of round 6: five bits of size exponent, 25 of index)
    enum : int { F_REAL = 1, F_BODY = 2, F_LDS = 4, F_ROUNDS = 8, F_IO = 16, F_TRIM = 32, F_OCT = 64 };

    struct Args
    {
        const uint4* table;       // 32-byte records (two uint4)
        uint32_t records;         // number of records (REAL) / mask + 1 (random chase: a power of two)
        const int32_t* starts;    // first cells of walks
        uint32_t numStarts;
        const double* slotIn;     // [12][slotCap]
        double* slotOut;          // [3][slotCap]
        uint32_t slotCap;
        unsigned long long* counters;  // [0] slot cursor, [1] lane-steps, [2] wave-steps, [3] rounds, [4] walks, [5] sink
        int steps;                // wave-steps per wave
        int refill;
        int meanLength;           // random chase: mean walk length (REAL: cap on the length)
        uint32_t tabEntries;      // coordinate table entries per axis (1025)
        uint32_t looseBase;       // F_OCT: first loose cell of the octet-line table (generation-8 experiment)
    };

    __device__ __forceinline__ uint32_t mix(uint32_t x)
    {
        x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
        return x;
    }
    __device__ __forceinline__ double exactQuotient(double d, double k, double y)
    {
        double q = d * y;
        double r = __builtin_fma(-k, q, d);
        q = __builtin_fma(r, y, q);
        r = __builtin_fma(-k, q, d);
        q = __builtin_fma(r, y, q);
        return fmin(q, 1.7976931348623157e308);
    }
    typedef __attribute__((address_space(3))) const double LdsDouble;
    __device__ __forceinline__ double ldsAt(uint32_t byteOffset) { return *reinterpret_cast<LdsDouble*>(static_cast<uintptr_t>(byteOffset)); }

    template<int F> __global__ __launch_bounds__(1024) void walk(const Args A)
    {
        extern __shared__ double lds[];
        const int tid = threadIdx.x, lane = tid & 63, block = blockDim.x;
        const uint32_t stride = A.tabEntries * 8u;
        for (uint32_t i = tid; i < 3u * A.tabEntries; i += block) lds[i] = (double)(mix(i) >> 8) * (1.0 / 16777216.0);
        __syncthreads();
        double* const trimTau = lds + 3 * A.tabEntries + tid;
        double* const trimS = trimTau + 5 * block;
        uint32_t* const trimCell = reinterpret_cast<uint32_t*>(lds + 3 * A.tabEntries + 10 * block) + tid;
        uint32_t rng = mix(blockIdx.x * block + tid + 1u);
        const double eps = 1e-9, sext = 0.37;
        uint4 ga = make_uint4(0, 0, 0, 0), gb = ga;
        uint32_t idx = 0, axis = 0, sgn = 0, left = 0, nrec = 0;
        double ds = 0., tau = 0., s = 0., rx = 0.3, ry = 0.4, rz = 0.5;
        double kx = 0.5, ky = 0.6, kz = 0.62, ikx = 2., iky = 1. / 0.6, ikz = 1. / 0.62;
        bool active = false;
        int slot = -1;
        unsigned long long poolNext = 0, poolEnd = 0;
        unsigned long long laneSteps = 0, waveSteps = 0, rounds = 0, walks = 0;
        double sinkD = 0.;
        uint32_t sinkU = 0;
        const unsigned long long below = (1ull << lane) - 1ull;

        auto sibl = [&](uint32_t cell, uint32_t ax) { return cell < A.looseBase && ((cell >> ax) & 1u) != (((sgn >> ax) & 1u) ^ 1u); };
        auto issue = [&](uint32_t cell) {
            if (F & F_OCT)
            {
                // octet-line table: 8 bytes of density now; the link (4 bytes of the same line) when the exit axis is known
                const char* line = reinterpret_cast<const char*>(A.table) + ((size_t)(cell >> 3) << 7);
                const uint2 v = *reinterpret_cast<const uint2*>(line + ((cell & 7u) << 3));
                ga.x = v.x, ga.y = v.y;
            }
            else
            {
                const uint4* p = A.table + 2ull * cell;
                ga = p[0];
                gb = p[1];
            }
        };
        auto issueLink = [&](uint32_t cell, uint32_t ax) {
            if ((F & F_OCT) && !sibl(cell, ax))
            {
                const char* line = reinterpret_cast<const char*>(A.table) + ((size_t)(cell >> 3) << 7);
                ga.z = *reinterpret_cast<const uint32_t*>(line + 64u + ((2u * ax + (((sgn >> ax) & 1u) ^ 1u)) << 2));
            }
        };
        // a new walk for this lane (no memory apart from the start list)
        auto restart = [&]() {
            rng = rng * 1664525u + 1013904223u;
            const uint32_t h = mix(rng);
            idx = (F & F_REAL) ? (uint32_t)A.starts[h % A.numStarts] : (h & (A.records - 1u));
            if ((F & F_REAL) && idx >= A.records) idx = h % A.records;
            sgn = (h >> 3) & 7u;
            axis = (h >> 7) % 3u;
            // walk length: geometric around meanLength (random chase) / cap (REAL)
            left = (F & F_REAL) ? (uint32_t)A.meanLength : 1u + (mix(h) % (2u * (uint32_t)A.meanLength));
            const double a = 0.3 + 0.6 * (double)((h >> 10) & 255u) * (1. / 256.), b = 0.3 + 0.6 * (double)((h >> 18) & 255u) * (1. / 256.);
            kx = (sgn & 1u) ? -a : a, ky = (sgn & 2u) ? -b : b, kz = (sgn & 4u) ? -0.55 : 0.55;
            tau = 0., s = 0., ds = 0.01, nrec = 0;
            active = true;
            walks += 1;
        };

        for (int it = 0; it < A.steps; it += 8)
        {
            // ---------------- round
            const unsigned long long act = __ballot(active);
            const int waiting = 64 - __popcll(act);
            if ((F & F_ROUNDS) ? waiting >= A.refill : waiting > 0)
            {
                rounds += 1;
                const bool want = !active;
                if (F & F_IO)
                {
                    // results of the finished walk; the sampling arithmetic between the passes
                    if (want && slot >= 0)
                    {
                        const double t = -log(1.0 - 0.37 * (1.0 - exp(-tau)));
                        A.slotOut[slot] = t;
                        A.slotOut[(size_t)A.slotCap + slot] = s;
                        A.slotOut[2 * (size_t)A.slotCap + slot] = tau;
                    }
                    // the next slot: a wave takes 256 at a time from the cursor
                    const unsigned long long idle = __ballot(want);
                    const int nidle = __popcll(idle);
                    if (poolNext + nidle > poolEnd)
                    {
                        unsigned long long got = 0;
                        if (lane == 0) got = atomicAdd(A.counters + 0, 256ull);
                        got = __shfl(got, 0, 64);
                        poolNext = got, poolEnd = got + 256;
                    }
                    if (want)
                    {
                        slot = (int)((poolNext + __popcll(idle & below)) % A.slotCap);
                        double v[12];
#pragma unroll
                        for (int j = 0; j < 12; ++j) v[j] = A.slotIn[(size_t)j * A.slotCap + slot];
                        restart();
                        // (position and direction of the slot, uniforms, first exit distance ...: used so that the loads stay)
                        rx = v[0], ry = v[1], rz = v[2];
                        kx += 1e-3 * v[3], ky += 1e-3 * v[4], kz += 1e-3 * v[5];
                        ds += 1e-6 * (v[6] + v[7] + v[8] + v[9] + v[10] + v[11]);
                        ikx = 1. / kx, iky = 1. / ky, ikz = 1. / kz;
                    }
                    poolNext += nidle;
                }
                else if (want)
                {
                    restart();
                    ikx = 1. / kx, iky = 1. / ky, ikz = 1. / kz;
                }
                if (want && active)
                {
                    issue(idx);
                    issueLink(idx, axis);
                }
            }
            // ---------------- steps
#pragma unroll 1
            for (int q = 0; q < 8; ++q)
            {
                laneSteps += (unsigned long long)__popcll(__ballot(active));
                waveSteps += 1;
                if (active)
                {
                    const double step = ds + eps;
                    const double nrx = rx + kx * step, nry = ry + ky * step, nrz = rz + kz * step;
                    // the record of the current cell: density and the link through the exit wall
                    const uint32_t fx = (sgn & 1u) ? ga.z : ga.w, fy = (sgn & 2u) ? gb.x : gb.y, fz = (sgn & 4u) ? gb.z : gb.w;
                    uint32_t link = axis == 0u ? fx : axis == 1u ? fy : fz;
                    if (F & F_OCT)
                    {
                        // a sibling needs no link; a same-size octet across the wall: the mirror child
                        const bool sib = sibl(idx, axis);
                        link = ga.z;
                        if (sib)
                            link = (idx ^ (1u << axis)) << 5;
                        else if (idx < A.looseBase && (link & 0xC0000000u) == LINK_OCTET && link != LINK_NONE)
                            link = (((link >> 5) & LINK_INDEX) | ((idx & 7u) ^ (1u << axis))) << 5;
                    }
                    const double dens = __longlong_as_double(((long long)ga.y << 32) | ga.x);
                    const double tau1 = tau + sext * dens * ds;
                    if ((F & F_TRIM) && nrec < 5u)
                    {
                        trimTau[nrec * block] = tau1;
                        trimS[nrec * block] = s + ds;
                        trimCell[nrec * block] = idx;
                    }
                    nrec += 1;
                    tau = tau1;
                    s += ds;
                    // the next cell
                    uint32_t next;
                    bool end = --left == 0u;
                    if (F & F_REAL)
                    {
                        if (link == LINK_NONE || (int32_t)link < 0)
                            end = true, next = 0;
                        else if (link & LINK_OCTET)
                            next = ((link >> 5) & LINK_INDEX) + (mix(link + nrec) & 7u);
                        else
                            next = link >> 5;
                        if (next >= A.records) end = true, next = 0;
                    }
                    else
                        next = ((link ^ ga.x) + nrec * 0x9E3779B1u) & (A.records - 1u);
                    if (end)
                        active = false;
                    else
                    {
                        issue(next);
                        idx = next;
                        if ((F & F_OCT) && !(F & F_BODY))
                        {
                            axis = (mix(next + nrec) >> 5) % 3u;
                            issueLink(next, axis);
                        }
                        if (F & F_BODY)
                        {
                            // treeEnterBox: box of the next cell (here: table offsets from a hash of its index), strict inside test,
                            // three exit distances, tie order
                            const uint32_t h = mix(next);
                            const uint32_t szb = 8u << (link & 3u);
                            const uint32_t ox = (h & 1016u) << 3, oy = (((h >> 10) & 1016u) << 3) + stride, oz = (((h >> 20) & 1016u) << 3) + 2u * stride;  // (entries 0 .. 1016, + szb / 8 <= 8)
                            double X0, X1, Y0, Y1, Z0, Z1;
                            if (F & F_LDS)
                            {
                                X0 = ldsAt(ox), X1 = ldsAt(ox + szb);
                                Y0 = ldsAt(oy), Y1 = ldsAt(oy + szb);
                                Z0 = ldsAt(oz), Z1 = ldsAt(oz + szb);
                            }
                            else
                            {
                                X0 = __hiloint2double(0x3fd00000 | (ox & 0xffff), h), X1 = X0 + 0.125;
                                Y0 = __hiloint2double(0x3fd00000 | (oy & 0xffff), h), Y1 = Y0 + 0.125;
                                Z0 = __hiloint2double(0x3fd00000 | (oz & 0xffff), h), Z1 = Z0 + 0.125;
                            }
                            const double fxr = nrx - floor(nrx), fyr = nry - floor(nry), fzr = nrz - floor(nrz);
                            const double x0 = X0 - fxr, x1 = X1 - fxr, y0 = Y0 - fyr, y1 = Y1 - fyr, z0 = Z0 - fzr, z1 = Z1 - fzr;
                            const double clear = fmin(fmin(fmin(x1, -x0), fmin(y1, -y0)), fmin(z1, -z0));
                            if (!(clear > 0.)) sinkU += 1;
                            const double ax = (sgn & 1u) ? x0 : x1, ay = (sgn & 2u) ? y0 : y1, az = (sgn & 4u) ? z0 : z1;
                            const double dsx = exactQuotient(ax, kx, ikx), dsy = exactQuotient(ay, ky, iky), dsz = exactQuotient(az, kz, ikz);
                            const double m = fmin(dsx, fmin(dsy, dsz));
                            axis = (dsx == m) ? 0u : (dsy == m) ? 1u : 2u;
                            ds = fabs(m) * 1e-3 + 1e-4;
                            rx = fxr, ry = fyr, rz = fzr;
                            issueLink(next, axis);
                        }
                        else if (!(F & F_OCT))
                        {
                            axis = (mix(next + nrec) >> 5) % 3u;
                            sinkD += nrx + nry + nrz;
                        }
                    }
                }
            }
        }
        sinkD += tau + s + rx + ry + rz;
        if (lane == 0)
        {
            atomicAdd(A.counters + 1, laneSteps);
            atomicAdd(A.counters + 2, waveSteps);
            atomicAdd(A.counters + 3, rounds);
        }
        atomicAdd(A.counters + 4, walks);
        if (sinkD == 1.2345 || sinkU == 0x12345678u) A.counters[5] = 1;
    }

    template<int F> int launch(const Args& a, int grid, int block, size_t ldsBytes, float* ms)
    {
        hipFuncSetAttribute(reinterpret_cast<const void*>(walk<F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        Args warm = a;
        warm.steps = 64;
        hipLaunchKernelGGL(walk<F>, dim3(grid), dim3(block), ldsBytes, 0, warm);
        hipMemset(a.counters, 0, 8 * sizeof(unsigned long long));
        hipEventRecord(e0);
        hipLaunchKernelGGL(walk<F>, dim3(grid), dim3(block), ldsBytes, 0, a);
        hipEventRecord(e1);
        if (hipEventSynchronize(e1) != hipSuccess) return 1;
        hipEventElapsedTime(ms, e0, e1);
        return hipGetLastError() == hipSuccess ? 0 : 1;
    }
}

// runs one variant; out[0] = ms, out[1] = lane-steps, out[2] = wave-steps, out[3] = rounds, out[4] = walks
extern "C" int bridge_run(int flags, const void* table, uint32_t records, const int32_t* starts, uint32_t numStarts, int grid, int block, int steps,
                          int refill, int meanLength, double* out, uint32_t looseBase)
{
    static double* slotIn = nullptr;
    static double* slotOut = nullptr;
    static unsigned long long* counters = nullptr;
    const uint32_t slotCap = 2800000u;
    if (!slotIn)
    {
        hipMalloc(&slotIn, size_t(12) * slotCap * sizeof(double));
        hipMalloc(&slotOut, size_t(3) * slotCap * sizeof(double));
        hipMalloc(&counters, 8 * sizeof(unsigned long long));
        hipMemset(slotIn, 0x3f, size_t(12) * slotCap * sizeof(double));  // (doubles around 4.7e-4)
    }
    Args a = {reinterpret_cast<const uint4*>(table), records, starts, numStarts, slotIn, slotOut, slotCap, counters, steps, refill, meanLength, 1025u, looseBase};
    const size_t ldsBytes = size_t(3) * 1025 * 8 + ((flags & F_TRIM) ? size_t(block) * 172 : 0);
    float ms = 0;
    int rc = 1;
    switch (flags)
    {
#define CASE(f) case f: rc = launch<f>(a, grid, block, ldsBytes, &ms); break;
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(6) CASE(7) CASE(8) CASE(9) CASE(14) CASE(15) CASE(24) CASE(25) CASE(30) CASE(31) CASE(62) CASE(63)
        CASE(10) CASE(11) CASE(27) CASE(59) CASE(65) CASE(71) CASE(73) CASE(127) CASE(95)
#undef CASE
        default: return 2;
    }
    if (rc) return rc;
    unsigned long long host[8];
    hipMemcpy(host, counters, sizeof(host), hipMemcpyDeviceToHost);
    out[0] = ms, out[1] = (double)host[1], out[2] = (double)host[2], out[3] = (double)host[3], out[4] = (double)host[4];
    return 0;
}

// the plain chase of gather_knee.hip on the same table (two 16-byte loads of a 32-byte record); returns records per second
namespace
{
    __global__ __launch_bounds__(1024) void plainChase(const uint4* __restrict__ table, unsigned mask, int steps, unsigned* out)
    {
        unsigned idx = ((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u) & mask;
        unsigned acc = 0;
#pragma unroll 1
        for (int i = 0; i < steps; ++i)
        {
            const uint4 a = table[2 * idx], b = table[2 * idx + 1];
            acc += a.y + b.w;
            idx = ((a.x ^ b.z) + i * 0x9E3779B1u) & mask;
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc + idx;
    }
}
// the plain chase with features of the walker added one at a time: V & 1: all eight words of the record are used (two full 16-byte
// loads); V & 2: the record's address is a 64-bit value in vector registers (as in the walk kernels) instead of scalar base + 32-bit
// offset; V & 4: the next index comes from ONE of six words picked by a per-lane axis / sign; V & 8: a lane pauses for the rest of
// an eight-step window after a random number of steps (partial exec masks)
namespace
{
    template<int V> __global__ __launch_bounds__(1024) void chaseVar(const uint4* __restrict__ table, unsigned mask, int steps, unsigned* out, const unsigned long long zero)
    {
        const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
        unsigned idx = (tid * 2654435761u) & mask;
        unsigned acc = 0, axis = tid % 3u, sgn = (tid >> 2) & 7u;
        const uint4* base = table;
        if (V & 2) base = table + (zero & tid);  // (zero at run time: the compiler must keep a per-lane 64-bit address)
        unsigned pause = 0;
#pragma unroll 1
        for (int i = 0; i < steps; ++i)
        {
            if ((V & 8) && (i & 7) == 0) pause = 0;
            if (!(V & 8) || !pause)
            {
                const uint4 a = base[2 * (size_t)idx], b = base[2 * (size_t)idx + 1];
                unsigned link = a.x ^ b.z;
                if (V & 1) acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w; else acc += a.y + b.w;
                if (V & 4)
                {
                    const unsigned fx = (sgn & 1u) ? a.z : a.w, fy = (sgn & 2u) ? b.x : b.y, fz = (sgn & 4u) ? b.z : b.w;
                    link = axis == 0u ? fx : axis == 1u ? fy : fz;
                    axis = (link >> 9) % 3u;
                }
                idx = (link + i * 0x9E3779B1u) & mask;
                if ((V & 8) && ((link >> 20) & 63u) == 0u) pause = 1;
            }
        }
        out[tid] = acc + idx;
    }
    template<int V> double runChaseVar(const void* table, uint32_t records, int grid, int block, int steps, unsigned* out)
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        hipLaunchKernelGGL(chaseVar<V>, dim3(grid), dim3(block), 0, 0, reinterpret_cast<const uint4*>(table), records - 1u, 10, out, 0ull);
        hipEventRecord(e0);
        hipLaunchKernelGGL(chaseVar<V>, dim3(grid), dim3(block), 0, 0, reinterpret_cast<const uint4*>(table), records - 1u, steps, out, 0ull);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        return double(grid) * block * steps / (ms * 1e-3);
    }
}
extern "C" double bridge_chase_variant(int v, const void* table, uint32_t records, int grid, int block, int steps)
{
    static unsigned* out = nullptr;
    if (!out) hipMalloc(&out, size_t(1024) * 1024 * sizeof(unsigned));
    switch (v)
    {
        case 0: return runChaseVar<0>(table, records, grid, block, steps, out);
        case 1: return runChaseVar<1>(table, records, grid, block, steps, out);
        case 2: return runChaseVar<2>(table, records, grid, block, steps, out);
        case 3: return runChaseVar<3>(table, records, grid, block, steps, out);
        case 5: return runChaseVar<5>(table, records, grid, block, steps, out);
        case 7: return runChaseVar<7>(table, records, grid, block, steps, out);
        case 9: return runChaseVar<9>(table, records, grid, block, steps, out);
        case 15: return runChaseVar<15>(table, records, grid, block, steps, out);
    }
    return 0.;
}

// The honest random gather: every lane owns its trajectory (a per-lane salt enters the next index), so lanes can never merge.
// LOADS = 1: one 16-byte load of a 16-byte record; 2: two 16-byte loads of a 32-byte record.  final[]: the last index of every lane.
namespace
{
    template<int LOADS, bool PRIVATE> __global__ __launch_bounds__(1024) void chaseTrue(const uint4* __restrict__ table, unsigned mask, int steps, unsigned* final)
    {
        const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
        unsigned idx = (tid * 2654435761u) & mask;
        const unsigned salt = PRIVATE ? tid * 0x85ebca6bu + 0x27d4eb2fu : 0u;
        unsigned acc = 0;
#pragma unroll 1
        for (int i = 0; i < steps; ++i)
        {
            unsigned link;
            if (LOADS == 1)
            {
                const uint4 a = table[idx];
                acc += a.y;
                link = a.x;
            }
            else
            {
                const uint4 a = table[2 * (size_t)idx], b = table[2 * (size_t)idx + 1];
                acc += a.y + b.w;
                link = a.x ^ b.z;
            }
            idx = ((link ^ salt) + i * 0x9E3779B1u) & mask;
        }
        final[tid] = idx + (acc & 0u);
    }
}
// returns records per second; *distinct = number of different final indices among the grid * block lanes
extern "C" double bridge_true_gather(int loads, int priv, const void* table, uint32_t records, int grid, int block, int steps, double* distinct)
{
    static unsigned* out = nullptr;
    if (!out) hipMalloc(&out, size_t(2048) * 1024 * sizeof(unsigned));
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const uint4* t = reinterpret_cast<const uint4*>(table);
    auto go = [&](int n) {
        if (loads == 1 && priv) hipLaunchKernelGGL((chaseTrue<1, true>), dim3(grid), dim3(block), 0, 0, t, records - 1u, n, out);
        if (loads == 1 && !priv) hipLaunchKernelGGL((chaseTrue<1, false>), dim3(grid), dim3(block), 0, 0, t, records - 1u, n, out);
        if (loads == 2 && priv) hipLaunchKernelGGL((chaseTrue<2, true>), dim3(grid), dim3(block), 0, 0, t, records - 1u, n, out);
        if (loads == 2 && !priv) hipLaunchKernelGGL((chaseTrue<2, false>), dim3(grid), dim3(block), 0, 0, t, records - 1u, n, out);
    };
    go(10);
    hipEventRecord(e0);
    go(steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (distinct)
    {
        const size_t n = size_t(grid) * block;
        unsigned* host = new unsigned[n];
        hipMemcpy(host, out, n * sizeof(unsigned), hipMemcpyDeviceToHost);
        std::sort(host, host + n);
        *distinct = double(std::unique(host, host + n) - host);
        delete[] host;
    }
    return double(grid) * block * steps / (ms * 1e-3);
}

// The CU's own limit: private trajectories (no merging) on FEW CUs, so that L2 and fabric are far from their limits; what one
// lane-step asks of the vector memory pipeline is varied.  SHAPE 0: one 16-byte load; 1: two 16-byte loads of one 32-byte record;
// 2: an 8-byte and a 4-byte load of one 128-byte line (8 B at 8 j, 4 B at 64 + 4 w); 3: 8 + 16 + 8 bytes of one line; 4: two 16-byte
// loads of two different lines; 5: one 8-byte load; 6: as 1 with a scalar base and 32-bit vector offsets
namespace
{
    template<int SHAPE> __global__ __launch_bounds__(1024) void chaseShape(const char* __restrict__ table, unsigned mask, int steps, unsigned* final, const unsigned long long zero)
    {
        const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
        unsigned idx = (tid * 2654435761u) & mask;  // index of a 128-byte line
        const unsigned salt = tid * 0x85ebca6bu + 0x27d4eb2fu;
        unsigned acc = 0;
        const char* base = table + (zero & tid);  // (a per-lane 64-bit address, as in the walk kernels)
#pragma unroll 1
        for (int i = 0; i < steps; ++i)
        {
            unsigned link;
            const unsigned sub = (idx ^ i) & 7u;
            if (SHAPE == 0)
            {
                const uint4 a = *reinterpret_cast<const uint4*>(base + ((size_t)idx << 7) + ((sub & 3u) << 5));
                link = a.x ^ a.w, acc += a.y;
            }
            else if (SHAPE == 1)
            {
                const char* p = base + ((size_t)idx << 7) + ((sub & 3u) << 5);
                const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 16);
                link = a.x ^ b.z, acc += a.y + b.w;
            }
            else if (SHAPE == 2)
            {
                const char* p = base + ((size_t)idx << 7);
                const uint2 a = *reinterpret_cast<const uint2*>(p + (sub << 3));
                const unsigned b = *reinterpret_cast<const unsigned*>(p + 64 + ((sub % 6u) << 2));
                link = a.x ^ b, acc += a.y;
            }
            else if (SHAPE == 3)
            {
                const char* p = base + ((size_t)idx << 7);
                const uint2 a = *reinterpret_cast<const uint2*>(p + (sub << 3));
                const uint4 b = *reinterpret_cast<const uint4*>(p + 64);
                const uint2 c = *reinterpret_cast<const uint2*>(p + 80);
                link = a.x ^ b.z ^ c.y, acc += a.y + b.x + c.x;
            }
            else if (SHAPE == 4)
            {
                const uint4 a = *reinterpret_cast<const uint4*>(base + ((size_t)idx << 7));
                const uint4 b = *reinterpret_cast<const uint4*>(base + ((size_t)((idx * 2654435761u + 12345u) & mask) << 7));
                link = a.x ^ b.z, acc += a.y + b.w;
            }
            else if (SHAPE == 5)
            {
                const uint2 a = *reinterpret_cast<const uint2*>(base + ((size_t)idx << 7) + (sub << 3));
                link = a.x, acc += a.y;
            }
            else
            {
                const unsigned off = (idx << 7) + ((sub & 3u) << 5);
                const uint4 a = *reinterpret_cast<const uint4*>(table + off), b = *reinterpret_cast<const uint4*>(table + off + 16);
                link = a.x ^ b.z, acc += a.y + b.w;
            }
            idx = ((link ^ salt) + i * 0x9E3779B1u) & mask;
        }
        final[tid] = idx + (acc & 0u);
    }
}
extern "C" double bridge_shape(int shape, const void* table, uint32_t lines, int grid, int block, int steps)
{
    static unsigned* out = nullptr;
    if (!out) hipMalloc(&out, size_t(2048) * 1024 * sizeof(unsigned));
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const char* t = reinterpret_cast<const char*>(table);
    auto go = [&](int n) {
        switch (shape)
        {
            case 0: hipLaunchKernelGGL(chaseShape<0>, dim3(grid), dim3(block), 0, 0, t, lines - 1u, n, out, 0ull); break;
            case 1: hipLaunchKernelGGL(chaseShape<1>, dim3(grid), dim3(block), 0, 0, t, lines - 1u, n, out, 0ull); break;
            case 2: hipLaunchKernelGGL(chaseShape<2>, dim3(grid), dim3(block), 0, 0, t, lines - 1u, n, out, 0ull); break;
            case 3: hipLaunchKernelGGL(chaseShape<3>, dim3(grid), dim3(block), 0, 0, t, lines - 1u, n, out, 0ull); break;
            case 4: hipLaunchKernelGGL(chaseShape<4>, dim3(grid), dim3(block), 0, 0, t, lines - 1u, n, out, 0ull); break;
            case 5: hipLaunchKernelGGL(chaseShape<5>, dim3(grid), dim3(block), 0, 0, t, lines - 1u, n, out, 0ull); break;
            default: hipLaunchKernelGGL(chaseShape<6>, dim3(grid), dim3(block), 0, 0, t, lines - 1u, n, out, 0ull); break;
        }
    };
    go(10);
    hipEventRecord(e0);
    go(steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return double(grid) * block * steps / (ms * 1e-3);
}

extern "C" double bridge_plain_chase(const void* table, uint32_t records, int grid, int block, int steps)
{
    static unsigned* out = nullptr;
    if (!out) hipMalloc(&out, size_t(1024) * 1024 * sizeof(unsigned));
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(plainChase, dim3(grid), dim3(block), 0, 0, reinterpret_cast<const uint4*>(table), records - 1u, 10, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(plainChase, dim3(grid), dim3(block), 0, 0, reinterpret_cast<const uint4*>(table), records - 1u, steps, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return double(grid) * block * steps / (ms * 1e-3);
}

// a random table for the random chase (records: a power of two), on the device
extern "C" void* bridge_random_table(uint32_t records)
{
    uint4* host = new uint4[2ull * records];
    uint32_t x = 12345u;
    for (size_t i = 0; i < 2ull * records; ++i)
    {
        auto next = [&]() { x ^= x << 13, x ^= x >> 17, x ^= x << 5; return x; };
        host[i] = make_uint4(next(), next() & 0x3fefffffu, next(), next());
    }
    void* dev = nullptr;
    hipMalloc(&dev, 2ull * records * sizeof(uint4));
    hipMemcpy(dev, host, 2ull * records * sizeof(uint4), hipMemcpyHostToDevice);
    delete[] host;
    return dev;
}
