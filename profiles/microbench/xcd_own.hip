// xcd_own.hip -- round 6: does SPATIAL OWNERSHIP of the hot cell table per XCD lift the rate of the propagation walks?
// (round-5 review, item 1a.)  The synthetic walker of bridge.hip (one walk per lane on the scene's own CellRec table with its real links,
// the f64 body of the step, six LDS reads, rounds at `refill` waiting lanes, task loads / result stores) in two modes:
//   MODE 0  as bridge.hip: every workgroup walks anywhere in the table
//   MODE 2  the upper bound of MODE 1: the same ownership with hand-overs that cost NOTHING -- a walk that leaves its XCD's eighth is dropped
//           (and counted), its lane takes a fresh start in its own eighth at the next round: no queue, no record, no atomics
//   MODE 1  a workgroup reads HW_REG_XCC_ID = x and only ever touches cells of the x-th eighth of the depth-first table (its XCD's L2
//           then sees 3.8 MB of the 30.5 MB); a walk whose next cell belongs to another eighth is HANDED OVER: the lane writes a 128-byte
//           state record into the target XCD's queue (write-through `sc0 sc1` stores, a flag word per record behind `s_waitcnt vmcnt(0)`)
//           and waits for the next round; a round first takes handed-over walks from its own XCD's queue (a CAS on the queue's head by
//           lane 0, `sc0 sc1` loads), then fresh walks that START in its eighth (start list binned by the host)
// Reports lane-steps/s, lanes in use, hand-overs per walk and the share of the lane-steps per XCD (load balance).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC profiles/microbench/xcd_own.hip -o profiles/microbench/libxcdown.so
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

namespace
{
    constexpr uint32_t LINK_NONE = 0x7FFFFFFFu, LINK_OCTET = 0x40000000u, LINK_INDEX = 0x1FFFFFFu;  // (link word of round 6: five bits of size exponent, 25 of index)

    struct Args
    {
        const uint4* table;
        uint32_t records;
        const int32_t* starts;  // binned by owner
        uint32_t binOffset[9];
        uint4* queue;           // [8][qcap][8]
        uint32_t* flags;        // [8][qcap]
        unsigned long long* qctl;  // head of x at [32 x], tail at [32 x + 16]
        uint32_t qcap;
        const double* slotIn;   // [12][slotCap]
        double* slotOut;        // [3][slotCap]
        uint32_t slotCap;
        unsigned long long* counters;  // [0] slot cursor, [1] lane-steps, [2] wave-steps, [3] rounds, [4] walks, [5] sink, [6] hand-overs, [7] failed claims, [8 + x] lane-steps of XCD x, [16 + x] workgroups on x
        int steps, refill;
        uint32_t capLen, tabEntries;
        uint32_t ownerMagic;    // owner(cell) = min(7, umulhi(cell, ownerMagic))
        unsigned long long zero;
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

    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __device__ __forceinline__ void storeWT(uint4* p, uint4 w)
    {
        const u32x4 v = {w.x, w.y, w.z, w.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    }
    __device__ __forceinline__ void storeWT32(uint32_t* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
    __device__ __forceinline__ uint4 loadWT(const uint4* p)
    {
        u32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    __device__ __forceinline__ uint32_t loadWT32(const uint32_t* p)
    {
        uint32_t v;
        asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }

    template<int MODE> __global__ __launch_bounds__(1024) void walkOwn(const Args A)
    {
        extern __shared__ double lds[];
        const int tid = threadIdx.x, lane = tid & 63, block = blockDim.x;
        const uint32_t stride = A.tabEntries * 8u;
        for (uint32_t i = tid; i < 3u * A.tabEntries; i += block) lds[i] = (double)(mix(i) >> 8) * (1.0 / 16777216.0);
        __syncthreads();
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        if (tid == 0) atomicAdd(A.counters + 16 + xcc, 1ull);
        const uint32_t binLo = MODE ? A.binOffset[xcc] : 0u, binN = MODE ? A.binOffset[xcc + 1] - binLo : A.binOffset[8];
        uint32_t rng = mix(blockIdx.x * block + tid + 1u);
        const double eps = 1e-9, sext = 0.37;
        uint4 ga = make_uint4(0, 0, 0, 0), gb = ga;
        uint32_t idx = 0, axis = 0, sgn = 0, left = 0, nrec = 0;
        double ds = 0., tau = 0., s = 0., rx = 0.3, ry = 0.4, rz = 0.5;
        double kx = 0.5, ky = 0.6, kz = 0.62, ikx = 2., iky = 1. / 0.6, ikz = 1. / 0.62;
        bool active = false, pend = false;
        uint32_t pendTo = 0;
        int slot = -1;
        unsigned long long poolNext = 0, poolEnd = 0;
        unsigned long long laneSteps = 0, waveSteps = 0, rounds = 0, walks = 0, handed = 0, failed = 0;
        double sinkD = 0.;
        uint32_t sinkU = 0;
        const unsigned long long below = (1ull << lane) - 1ull;

        auto owner = [&](uint32_t cell) { return min(7u, __umulhi(cell, A.ownerMagic)); };
        auto issue = [&](uint32_t cell) {
            const uint4* p = A.table + 2ull * cell;
            ga = p[0];
            gb = p[1];
        };

        for (int it = 0; it < A.steps; it += 8)
        {
            const unsigned long long act = __ballot(active);
            const int waiting = 64 - __popcll(act);
            if (waiting >= A.refill)
            {
                rounds += 1;
                // ---- hand-overs of this wave, one claim per target queue
                if (MODE == 2)
                {
                    if (pend) handed += 1;
                    pend = false;
                }
                if (MODE == 1)
                {
                    unsigned long long pm = __ballot(pend);
                    if (pm)
                    {
                        for (uint32_t t = 0; t < 8u; ++t)
                        {
                            const unsigned long long m = __ballot(pend && pendTo == t);
                            if (!m) continue;
                            unsigned long long base = 0;
                            if (lane == __ffsll((long long)m) - 1) base = atomicAdd(A.qctl + 32 * t + 16, (unsigned long long)__popcll(m));
                            base = __shfl(base, __ffsll((long long)m) - 1, 64);
                            if (pend && pendTo == t)
                            {
                                const unsigned long long ticket = base + __popcll(m & below);
                                const uint32_t at = (uint32_t)ticket & (A.qcap - 1u);
                                uint4* rec = A.queue + ((size_t)t * A.qcap + at) * 8u;
                                storeWT(rec + 0, make_uint4(idx, axis | (sgn << 2) | (nrec << 8), left, (uint32_t)slot));
                                storeWT(rec + 1, make_uint4(__double2loint(rx), __double2hiint(rx), __double2loint(ry), __double2hiint(ry)));
                                storeWT(rec + 2, make_uint4(__double2loint(rz), __double2hiint(rz), __double2loint(kx), __double2hiint(kx)));
                                storeWT(rec + 3, make_uint4(__double2loint(ky), __double2hiint(ky), __double2loint(kz), __double2hiint(kz)));
                                storeWT(rec + 4, make_uint4(__double2loint(tau), __double2hiint(tau), __double2loint(s), __double2hiint(s)));
                                storeWT(rec + 5, make_uint4(__double2loint(ds), __double2hiint(ds), 0, 0));
                                storeWT(rec + 6, make_uint4(1, 2, 3, 4));
                                storeWT(rec + 7, make_uint4(5, 6, 7, 8));
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                storeWT32(A.flags + (size_t)t * A.qcap + at, (uint32_t)(ticket / A.qcap) + 1u);
                                handed += 1;
                            }
                        }
                        pend = false;
                    }
                }
                const bool want = !active;
                // ---- handed-over walks of this XCD first
                unsigned long long took = 0;
                if (MODE == 1)
                {
                    const unsigned long long idle = __ballot(want);
                    const int nidle = __popcll(idle);
                    unsigned long long h = 0;
                    int take = 0;
                    if (lane == 0)
                    {
                        // (read-modify-write reads: a relaxed agent-scope LOAD is served by this XCD's L2 and returned a stale tail -- the first
                        // version of this benchmark left 1.2e7 handed-over walks in the queues; atomics execute at the memory side)
                        // (A.zero is 0 at run time: `atomicAdd(p, 0)` with a literal is folded into an atomic LOAD by the compiler -- and stays stale)
                        h = atomicAdd(A.qctl + 32 * xcc, A.zero);
                        const unsigned long long t = atomicAdd(A.qctl + 32 * xcc + 16, A.zero);
                        take = t > h ? (int)min((unsigned long long)nidle, t - h) : 0;
                        if (take && atomicCAS(A.qctl + 32 * xcc, h, h + take) != h) take = 0, failed += 1;
                    }
                    take = __shfl(take, 0, 64);
                    h = __shfl(h, 0, 64);
                    const int rank = __popcll(idle & below);
                    if (want && rank < take)
                    {
                        const unsigned long long ticket = h + rank;
                        const uint32_t at = (uint32_t)ticket & (A.qcap - 1u);
                        const uint32_t expect = (uint32_t)(ticket / A.qcap) + 1u;
                        int polls = 0;
                        while (loadWT32(A.flags + (size_t)xcc * A.qcap + at) != expect && ++polls < 2000) __builtin_amdgcn_s_sleep(1);
                        if (polls >= 2000) failed += 1000000ull;  // (a record that never arrived: reported, the walk is taken as it is)
                        const uint4* rec = A.queue + ((size_t)xcc * A.qcap + at) * 8u;
                        u32x4 a, b, c, d, e, f, g, hh;
                        asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1\n\tglobal_load_dwordx4 %1, %8, off offset:16 sc0 sc1\n\t"
                                     "global_load_dwordx4 %2, %8, off offset:32 sc0 sc1\n\tglobal_load_dwordx4 %3, %8, off offset:48 sc0 sc1\n\t"
                                     "global_load_dwordx4 %4, %8, off offset:64 sc0 sc1\n\tglobal_load_dwordx4 %5, %8, off offset:80 sc0 sc1\n\t"
                                     "global_load_dwordx4 %6, %8, off offset:96 sc0 sc1\n\tglobal_load_dwordx4 %7, %8, off offset:112 sc0 sc1\n\t"
                                     "s_waitcnt vmcnt(0)"
                                     : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e), "=&v"(f), "=&v"(g), "=&v"(hh)
                                     : "v"(rec)
                                     : "memory");
                        idx = a.x, axis = a.y & 3u, sgn = (a.y >> 2) & 7u, nrec = a.y >> 8, left = a.z, slot = (int)a.w;
                        rx = __hiloint2double(b.y, b.x), ry = __hiloint2double(b.w, b.z), rz = __hiloint2double(c.y, c.x);
                        kx = __hiloint2double(c.w, c.z), ky = __hiloint2double(d.y, d.x), kz = __hiloint2double(d.w, d.z);
                        tau = __hiloint2double(e.y, e.x), s = __hiloint2double(e.w, e.z), ds = __hiloint2double(f.y, f.x);
                        sinkU += g.x + hh.w;
                        ikx = 1. / kx, iky = 1. / ky, ikz = 1. / kz;
                        active = true;
                        issue(idx);
                    }
                    took = __ballot(want && rank < take);
                }
                const bool fresh = want && !active;
                // results of the finished walk; the sampling arithmetic between the passes
                if (fresh && slot >= 0)
                {
                    const double t = -log(1.0 - 0.37 * (1.0 - exp(-tau)));
                    A.slotOut[slot] = t;
                    A.slotOut[(size_t)A.slotCap + slot] = s;
                    A.slotOut[2 * (size_t)A.slotCap + slot] = tau;
                }
                const unsigned long long idle = __ballot(fresh);
                const int nidle = __popcll(idle);
                if (nidle && poolNext + nidle > poolEnd)
                {
                    unsigned long long got = 0;
                    if (lane == 0) got = atomicAdd(A.counters + 0, 256ull);
                    got = __shfl(got, 0, 64);
                    poolNext = got, poolEnd = got + 256;
                }
                if (fresh)
                {
                    slot = (int)((poolNext + __popcll(idle & below)) % A.slotCap);
                    double v[12];
#pragma unroll
                    for (int j = 0; j < 12; ++j) v[j] = A.slotIn[(size_t)j * A.slotCap + slot];
                    rng = rng * 1664525u + 1013904223u;
                    const uint32_t h = mix(rng);
                    idx = (uint32_t)A.starts[binLo + h % binN];
                    if (idx >= A.records) idx = h % A.records;
                    sgn = (h >> 3) & 7u;
                    axis = (h >> 7) % 3u;
                    left = A.capLen;
                    const double a = 0.3 + 0.6 * (double)((h >> 10) & 255u) * (1. / 256.), b = 0.3 + 0.6 * (double)((h >> 18) & 255u) * (1. / 256.);
                    kx = (sgn & 1u) ? -a : a, ky = (sgn & 2u) ? -b : b, kz = (sgn & 4u) ? -0.55 : 0.55;
                    tau = 0., s = 0., ds = 0.01, nrec = 0;
                    active = true;
                    walks += 1;
                    rx = v[0], ry = v[1], rz = v[2];
                    kx += 1e-3 * v[3], ky += 1e-3 * v[4], kz += 1e-3 * v[5];
                    ds += 1e-6 * (v[6] + v[7] + v[8] + v[9] + v[10] + v[11]);
                    ikx = 1. / kx, iky = 1. / ky, ikz = 1. / kz;
                    issue(idx);
                }
                poolNext += nidle;
                (void)took;
            }
#pragma unroll 1
            for (int q = 0; q < 8; ++q)
            {
                laneSteps += (unsigned long long)__popcll(__ballot(active));
                waveSteps += 1;
                if (active)
                {
                    const double step = ds + eps;
                    const double nrx = rx + kx * step, nry = ry + ky * step, nrz = rz + kz * step;
                    const uint32_t fx = (sgn & 1u) ? ga.z : ga.w, fy = (sgn & 2u) ? gb.x : gb.y, fz = (sgn & 4u) ? gb.z : gb.w;
                    const uint32_t link = axis == 0u ? fx : axis == 1u ? fy : fz;
                    const double dens = __longlong_as_double(((long long)ga.y << 32) | ga.x);
                    tau += sext * dens * ds;
                    nrec += 1;
                    s += ds;
                    uint32_t next;
                    bool end = --left == 0u;
                    if (link == LINK_NONE || (int32_t)link < 0)
                        end = true, next = 0;
                    else if (link & LINK_OCTET)
                        next = ((link >> 5) & LINK_INDEX) + (mix(link + nrec) & 7u);
                    else
                        next = link >> 5;
                    if (next >= A.records) end = true, next = 0;
                    if (end)
                        active = false;
                    else
                    {
                        const uint32_t h = mix(next);
                        const bool away = MODE && owner(next) != xcc;
                        if (!away) issue(next);
                        idx = next;
                        const uint32_t szb = 8u << (link & 3u);
                        const uint32_t ox = (h & 1016u) << 3, oy = (((h >> 10) & 1016u) << 3) + stride, oz = (((h >> 20) & 1016u) << 3) + 2u * stride;
                        const double X0 = ldsAt(ox), X1 = ldsAt(ox + szb), Y0 = ldsAt(oy), Y1 = ldsAt(oy + szb), Z0 = ldsAt(oz), Z1 = ldsAt(oz + szb);
                        const double fxr = nrx - floor(nrx), fyr = nry - floor(nry), fzr = nrz - floor(nrz);
                        const double x0 = X0 - fxr, x1 = X1 - fxr, y0 = Y0 - fyr, y1 = Y1 - fyr, z0 = Z0 - fzr, z1 = Z1 - fzr;
                        const double clear = fmin(fmin(fmin(x1, -x0), fmin(y1, -y0)), fmin(z1, -z0));
                        if (!(clear > 0.)) sinkU += 1;
                        const double ax = (sgn & 1u) ? x0 : x1, ay = (sgn & 2u) ? y0 : y1, az = (sgn & 4u) ? z0 : z1;
                        const double dsx = exactQuotient(ax, kx, ikx), dsy = exactQuotient(ay, ky, iky), dsz = exactQuotient(az, kz, ikz);
                        const double m = fmin(dsx, fmin(dsy, dsz));
                        axis = (dsx == m) ? 0u : (dsy == m) ? 1u : 2u;
                        // (the synthetic walk picks its axis at random: the table's links, not the box arithmetic, decide where it goes)
                        axis = (axis + (mix(next + nrec) >> 5)) % 3u;
                        ds = fabs(m) * 1e-3 + 1e-4;
                        rx = fxr, ry = fyr, rz = fzr;
                        if (away) active = false, pend = true, pendTo = owner(next);
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
            atomicAdd(A.counters + 8 + xcc, laneSteps);
        }
        atomicAdd(A.counters + 4, walks);
        if (handed) atomicAdd(A.counters + 6, handed);
        if (failed) atomicAdd(A.counters + 7, failed);
        if (sinkD == 1.2345 || sinkU == 0x12345678u) A.counters[5] = 1;
    }
}

// out: [0] ms, [1] lane-steps, [2] wave-steps, [3] rounds, [4] walks, [5] hand-overs, [6] failed claims, [7] queue entries left, [8..15] lane-steps per XCD,
// [16..23] workgroups per XCD, [24..31] fresh starts per bin
extern "C" int xcdown_run(int mode, const void* table, uint32_t records, const int32_t* startsDev, uint32_t numStarts, int grid, int block, int steps, int refill,
                          int capLen, double* out)
{
    static double* slotIn = nullptr;
    static double* slotOut = nullptr;
    static unsigned long long* counters = nullptr;
    static uint4* queue = nullptr;
    static uint32_t* flags = nullptr;
    static unsigned long long* qctl = nullptr;
    static int32_t* binned = nullptr;
    static uint32_t binOffset[9];
    const uint32_t slotCap = 2800000u, qcap = 1u << 21;
    const uint32_t magic = (uint32_t)(((8ull << 32) + records - 1) / records);
    if (!slotIn)
    {
        hipMalloc(&slotIn, size_t(12) * slotCap * sizeof(double));
        hipMalloc(&slotOut, size_t(3) * slotCap * sizeof(double));
        hipMalloc(&counters, 32 * sizeof(unsigned long long));
        hipMemset(slotIn, 0x3f, size_t(12) * slotCap * sizeof(double));
        hipMalloc(&queue, size_t(8) * qcap * 128);
        hipMalloc(&flags, size_t(8) * qcap * 4);
        hipMalloc(&qctl, 256 * sizeof(unsigned long long));
        std::vector<int32_t> host(numStarts), sorted(numStarts);
        if (startsDev)
            hipMemcpy(host.data(), startsDev, size_t(numStarts) * 4, hipMemcpyDeviceToHost);
        else
        {
            // (no task records at hand -- a counter pass under rocprofv3 --: walks start in cells drawn uniformly from the table)
            uint32_t x = 2463534242u;
            for (auto& c : host)
            {
                x ^= x << 13, x ^= x >> 17, x ^= x << 5;
                c = (int32_t)(x % records);
            }
        }
        uint32_t count[9] = {0};
        auto own = [&](int32_t c) { uint32_t u = (uint32_t)c; if (u >= records) u = 0; uint32_t o = (uint32_t)(((unsigned long long)u * magic) >> 32); return o > 7u ? 7u : o; };
        for (auto c : host) count[own(c) + 1] += 1;
        for (int i = 0; i < 8; ++i) count[i + 1] += count[i];
        for (int i = 0; i < 9; ++i) binOffset[i] = count[i];
        uint32_t cur[8];
        for (int i = 0; i < 8; ++i) cur[i] = count[i];
        for (auto c : host) sorted[cur[own(c)]++] = c;
        hipMalloc(&binned, size_t(numStarts) * 4);
        hipMemcpy(binned, sorted.data(), size_t(numStarts) * 4, hipMemcpyHostToDevice);
    }
    for (int i = 0; i < 8; ++i) out[24 + i] = binOffset[i + 1] - binOffset[i];
    for (int i = 0; i < 8; ++i)
        if (binOffset[i + 1] == binOffset[i]) return 3;
    Args a;
    a.table = reinterpret_cast<const uint4*>(table), a.records = records, a.starts = binned;
    for (int i = 0; i < 9; ++i) a.binOffset[i] = binOffset[i];
    a.queue = queue, a.flags = flags, a.qctl = qctl, a.qcap = qcap, a.slotIn = slotIn, a.slotOut = slotOut, a.slotCap = slotCap, a.counters = counters;
    a.steps = steps, a.refill = refill, a.capLen = (uint32_t)capLen, a.tabEntries = 1025u, a.ownerMagic = magic, a.zero = 0ull;
    const size_t ldsBytes = size_t(3) * 1025 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    auto go = [&](int n) {
        Args b = a;
        b.steps = n;
        hipMemset(flags, 0, size_t(8) * qcap * 4);
        hipMemset(qctl, 0, 256 * sizeof(unsigned long long));
        hipMemset(counters, 0, 32 * sizeof(unsigned long long));
        hipEventRecord(e0);
        if (mode == 2)
            hipLaunchKernelGGL(walkOwn<2>, dim3(grid), dim3(block), ldsBytes, 0, b);
        else if (mode)
            hipLaunchKernelGGL(walkOwn<1>, dim3(grid), dim3(block), ldsBytes, 0, b);
        else
            hipLaunchKernelGGL(walkOwn<0>, dim3(grid), dim3(block), ldsBytes, 0, b);
        hipEventRecord(e1);
    };
    go(64);
    hipDeviceSynchronize();
    go(steps);
    if (hipEventSynchronize(e1) != hipSuccess) return 1;
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) return 1;
    unsigned long long host[32], q[256];
    hipMemcpy(host, counters, sizeof(host), hipMemcpyDeviceToHost);
    hipMemcpy(q, qctl, sizeof(q), hipMemcpyDeviceToHost);
    out[0] = ms, out[1] = (double)host[1], out[2] = (double)host[2], out[3] = (double)host[3], out[4] = (double)host[4], out[5] = (double)host[6], out[6] = (double)host[7];
    double leftOver = 0;
    for (int i = 0; i < 8; ++i) leftOver += (double)(q[32 * i + 16] - q[32 * i]);
    out[7] = leftOver;
    for (int i = 0; i < 8; ++i) out[8 + i] = (double)host[8 + i], out[16 + i] = (double)host[16 + i];
    return 0;
}
