// pmc_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) of the primary-emission photon loop.
//
// One photon history per lane, persistent wavefronts.  Every lane is a small state machine:
//
//      LAUNCH -> [peel-off walk per observer] -> PASS1 walk (tau of the whole path) -> sample tau ->
//      PASS2 walk (same path again, up to the sampled tau) -> interaction -> [peel-off walk per observer] ->
//      scatter -> PASS1 ...                                   (MonteCarloSimulation.cpp:538-613 performLifeCycle)
//
// >99 % of the time is spent in grid walks, and ALL three kinds of walk share one step routine, so the wave stays
// converged on it: lanes whose walk has ended park until at least PMC_REFILL_THRESHOLD lanes of the wave wait (or
// nobody walks), then the divergent transition code (launch, sampling, scattering, detection) runs once for all
// of them (ballot/compaction of finished packets).  Histories are handed out to waves in chunks from one global
// counter; the random stream of a history depends only on (seed, history index) (include/pmc_philox.h).
//
// The reference stores the whole path (<= 1000 x 40 B) and binary-searches the interaction point
// (SpatialGridPath.cpp:164-206).  Here the path is walked twice with bit-identical arithmetic instead, which needs
// no per-lane path buffer: pass 1 yields tau_path, pass 2 stops in the segment whose cumulative tau exceeds the
// sampled value and interpolates exactly as findInteractionPoint does.
//
// Arithmetic is IEEE double with contraction off (-ffp-contract=off): the reference build has no FMA, and the
// traversal must produce the same (m, ds) sequence bit for bit (pmc_trace_ray).
//
// Octree traversal (TreeSpatialGrid.cpp:132-217): the reference hops through per-wall neighbour lists of heap
// nodes.  Here a cell is ONE 64-byte record (LeafRec) holding a dyadic box code, the density and one link per
// wall; wall coordinates come from a per-axis table staged in LDS (exactly the reference's doubles).  A link leads
// to the same-size-or-coarser neighbour leaf, or to the internal node covering finer neighbours, from which the
// position descends with the reference's child rule (OctTreeNode.cpp:36-41).  This gives the reference's answer
// whenever the new position lies strictly inside one neighbour; in every other case (position on a shared boundary,
// corner overshoot, rounding) the code falls back to the literal reference algorithm on the neighbour lists kept in
// HBM in the reference's order, followed by the reference's top-down search and next-after escape.

#include "pmc_device.h"
#include "../../include/pmc_philox.h"
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>

#ifndef PMC_REFILL_THRESHOLD
    #define PMC_REFILL_THRESHOLD 16
#endif
#ifndef PMC_STEPS_PER_ROUND
    #define PMC_STEPS_PER_ROUND 8
#endif
#define PMC_HISTORY_CHUNK 256

// the scene of every live context, in constant memory: all accesses are scalar loads
__constant__ DevScene c_scene[PMC_MAX_CONTEXTS];

namespace
{
    enum Phase : int
    {
        PH_WALK = 0,
        PH_LAUNCH = 1,      // needs a new history
        PH_PASS1_DONE = 2,  // full path walked: tau_path known
        PH_PASS2_DONE = 3,  // interaction point found
        PH_PEEL_DONE = 4,   // optical depth towards the observer known
        PH_START = 5,       // walk parameters set: run start-of-walk code
        PH_DONE = 6         // no more histories
    };
    enum Mode : int { MODE_PASS1 = 0, MODE_PASS2 = 1, MODE_PEEL = 2 };
    enum Grid : int { GRID_CART = PMC_GRID_CARTESIAN, GRID_TREE = PMC_GRID_OCTREE };

    // ------------------------------------------------------------------------------------------------
    struct Rng
    {
        uint32_t h0, h1, block, have;
        double spare;
    };
    __device__ __forceinline__ void rngInit(Rng& g, uint64_t history)
    {
        g.h0 = (uint32_t)history;
        g.h1 = (uint32_t)(history >> 32);
        g.block = 0;
        g.have = 0;
        g.spare = 0.;
    }
    __device__ __forceinline__ double rngUniform(Rng& g, uint64_t seed)
    {
        if (g.have)
        {
            g.have = 0;
            return g.spare;
        }
        uint32_t c[4] = {g.h0, g.h1, g.block, 0x504d4331u};
        g.block += 1;
        pmc_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        g.spare = pmc_bits_to_unit(c[2], c[3]);
        g.have = 1;
        return pmc_bits_to_unit(c[0], c[1]);
    }

    // ------------------------------------------------------------------------------------------------
    // per-lane walk state (the PathSegmentGenerator of the reference plus the running sums of the caller)
    struct Walk
    {
        double rx, ry, rz;  // generator position (PathSegmentGenerator::_rx..)
        double kx, ky, kz;  // direction
        double tau, s;      // cumulative optical depth and path length of the segments emitted so far
        double ds;          // length of the pending segment (exit distance of the current cell)
        double dens;        // number density of the current cell
        int cell;           // octree: leaf m; Cartesian: m
        int ci, cj, ck;     // Cartesian indices
        int axis;           // exit axis 0,1,2 of the pending segment
        int link;           // octree: link through the exit wall
        int lastm;          // cell of the last segment added to the path (ds > 0)
    };

    struct Lds
    {
        const double* grid;  // octree: coordinate tables [3][tabn]; Cartesian: xv | yv | zv
        const double* lamb;  // dust tables (LDS or global): border, ext, sca, g
        const double* sext;
        const double* ssca;
        const double* asym;
        const double* src;  // Sersic s | M, or oligo lambda | weight
        double* sed;        // privatised SED blocks
        int tabn;
    };

    __device__ __forceinline__ int locateBasic(const double* xv, double x, int n)
    {
        int jl = -1, ju = n;
        while (ju - jl > 1)
        {
            int jm = (ju + jl) >> 1;
            if (x < xv[jm])
                ju = jm;
            else
                jl = jm;
        }
        return jl;
    }
    __device__ __forceinline__ int locateClip(const double* xv, int n, double x)
    {
        if (x < xv[0]) return 0;
        return locateBasic(xv, x, n - 1);
    }
    __device__ __forceinline__ int locate(const double* xv, int n, double x)
    {
        if (x == xv[n - 1]) return n - 2;
        return locateBasic(xv, x, n);
    }

    // ------------------------------------------------------------------------------------------------
    // octree helpers

    __device__ __forceinline__ void decodeBox(uint64_t code, int lmax, int& fx, int& fy, int& fz, int& size)
    {
        int level = (int)((code >> 48) & 0xF);
        fx = (int)((code >> 32) & 0xFFFF);
        fy = (int)((code >> 16) & 0xFFFF);
        fz = (int)(code & 0xFFFF);
        size = 1 << (lmax - level);
    }

    // TreeNode::leafChild from the given link downwards (TreeNode.cpp:66-76, OctTreeNode.cpp:36-41)
    __device__ __forceinline__ int descend(const DevScene& S, const Lds& L, int link, double x, double y, double z)
    {
        while (link <= -2)
        {
            const NodeRec* rec = S.nodes + (-2 - link);
            int fx, fy, fz, size;
            decodeBox(rec->code, S.lmax, fx, fy, fz, size);
            int half = size >> 1;
            double xc = L.grid[fx + half];
            double yc = L.grid[L.tabn + fy + half];
            double zc = L.grid[2 * L.tabn + fz + half];
            int l = (x < xc ? 0 : 1) + (y < yc ? 0 : 2) + (z < zc ? 0 : 4);
            link = rec->child[l];
        }
        return link;
    }
    // root()->leafChild(r): -1 if r is outside the (closed) root box
    __device__ __forceinline__ int topDown(const DevScene& S, const Lds& L, double x, double y, double z)
    {
        if (!(x >= S.gx0 && x <= S.gx1 && y >= S.gy0 && y <= S.gy1 && z >= S.gz0 && z <= S.gz1)) return -1;
        return descend(S, L, S.root_link, x, y, z);
    }
    __device__ __forceinline__ bool leafContains(const DevScene& S, const Lds& L, int m, double x, double y, double z)
    {
        int fx, fy, fz, size;
        decodeBox(S.leaves[m].code, S.lmax, fx, fy, fz, size);
        return x >= L.grid[fx] && x <= L.grid[fx + size] && y >= L.grid[L.tabn + fy] && y <= L.grid[L.tabn + fy + size]
               && z >= L.grid[2 * L.tabn + fz] && z <= L.grid[2 * L.tabn + fz + size];
    }
    // TreeNode::neighbor on the reference's neighbour list (TreeNode.cpp:103-112)
    __device__ __forceinline__ int listNeighbor(const DevScene& S, const Lds& L, int m, int wall, double x, double y, double z)
    {
        int b = S.nbr_start[6 * (int64_t)m + wall], e = S.nbr_start[6 * (int64_t)m + wall + 1];
        for (int q = b; q < e; ++q)
        {
            int cand = S.nbr_list[q];
            if (leafContains(S, L, cand, x, y, z)) return cand;
        }
        return -1;
    }

    // loads the record of leaf m and prepares the pending segment: exit distances and the link through the exit wall
    // (TreeSpatialGrid.cpp:160-185).  If `check` is set, first verifies that the position lies inside the closed box
    // and strictly between the transverse walls (no other neighbour can contain it); returns false otherwise.
    template<bool CHECK> __device__ __forceinline__ bool treeEnter(const DevScene& S, const Lds& L, Walk& w, int m, int axis)
    {
        const LeafRec* rec = S.leaves + m;
        // 40 useful bytes of one 64-byte line: 2 x 16 B + 8 B
        const uint4 q0 = *reinterpret_cast<const uint4*>(rec);
        const uint4 q1 = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(rec) + 16);
        const uint2 q2 = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(rec) + 32);
        uint64_t code = ((uint64_t)q0.y << 32) | q0.x;
        double dens = __longlong_as_double(((long long)q0.w << 32) | q0.z);
        int fx, fy, fz, size;
        decodeBox(code, S.lmax, fx, fy, fz, size);
        const double X0 = L.grid[fx], X1 = L.grid[fx + size];
        const double Y0 = L.grid[L.tabn + fy], Y1 = L.grid[L.tabn + fy + size];
        const double Z0 = L.grid[2 * L.tabn + fz], Z1 = L.grid[2 * L.tabn + fz + size];
        if (CHECK)
        {
            bool inside = w.rx >= X0 && w.rx <= X1 && w.ry >= Y0 && w.ry <= Y1 && w.rz >= Z0 && w.rz <= Z1;
            bool tie;
            if (axis == 0)
                tie = w.ry == Y0 || w.ry == Y1 || w.rz == Z0 || w.rz == Z1;
            else if (axis == 1)
                tie = w.rx == X0 || w.rx == X1 || w.rz == Z0 || w.rz == Z1;
            else
                tie = w.rx == X0 || w.rx == X1 || w.ry == Y0 || w.ry == Y1;
            if (!inside || tie) return false;
        }
        const bool nx = w.kx < 0.0, ny = w.ky < 0.0, nz = w.kz < 0.0;
        const double xnext = nx ? X0 : X1;
        const double ynext = ny ? Y0 : Y1;
        const double znext = nz ? Z0 : Z1;
        const int lx = nx ? (int)q1.x : (int)q1.y;
        const int ly = ny ? (int)q1.z : (int)q1.w;
        const int lz = nz ? (int)q2.x : (int)q2.y;
        const double dsx = (fabs(w.kx) > 1e-15) ? (xnext - w.rx) / w.kx : DBL_MAX;
        const double dsy = (fabs(w.ky) > 1e-15) ? (ynext - w.ry) / w.ky : DBL_MAX;
        const double dsz = (fabs(w.kz) > 1e-15) ? (znext - w.rz) / w.kz : DBL_MAX;
        if (dsx <= dsy && dsx <= dsz)
        {
            w.ds = dsx;
            w.axis = 0;
            w.link = lx;
        }
        else if (dsy <= dsx && dsy <= dsz)
        {
            w.ds = dsy;
            w.axis = 1;
            w.link = ly;
        }
        else
        {
            w.ds = dsz;
            w.axis = 2;
            w.link = lz;
        }
        w.cell = m;
        w.dens = dens;
        return true;
    }

    // after the pending segment has been emitted and the position advanced: find the next cell
    // (TreeSpatialGrid.cpp:186-207).  Returns false when the path leaves the grid (State::Outside).
    __device__ __forceinline__ bool treeAdvance(const DevScene& S, const Lds& L, Walk& w)
    {
        const int old = w.cell;
        const int axis = w.axis;
        int next = w.link;
        if (next <= -2) next = descend(S, L, next, w.rx, w.ry, w.rz);
        if (next >= 0 && treeEnter<true>(S, L, w, next, axis)) return true;  // the common case

        // ---- everything else: the reference algorithm, literally
        const bool neg = axis == 0 ? (w.kx < 0.0) : axis == 1 ? (w.ky < 0.0) : (w.kz < 0.0);
        const int wall = 2 * axis + (neg ? 0 : 1);
        next = (w.link == PMC_LINK_NONE) ? -1 : listNeighbor(S, L, old, wall, w.rx, w.ry, w.rz);
        if (next < 0) next = topDown(S, L, w.rx, w.ry, w.rz);
        if (next == old)
        {
            // PathSegmentGenerator::propagateToNextAfter (PathSegmentGenerator.hpp:148-153)
            w.rx = nextafter(w.rx, (w.kx < 0.) ? -DBL_MAX : DBL_MAX);
            w.ry = nextafter(w.ry, (w.ky < 0.) ? -DBL_MAX : DBL_MAX);
            w.rz = nextafter(w.rz, (w.kz < 0.) ? -DBL_MAX : DBL_MAX);
            next = topDown(S, L, w.rx, w.ry, w.rz);
        }
        if (next < 0 || next == old) return false;
        treeEnter<false>(S, L, w, next, axis);
        return true;
    }

    // ------------------------------------------------------------------------------------------------
    // Cartesian helpers (CartesianSpatialGrid.cpp:87-163)

    __device__ __forceinline__ void cartEnter(const DevScene& S, const Lds& L, Walk& w)
    {
        const double* xv = L.grid;
        const double* yv = L.grid + (S.nx + 1);
        const double* zv = yv + (S.ny + 1);
        const int m = w.ck + S.nz * w.cj + S.nz * S.ny * w.ci;
        const double xE = (w.kx < 0.0) ? xv[w.ci] : xv[w.ci + 1];
        const double yE = (w.ky < 0.0) ? yv[w.cj] : yv[w.cj + 1];
        const double zE = (w.kz < 0.0) ? zv[w.ck] : zv[w.ck + 1];
        const double dsx = (fabs(w.kx) > 1e-15) ? (xE - w.rx) / w.kx : DBL_MAX;
        const double dsy = (fabs(w.ky) > 1e-15) ? (yE - w.ry) / w.ky : DBL_MAX;
        const double dsz = (fabs(w.kz) > 1e-15) ? (zE - w.rz) / w.kz : DBL_MAX;
        if (dsx <= dsy && dsx <= dsz)
        {
            w.ds = dsx;
            w.axis = 0;
        }
        else if (dsy < dsx && dsy <= dsz)
        {
            w.ds = dsy;
            w.axis = 1;
        }
        else
        {
            w.ds = dsz;
            w.axis = 2;
        }
        w.cell = m;
        w.dens = S.cell_density[m];
    }
    __device__ __forceinline__ bool cartAdvance(const DevScene& S, const Lds& L, Walk& w)
    {
        const double* xv = L.grid;
        const double* yv = L.grid + (S.nx + 1);
        const double* zv = yv + (S.ny + 1);
        const double ds = w.ds;
        bool inside = true;
        if (w.axis == 0)
        {
            w.rx = (w.kx < 0.0) ? xv[w.ci] : xv[w.ci + 1];
            w.ry += w.ky * ds;
            w.rz += w.kz * ds;
            w.ci += (w.kx < 0.0) ? -1 : 1;
            if (w.ci >= S.nx || w.ci < 0) inside = false;
        }
        else if (w.axis == 1)
        {
            w.ry = (w.ky < 0.0) ? yv[w.cj] : yv[w.cj + 1];
            w.rx += w.kx * ds;
            w.rz += w.kz * ds;
            w.cj += (w.ky < 0.0) ? -1 : 1;
            if (w.cj >= S.ny || w.cj < 0) inside = false;
        }
        else
        {
            w.rz = (w.kz < 0.0) ? zv[w.ck] : zv[w.ck + 1];
            w.rx += w.kx * ds;
            w.ry += w.ky * ds;
            w.ck += (w.kz < 0.0) ? -1 : 1;
            if (w.ck >= S.nz || w.ck < 0) inside = false;
        }
        if (!inside) return false;
        cartEnter(S, L, w);
        return true;
    }

    // ------------------------------------------------------------------------------------------------
    // PathSegmentGenerator::moveInside (PathSegmentGenerator.cpp:11-112); returns false if the path misses the grid;
    // cumds receives the length of the initial segment outside the grid
    __device__ __forceinline__ bool moveInside(const DevScene& S, Walk& w, double& cumds)
    {
        const double eps = S.eps;
        cumds = 0.;
        if (w.rx <= S.gx0)
        {
            if (w.kx <= 0.0) return false;
            double d = (S.gx0 - w.rx) / w.kx;
            w.rx = S.gx0 + eps;
            w.ry += w.ky * d;
            w.rz += w.kz * d;
            cumds += d;
        }
        else if (w.rx >= S.gx1)
        {
            if (w.kx >= 0.0) return false;
            double d = (S.gx1 - w.rx) / w.kx;
            w.rx = S.gx1 - eps;
            w.ry += w.ky * d;
            w.rz += w.kz * d;
            cumds += d;
        }
        if (w.ry <= S.gy0)
        {
            if (w.ky <= 0.0) return false;
            double d = (S.gy0 - w.ry) / w.ky;
            w.rx += w.kx * d;
            w.ry = S.gy0 + eps;
            w.rz += w.kz * d;
            cumds += d;
        }
        else if (w.ry >= S.gy1)
        {
            if (w.ky >= 0.0) return false;
            double d = (S.gy1 - w.ry) / w.ky;
            w.rx += w.kx * d;
            w.ry = S.gy1 - eps;
            w.rz += w.kz * d;
            cumds += d;
        }
        if (w.rz <= S.gz0)
        {
            if (w.kz <= 0.0) return false;
            double d = (S.gz0 - w.rz) / w.kz;
            w.rx += w.kx * d;
            w.ry += w.ky * d;
            w.rz = S.gz0 + eps;
            cumds += d;
        }
        else if (w.rz >= S.gz1)
        {
            if (w.kz >= 0.0) return false;
            double d = (S.gz1 - w.rz) / w.kz;
            w.rx += w.kx * d;
            w.ry += w.ky * d;
            w.rz = S.gz1 - eps;
            cumds += d;
        }
        if (!(w.rx >= S.gx0 && w.rx <= S.gx1 && w.ry >= S.gy0 && w.ry <= S.gy1 && w.rz >= S.gz0 && w.rz <= S.gz1))
            return false;
        return true;
    }

    // start of a walk from (r, k): State::Unknown branch of next().  hint = a leaf that probably contains r
    // (octree only).  Returns false if the path has no cell segments at all; the initial outside segment, if any,
    // has been added to w.s.
    template<int GRID> __device__ __forceinline__ bool startWalk(const DevScene& S, const Lds& L, Walk& w, int hint)
    {
        w.tau = 0.;
        w.s = 0.;
        w.lastm = -1;
        double cumds;
        if (!moveInside(S, w, cumds)) return false;
        if (cumds > 0.) w.s += cumds;  // SpatialGridPath::addSegment(-1, cumds)
        if (GRID == GRID_CART)
        {
            w.ci = locateClip(L.grid, S.nx + 1, w.rx);
            w.cj = locateClip(L.grid + (S.nx + 1), S.ny + 1, w.ry);
            w.ck = locateClip(L.grid + (S.nx + 1) + (S.ny + 1), S.nz + 1, w.rz);
            cartEnter(S, L, w);
            return true;
        }
        else
        {
            int m = -1;
            if (hint >= 0)
            {
                // strictly inside the hinted leaf => the top-down search would end there as well
                int fx, fy, fz, size;
                decodeBox(S.leaves[hint].code, S.lmax, fx, fy, fz, size);
                if (w.rx > L.grid[fx] && w.rx < L.grid[fx + size] && w.ry > L.grid[L.tabn + fy]
                    && w.ry < L.grid[L.tabn + fy + size] && w.rz > L.grid[2 * L.tabn + fz]
                    && w.rz < L.grid[2 * L.tabn + fz + size])
                    m = hint;
            }
            if (m < 0) m = topDown(S, L, w.rx, w.ry, w.rz);
            // (moveInside guarantees that r is inside the root box, so m >= 0)
            treeEnter<false>(S, L, w, m, 0);
            return true;
        }
    }

    // ------------------------------------------------------------------------------------------------
    __device__ __forceinline__ void atomicAddF64(double* address, double value)
    {
        unsafeAtomicAdd(address, value);
    }

    // Henyey-Greenstein helpers (DustMix.cpp:395-425)
    __device__ __forceinline__ double valueHG(double g, double costheta)
    {
        double t = 1. + g * g - 2. * g * costheta;
        return (1. - g) * (1. + g) / sqrt(t * t * t);
    }
    __device__ __forceinline__ double integralHG(double g, double cosalpha, double cosbeta)
    {
        double ta = sqrt(1. + g * g - 2. * g * cosalpha);
        double tb = sqrt(1. + g * g - 2. * g * cosbeta);
        double f1 = (1. - g) * (1. + g) / g;
        double f2 = (tb - ta) / (tb * ta);
        return f1 * f2;
    }
    __device__ __noinline__ double meanHG(double g, double costheta)
    {
        const double delta = 4. * M_PI / 180.;
        double theta = acos(costheta);
        double cosalpha = cos(theta - delta);
        double cosbeta = cos(theta + delta);
        if (theta < delta) return (integralHG(g, 1., cosalpha) + integralHG(g, 1., cosbeta)) / (2. - cosalpha - cosbeta);
        if (theta > M_PI - delta)
            return (integralHG(g, cosalpha, -1.) + integralHG(g, cosbeta, -1.)) / (2. + cosalpha + cosbeta);
        return integralHG(g, cosalpha, cosbeta) / (cosalpha - cosbeta);
    }

    // Direction(theta, phi) + Random::direction() (Direction.cpp:11-38, Random.cpp:121-126)
    __device__ __forceinline__ void randomDirection(Rng& rng, uint64_t seed, double& kx, double& ky, double& kz)
    {
        double theta = acos(2.0 * rngUniform(rng, seed) - 1.0);
        double phi = 2.0 * M_PI * rngUniform(rng, seed);
        const double eps = 1e-8;
        if (theta <= eps)
        {
            kx = 0, ky = 0, kz = 1;
        }
        else if (theta >= M_PI - eps)
        {
            kx = 0, ky = 0, kz = -1;
        }
        else
        {
            double sintheta, costheta, sinphi, cosphi;
            sincos(theta, &sintheta, &costheta);
            sincos(phi, &sinphi, &cosphi);
            kx = sintheta * cosphi;
            ky = sintheta * sinphi;
            kz = costheta;
        }
    }

    __device__ __forceinline__ double interpolateLogLog(double x, double x1, double x2, double f1, double f2)
    {
        if (f1 <= 0 || f2 <= 0)
        {
            if (x == x1) return f1;
            if (x == x2) return f2;
            return 0;
        }
        return f1 * exp(log(x / x1) / log(x2 / x1) * (log(f2 / f1)));
    }
    __device__ __noinline__ double gexp(double p, double x)
    {
        const double q = 1.0 - p;
        if (q == 0.0) return exp(x);
        if (fabs(q) < 1e-3)
        {
            double x2 = x * x;
            return exp(x)
                   * (1.0 - 0.5 * x2 * q + 1.0 / 24.0 * x * x2 * (8.0 + 3.0 * x) * q * q
                      - 1.0 / 48.0 * x2 * x2 * (12.0 + 8.0 * x + x2) * q * q * q);
        }
        return pow(1.0 + q * x, 1.0 / q);
    }

    // ------------------------------------------------------------------------------------------------
    // the per-lane packet state outside of walks
    struct Packet
    {
        double rx, ry, rz;  // position
        double kx, ky, kz;  // propagation direction
        double lambda, W;   // wavelength and weight (luminosity = W / lambda)
        double Lthreshold;
        double sext, ssca, g;  // dust properties at lambda
        // peel-off packet under way
        double pW;       // weight of the peel-off packet
        double ptau;     // optical depth towards the current observer
        double target;   // sampled optical depth (pass 2)
        double taupath;
        double sint;     // interaction distance
        double nint;     // density of the interaction cell
        int mint;        // interaction cell
        int nscatt;
        int pinst;       // instrument being served by the peel-off sequence
        int pscatt;      // numScatt of the peel-off packet (0 emission, nscatt+1 scattering)
        int cellhint;    // octree leaf that contains the packet position
        int ell[PMC_MAX_INSTRUMENTS];  // wavelength bin per instrument (-1: out of range)
        int nstat[PMC_MAX_INSTRUMENTS];
        uint64_t history;
        bool alive;
    };

    struct Counters
    {
        uint32_t histories, paths, visits, updates, scatterings, overflows, rewalks;  // per lane and launch: < 2^32
    };

    // FrameInstrument::pixelOnDetector (FrameInstrument.cpp:45-65)
    __device__ __forceinline__ int pixelOnDetector(const DevInstrument& I, double x, double y, double z)
    {
        double xpp = -I.sinphi * x + I.cosphi * y;
        double ypp = -I.cosphi * I.costheta * x - I.sinphi * I.costheta * y + I.sintheta * z;
        double xp = I.cosomega * xpp - I.sinomega * ypp;
        double yp = I.sinomega * xpp + I.cosomega * ypp;
        int i = (int)floor((xp - I.xpmin) / I.xpsiz);
        int j = (int)floor((yp - I.ypmin) / I.ypsiz);
        if (i < 0 || i >= I.nxp || j < 0 || j >= I.nyp) return -1;
        return i + I.nxp * j;
    }

    // FluxRecorder::detect for one peel-off packet whose optical depth is known (FluxRecorder.cpp:304-468)
    __device__ __forceinline__ void detect(const DevScene& S, const Lds& L, Packet& p, Counters& cnt, int inst, int lane,
                                           int64_t laneIndex)
    {
        const DevInstrument& I = S.inst[inst];
        const int l = pixelOnDetector(I, p.rx, p.ry, p.rz);
        if (!I.include_sed && l < 0) return;
        const int ell = p.ell[inst];
        if (ell < 0) return;
        const double Lum = p.pW / p.lambda;
        const double Lext = Lum * exp(-p.ptau);
        const int numScatt = p.pscatt;
        // component slot(s): total only, or transparent+direct / scattered (+level)
        if (I.include_sed)
        {
            double* sed = L.sed + I.sed_lds_offset;
            const int nl = I.num_lambda;
            if (!I.record_components)
                atomicAdd(&sed[ell], Lext);
            else if (numScatt == 0)
            {
                atomicAdd(&sed[0 * nl + ell], Lum);
                atomicAdd(&sed[1 * nl + ell], Lext);
            }
            else
            {
                atomicAdd(&sed[2 * nl + ell], Lext);
                if (numScatt <= I.num_levels) atomicAdd(&sed[(3 + numScatt - 1) * nl + ell], Lext);
            }
        }
        if (I.include_ifu && l >= 0)
        {
            const int64_t len = I.npix * I.num_lambda;
            double* ifu = S.frames + I.ifu_offset + l + (int64_t)ell * I.npix;
            if (!I.record_components)
            {
                atomicAddF64(ifu, Lext);
                cnt.updates += 1;
            }
            else if (numScatt == 0)
            {
                atomicAddF64(ifu, Lum);
                atomicAddF64(ifu + len, Lext);
                cnt.updates += 2;
            }
            else
            {
                atomicAddF64(ifu + 2 * len, Lext);
                cnt.updates += 1;
                if (numScatt <= I.num_levels)
                {
                    atomicAddF64(ifu + (3 + numScatt - 1) * len, Lext);
                    cnt.updates += 1;
                }
            }
        }
        if (I.record_stats)
        {
            int n = p.nstat[inst];
            if (n < PMC_STAT_CAP)
            {
                int64_t slot = ((int64_t)inst * PMC_STAT_CAP + n) * S.stat_lanes + laneIndex;
                S.stat_bin[slot] = l;
                S.stat_w[slot] = Lext;
                p.nstat[inst] = n + 1;
            }
            else
                p.nstat[inst] = PMC_STAT_CAP + 1;  // overflow marker
        }
    }

    // FluxRecorder::recordContributions for the history that just ended (FluxRecorder.cpp:962-1014): contributions
    // to the same bin are summed before taking powers.  All contributions of a history share the wavelength bin.
    __device__ __forceinline__ void flushStatistics(const DevScene& S, const Lds& L, Packet& p, Counters& cnt, int64_t laneIndex)
    {
        for (int inst = 0; inst < S.num_instruments; ++inst)
        {
            const DevInstrument& I = S.inst[inst];
            if (!I.record_stats) continue;
            int n = p.nstat[inst];
            if (n > PMC_STAT_CAP)
            {
                n = PMC_STAT_CAP;
                cnt.overflows += 1;
            }
            p.nstat[inst] = 0;
            const int ell = p.ell[inst];
            if (n == 0 || ell < 0) continue;
            const int64_t base = (int64_t)inst * PMC_STAT_CAP * S.stat_lanes + laneIndex;
            double wsed = 0.;
            for (int e = 0; e < n; ++e)
            {
                const int bin = S.stat_bin[base + e * S.stat_lanes];
                const double we = S.stat_w[base + e * S.stat_lanes];
                wsed += we;
                if (bin < 0 || !I.include_ifu) continue;
                bool first = true;
                for (int j = 0; j < e; ++j)
                    if (S.stat_bin[base + j * S.stat_lanes] == bin) first = false;
                if (!first) continue;
                double w = we;
                for (int j = e + 1; j < n; ++j)
                    if (S.stat_bin[base + j * S.stat_lanes] == bin) w += S.stat_w[base + j * S.stat_lanes];
                double* wifu = S.frames + I.wifu_offset + bin + (int64_t)ell * I.npix;
                const int64_t len = I.npix * I.num_lambda;
                double wn = 1.;
                for (int k = 0; k <= 4; ++k)
                {
                    atomicAddF64(wifu + k * len, wn);
                    wn *= w;
                }
                cnt.updates += 5;
            }
            if (I.include_sed)
            {
                double* ws = L.sed + I.sed_lds_offset + I.num_components * I.num_lambda;
                double wn = 1.;
                for (int k = 0; k <= 4; ++k)
                {
                    atomicAdd(&ws[k * I.num_lambda + ell], wn);
                    wn *= wsed;
                }
            }
        }
    }

    // SourceSystem::launch ... PhotonPacket::launch (SourceSystem.cpp:101-112, NormalizedSource.cpp:73-110,
    // PointSource.cpp:32-43, GeometricSource.cpp:66-82, SpheGeometry.cpp:25-32, SersicGeometry.cpp:41-45)
    __device__ __forceinline__ void launch(const DevScene& S, const Lds& L, Packet& p, Rng& rng, uint64_t history,
                                           const uint64_t seed)
    {
        rngInit(rng, history);
        double lambda, w;
        if (S.lambda_mode == PMC_LAMBDA_OLIGO)
        {
            (void)rngUniform(rng, seed);  // the `uniform() > xi` test of NormalizedSource::launch with xi = 1
            int index = (int)(rngUniform(rng, seed) * S.num_oligo);
            if (index > S.num_oligo - 1) index = S.num_oligo - 1;
            lambda = S.oligo_lambda[index];
            w = S.oligo_weight[index];
        }
        else
        {
            const double xi = S.lambda_bias;
            bool fromSed = true;
            if (xi != 0.) fromSed = rngUniform(rng, seed) > xi;
            if (fromSed)
            {
                // Random::cdfLogLog (Random.cpp:209-216)
                double X = rngUniform(rng, seed);
                int i = locateClip(S.sed_P, S.num_sed, X);
                double alpha = log(S.sed_p[i + 1] / S.sed_p[i]) / log(S.sed_lambda[i + 1] / S.sed_lambda[i]);
                lambda = S.sed_lambda[i] * gexp(-alpha, (X - S.sed_P[i]) / (S.sed_p[i] * S.sed_lambda[i]));
            }
            else if (S.bias_kind == PMC_BIAS_LIN)
                lambda = S.bias_min + (S.bias_max - S.bias_min) * rngUniform(rng, seed);
            else
                lambda = exp(log(S.bias_min) + (log(S.bias_max) - log(S.bias_min)) * rngUniform(rng, seed));
            if (xi == 0.)
                w = 1.;
            else
            {
                double sl = 0.;
                if (lambda >= S.sed_lambda[0] && lambda <= S.sed_lambda[S.num_sed - 1])
                {
                    int i = locate(S.sed_lambda, S.num_sed, lambda);
                    if (i < 0) i = 0;
                    sl = interpolateLogLog(lambda, S.sed_lambda[i], S.sed_lambda[i + 1], S.sed_p[i], S.sed_p[i + 1]);
                }
                if (sl == 0.)
                    w = 0.;
                else
                {
                    double b = 0.;
                    if (lambda >= S.bias_min && lambda <= S.bias_max)
                        b = S.bias_kind == PMC_BIAS_LIN ? 1. / (S.bias_max - S.bias_min)
                                                        : 1. / ((log(S.bias_max) - log(S.bias_min)) * lambda);
                    w = sl / ((1 - xi) * sl + xi * b);
                }
            }
        }
        if (S.source_kind == PMC_SOURCE_POINT)
        {
            p.rx = S.src_pos[0];
            p.ry = S.src_pos[1];
            p.rz = S.src_pos[2];
        }
        else if (S.source_kind == PMC_SOURCE_SERSIC)
        {
            const double* sv = L.src;
            const double* Mv = L.src + S.sersic_n;
            double X = rngUniform(rng, seed);
            int n = S.sersic_n;
            int i = locate(Mv, n, X);
            double s;
            if (i < 0)
                s = sv[0];
            else if (i >= n - 1)
                s = sv[n - 1];
            else
                s = interpolateLogLog(X, Mv[i], Mv[i + 1], sv[i], sv[i + 1]);
            double radius = S.reff * s;
            double dx, dy, dz;
            randomDirection(rng, seed, dx, dy, dz);
            p.rx = dx * radius;
            p.ry = dy * radius;
            p.rz = dz * radius;
        }
        else
        {
            double x = rngUniform(rng, seed);
            double y = rngUniform(rng, seed);
            double z = rngUniform(rng, seed);
            p.rx = S.src_box[0] + x * (S.src_box[3] - S.src_box[0]);
            p.ry = S.src_box[1] + y * (S.src_box[4] - S.src_box[1]);
            p.rz = S.src_box[2] + z * (S.src_box[5] - S.src_box[2]);
        }
        randomDirection(rng, seed, p.kx, p.ky, p.kz);
        const double Lw = S.packet_luminosity * w;
        p.lambda = lambda;
        p.W = Lw * lambda;
        p.nscatt = 0;
        p.history = history;
        p.cellhint = -1;
        // dust properties at this wavelength (DustMix::indexForLambda, DustMix.cpp:276-279)
        const int il = locateClip(L.lamb, S.num_lambda, lambda);
        p.sext = L.sext[il];
        p.ssca = L.ssca[il];
        p.g = L.asym[il];
        // wavelength bin of every instrument (DisjointWavelengthGrid::bin, DisjointWavelengthGrid.cpp:334-345)
        for (int i = 0; i < S.num_instruments; ++i)
        {
            const DevInstrument& I = S.inst[i];
            int lo = 0, hi = I.num_border;  // upper_bound
            while (lo < hi)
            {
                int mid = (lo + hi) >> 1;
                if (lambda < I.border[mid])
                    hi = mid;
                else
                    lo = mid + 1;
            }
            p.ell[i] = I.ellv[lo];
        }
    }

    // Random::exponCutoff (Random.cpp:105-116)
    __device__ __forceinline__ double exponCutoff(Rng& rng, uint64_t seed, double xmax)
    {
        if (xmax == 0.0) return 0.0;
        if (xmax < 1e-10) return rngUniform(rng, seed) * xmax;
        double x = -log(1.0 - rngUniform(rng, seed) * (1.0 - exp(-xmax)));
        while (x > xmax) x = -log(1.0 - rngUniform(rng, seed) * (1.0 - exp(-xmax)));
        return x;
    }

    // DustMix::performScattering, HG (DustMix.cpp:490-511) with Random::direction(bfk, costheta) (Random.cpp:130-164)
    __device__ __forceinline__ void scatter(Packet& p, Rng& rng, const uint64_t seed)
    {
        const double g = p.g;
        if (fabs(g) < 1e-6)
            randomDirection(rng, seed, p.kx, p.ky, p.kz);
        else
        {
            double f = ((1.0 - g) * (1.0 + g)) / (1.0 - g + 2.0 * g * rngUniform(rng, seed));
            double costheta = (1.0 + g * g - f * f) / (2.0 * g);
            double phi = 2.0 * M_PI * rngUniform(rng, seed);
            double sinphi, cosphi;
            sincos(phi, &sinphi, &cosphi);
            double sintheta = sqrt(fabs((1.0 - costheta) * (1.0 + costheta)));
            const double kx = p.kx, ky = p.ky, kz = p.kz;
            double kxnew, kynew, kznew;
            if (kz > 0.99999)
            {
                kxnew = cosphi * sintheta;
                kynew = sinphi * sintheta;
                kznew = costheta;
            }
            else if (kz < -0.99999)
            {
                kxnew = cosphi * sintheta;
                kynew = sinphi * sintheta;
                kznew = -costheta;
            }
            else
            {
                double root = sqrt((1.0 - kz) * (1.0 + kz));
                kxnew = sintheta / root * (-kx * kz * cosphi + ky * sinphi) + kx * costheta;
                kynew = -sintheta / root * (ky * kz * cosphi + kx * sinphi) + ky * costheta;
                kznew = root * sintheta * cosphi + kz * costheta;
            }
            p.kx = kxnew;
            p.ky = kynew;
            p.kz = kznew;
        }
        p.nscatt += 1;
    }

    __device__ __forceinline__ unsigned long long waveSum(uint32_t value)
    {
        unsigned long long v = value;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        return v;
    }

    // ================================================================================================
    //  the photon loop kernel
    // ================================================================================================
    template<int GRID> __global__ __launch_bounds__(256) void primaryEmissionKernel(const int slot, const uint64_t first,
                                                                                   const uint64_t count, const uint64_t seedArg)
    {
        const DevScene& S = c_scene[slot];
        extern __shared__ double lds[];
        const int tid = threadIdx.x;
        const int lane = tid & 63;

        // ---- stage the read-only tables in LDS (coalesced reads from HBM, once per workgroup)
        {
            const int ngrid = (GRID == GRID_TREE) ? 3 * ((1 << S.lmax) + 1) : (S.nx + 1) + (S.ny + 1) + (S.nz + 1);
            double* g = lds + S.lds_grid_off;
            if (GRID == GRID_TREE)
                for (int i = tid; i < ngrid; i += blockDim.x) g[i] = S.coord_tab[i];
            else
            {
                for (int i = tid; i <= S.nx; i += blockDim.x) g[i] = S.xv[i];
                for (int i = tid; i <= S.ny; i += blockDim.x) g[(S.nx + 1) + i] = S.yv[i];
                for (int i = tid; i <= S.nz; i += blockDim.x) g[(S.nx + 1) + (S.ny + 1) + i] = S.zv[i];
            }
            if (S.dust_in_lds)
            {
                double* d = lds + S.lds_dust_off;
                for (int i = tid; i < S.num_lambda; i += blockDim.x)
                {
                    d[i] = S.lambda_border[i];
                    d[S.num_lambda + i] = S.sigma_ext[i];
                    d[2 * S.num_lambda + i] = S.sigma_sca[i];
                    d[3 * S.num_lambda + i] = S.asymmpar[i];
                }
            }
            if (S.source_kind == PMC_SOURCE_SERSIC)
            {
                double* s = lds + S.lds_src_off;
                for (int i = tid; i < S.sersic_n; i += blockDim.x)
                {
                    s[i] = S.sersic_s[i];
                    s[S.sersic_n + i] = S.sersic_M[i];
                }
            }
            double* sed = lds + S.lds_sed_off;
            for (int i = tid; i < S.lds_sed_len; i += blockDim.x) sed[i] = 0.;
        }
        __syncthreads();

        Lds L;
        L.grid = lds + S.lds_grid_off;
        L.tabn = (1 << S.lmax) + 1;
        if (S.dust_in_lds)
        {
            L.lamb = lds + S.lds_dust_off;
            L.sext = L.lamb + S.num_lambda;
            L.ssca = L.sext + S.num_lambda;
            L.asym = L.ssca + S.num_lambda;
        }
        else
        {
            L.lamb = S.lambda_border;
            L.sext = S.sigma_ext;
            L.ssca = S.sigma_sca;
            L.asym = S.asymmpar;
        }
        L.src = lds + S.lds_src_off;
        L.sed = lds + S.lds_sed_off;

        const int64_t laneIndex = (int64_t)blockIdx.x * blockDim.x + tid;
        const uint64_t seed = seedArg;

        Packet p;
        p.alive = false;
        p.pinst = 0;
        p.pscatt = 0;
        p.mint = -1;
        p.cellhint = -1;
        for (int i = 0; i < PMC_MAX_INSTRUMENTS; ++i)
        {
            p.ell[i] = -1;
            p.nstat[i] = 0;
        }
        Walk w;
        w.cell = -1;
        w.ds = 0.;
        Rng rng;
        rngInit(rng, 0);
        Counters cnt = {0, 0, 0, 0, 0, 0, 0};
        int phase = PH_LAUNCH;
        int mode = MODE_PEEL;
        double taumax = 0.;
        // wave-uniform pool of histories
        unsigned long long poolNext = 0, poolEnd = 0;
        bool exhausted = false;

        while (true)
        {
            // =========================== transitions (divergent, amortised) ===========================
            const unsigned long long waiting = __ballot(phase != PH_WALK && phase != PH_DONE);
            const unsigned long long walking = __ballot(phase == PH_WALK);
            if (waiting && (__popcll(waiting) >= PMC_REFILL_THRESHOLD || !walking))
            {
                int guard = 0;
                while (__ballot(phase != PH_WALK && phase != PH_DONE))
                {
                    // ---- the optical depth towards an observer is known: record, then serve the next instrument
                    if (phase == PH_PEEL_DONE)
                    {
                        int inst = p.pinst;
                        detect(S, L, p, cnt, inst, lane, laneIndex);
                        // instruments that share the observer reuse the optical depth (FluxRecorder.cpp:329-338)
                        ++inst;
                        while (inst < S.num_instruments && S.inst[inst].same_observer)
                        {
                            detect(S, L, p, cnt, inst, lane, laneIndex);
                            ++inst;
                        }
                        if (inst < S.num_instruments)
                        {
                            // next observer: new peel-off packet from the same position
                            p.pinst = inst;
                            const DevInstrument& I = S.inst[inst];
                            if (p.pscatt == 0)
                                p.pW = p.W;
                            else
                            {
                                double costheta = p.kx * I.kx + p.ky * I.ky + p.kz * I.kz;
                                double value = fabs(p.g) > 0.95 ? meanHG(p.g, costheta) : valueHG(p.g, costheta);
                                p.pW = p.W * (0. + value * 1.);
                            }
                            w.rx = p.rx, w.ry = p.ry, w.rz = p.rz;
                            w.kx = I.kx, w.ky = I.ky, w.kz = I.kz;
                            mode = MODE_PEEL;
                            phase = PH_START;
                        }
                        else if (p.pscatt == 0)
                        {
                            // emission peel-off done: start the photon cycle (MonteCarloSimulation.cpp:561-586)
                            p.Lthreshold = (p.W / p.lambda) / S.min_weight_reduction;
                            w.rx = p.rx, w.ry = p.ry, w.rz = p.rz;
                            w.kx = p.kx, w.ky = p.ky, w.kz = p.kz;
                            mode = S.force_scattering ? MODE_PASS1 : MODE_PASS2;
                            if (!S.force_scattering) p.target = -log(rngUniform(rng, seed));  // Random::expon
                            phase = PH_START;
                        }
                        else
                        {
                            // scattering peel-off done: scatter and continue the cycle
                            scatter(p, rng, seed);
                            cnt.scatterings += 1;
                            w.rx = p.rx, w.ry = p.ry, w.rz = p.rz;
                            w.kx = p.kx, w.ky = p.ky, w.kz = p.kz;
                            mode = S.force_scattering ? MODE_PASS1 : MODE_PASS2;
                            if (!S.force_scattering) p.target = -log(rngUniform(rng, seed));
                            phase = PH_START;
                        }
                    }
                    // ---- interaction point known (MonteCarloSimulation.cpp:724-741 / 746-780)
                    if (phase == PH_PASS2_DONE)
                    {
                        if (p.mint < 0 && !S.force_scattering)
                        {
                            // non-forced: the packet escaped
                            phase = PH_LAUNCH;
                        }
                        else
                        {
                            // MediumSystem::albedoForScattering (MediumSystem.cpp:678-693)
                            const double n = p.nint;
                            const double ksca = n > 0. ? n * p.ssca : 0.;
                            const double kext = n > 0. ? n * p.sext : 0.;
                            const double albedo = kext > 0. ? ksca / kext : 0.;
                            if (S.force_scattering)
                                p.W *= (-expm1(-p.taupath) * albedo);
                            else
                                p.W *= albedo;
                            p.rx += p.sint * p.kx;
                            p.ry += p.sint * p.ky;
                            p.rz += p.sint * p.kz;
                            p.cellhint = p.mint;
                            const double lum = p.W / p.lambda;
                            bool terminate = S.force_scattering
                                                 ? (lum <= 0 || (lum <= p.Lthreshold && p.nscatt >= S.min_scatt_events))
                                                 : (lum <= 0);
                            if (terminate)
                                phase = PH_LAUNCH;
                            else
                            {
                                // peelOffScattering towards the first observer (MonteCarloSimulation.cpp:784-842)
                                const DevInstrument& I = S.inst[0];
                                double costheta = p.kx * I.kx + p.ky * I.ky + p.kz * I.kz;
                                double value = fabs(p.g) > 0.95 ? meanHG(p.g, costheta) : valueHG(p.g, costheta);
                                p.pW = p.W * (0. + value * 1.);
                                p.pscatt = p.nscatt + 1;
                                p.pinst = 0;
                                w.rx = p.rx, w.ry = p.ry, w.rz = p.rz;
                                w.kx = I.kx, w.ky = I.ky, w.kz = I.kz;
                                mode = MODE_PEEL;
                                phase = PH_START;
                            }
                        }
                    }
                    // ---- whole path walked: sample the interaction optical depth (MonteCarloSimulation.cpp:696-722)
                    if (phase == PH_PASS1_DONE)
                    {
                        const double taupath = w.tau;
                        if (taupath <= 0.)
                            phase = PH_LAUNCH;  // applyBias(0): the packet cannot scatter
                        else
                        {
                            const double xi = S.path_length_bias;
                            double tau;
                            if (xi == 0.)
                                tau = exponCutoff(rng, seed, taupath);
                            else
                            {
                                tau = rngUniform(rng, seed) < xi ? rngUniform(rng, seed) * taupath : exponCutoff(rng, seed, taupath);
                                double pp = -exp(-tau) / expm1(-taupath);
                                double q = (1.0 - xi) * pp + xi / taupath;
                                p.W *= pp / q;
                            }
                            p.taupath = taupath;
                            p.target = tau;
                            w.rx = p.rx, w.ry = p.ry, w.rz = p.rz;
                            w.kx = p.kx, w.ky = p.ky, w.kz = p.kz;
                            mode = MODE_PASS2;
                            phase = PH_START;
                        }
                    }
                    // ---- new history
                    if (phase == PH_LAUNCH)
                    {
                        if (p.alive && S.any_stats) flushStatistics(S, L, p, cnt, laneIndex);
                        p.alive = false;
                        unsigned long long need = __ballot(true);
                        uint64_t history = 0;
                        bool got = false;
                        while (need && !exhausted)
                        {
                            if (poolNext >= poolEnd)
                            {
                                unsigned long long base = 0;
                                const int leader = __ffsll((long long)need) - 1;
                                if (lane == leader) base = atomicAdd(S.history_counter, (unsigned long long)PMC_HISTORY_CHUNK);
                                base = __shfl(base, leader, 64);
                                poolNext = base < count ? base : count;
                                poolEnd = (base + PMC_HISTORY_CHUNK) < count ? (base + PMC_HISTORY_CHUNK) : count;
                                if (poolNext >= poolEnd)
                                {
                                    exhausted = true;
                                    break;
                                }
                            }
                            const unsigned long long avail = poolEnd - poolNext;
                            const int rank = __popcll(need & ((1ull << lane) - 1ull));
                            const int want = __popcll(need);
                            const int take = (unsigned long long)want < avail ? want : (int)avail;
                            if (!got && rank < take)
                            {
                                history = first + poolNext + rank;
                                got = true;
                            }
                            poolNext += take;
                            need = __ballot(!got);
                        }
                        // `exhausted`, poolNext, poolEnd must stay wave-uniform: lanes outside this branch update them too
                        if (got)
                        {
                            launch(S, L, p, rng, history, seed);
                            p.alive = true;
                            cnt.histories += 1;
                            if (p.W / p.lambda > 0)
                            {
                                // peelOffEmission towards the first observer (MonteCarloSimulation.cpp:617-634)
                                const DevInstrument& I = S.inst[0];
                                p.pW = p.W;
                                p.pscatt = 0;
                                p.pinst = 0;
                                w.rx = p.rx, w.ry = p.ry, w.rz = p.rz;
                                w.kx = I.kx, w.ky = I.ky, w.kz = I.kz;
                                mode = MODE_PEEL;
                                phase = PH_START;
                            }
                            else
                                phase = PH_LAUNCH;  // zero-luminosity packet: next history
                        }
                        else
                            phase = PH_DONE;
                    }
                    // the pool variables were modified only by the lanes inside the PH_LAUNCH branch: re-broadcast
                    {
                        const unsigned long long any = __ballot(true);
                        const int src = __ffsll((long long)any) - 1;
                        unsigned long long pn = poolNext, pe = poolEnd;
                        int ex = exhausted ? 1 : 0;
                        // take the maximum over lanes (lanes that did not take part hold stale, smaller values)
                        for (int off = 32; off > 0; off >>= 1)
                        {
                            unsigned long long opn = __shfl_xor(pn, off, 64), ope = __shfl_xor(pe, off, 64);
                            int oex = __shfl_xor(ex, off, 64);
                            pn = opn > pn ? opn : pn;
                            pe = ope > pe ? ope : pe;
                            ex = oex > ex ? oex : ex;
                        }
                        (void)src;
                        poolNext = pn;
                        poolEnd = pe;
                        exhausted = ex != 0;
                    }
                    // ---- start of a walk
                    if (phase == PH_START)
                    {
                        cnt.paths += 1;
                        bool ok = startWalk<GRID>(S, L, w, (GRID == GRID_TREE) ? p.cellhint : -1);
                        if (mode == MODE_PEEL)
                        {
                            // MediumSystem::getExtinctionOpticalDepth preamble (MediumSystem.cpp:1192-1199)
                            const double lum = p.pW / p.lambda;
                            p.ptau = 0.;
                            if (lum <= 0)
                            {
                                p.ptau = INFINITY;
                                ok = false;
                            }
                            taumax = log(lum) + 745;
                        }
                        if (mode == MODE_PASS2)
                        {
                            p.mint = -1;
                            p.sint = 0.;
                        }
                        if (ok)
                            phase = PH_WALK;
                        else
                            phase = mode == MODE_PEEL ? PH_PEEL_DONE : mode == MODE_PASS1 ? PH_PASS1_DONE : PH_PASS2_DONE;
                    }
                    if (++guard > 1000000) break;  // never reached; bounds the loop for safety
                }
            }
            if (!__ballot(phase != PH_DONE)) break;

            // =========================== walk steps (convergent hot loop) ===========================
#pragma unroll 1
            for (int it = 0; it < PMC_STEPS_PER_ROUND; ++it)
            {
                if (phase == PH_WALK)
                {
                    // ---- emit the pending segment (m = w.cell, ds = w.ds)
                    const double ds = w.ds;
                    const double tau0 = w.tau, s0 = w.s;
                    if (mode == MODE_PASS2 && S.force_scattering)
                        cnt.rewalks += 1;
                    else if (mode == MODE_PEEL || ds > 0. || !S.force_scattering)
                        cnt.visits += 1;  // V: segments the reference's path holds (zero-length ones are dropped)
                    bool stop = false;
                    if (mode == MODE_PEEL)
                    {
                        // MediumSystem.cpp:1207-1219
                        w.tau += p.sext * w.dens * ds;
                        if (w.tau >= taumax)
                        {
                            p.ptau = INFINITY;
                            phase = PH_PEEL_DONE;
                            stop = true;
                        }
                    }
                    else if (ds > 0. || !S.force_scattering)
                    {
                        // SpatialGridPath::addSegment + MediumSystem.cpp:863-871
                        w.s += ds;
                        w.tau += p.sext * w.dens * ds;
                        w.lastm = w.cell;
                        if (mode == MODE_PASS2 && p.target < w.tau)
                        {
                            // findInteractionPoint (SpatialGridPath.cpp:177-196): first segment with tau > target
                            p.mint = w.cell;
                            p.nint = w.dens;
                            p.sint = s0 + ((p.target - tau0) / (w.tau - tau0)) * (w.s - s0);
                            phase = PH_PASS2_DONE;
                            stop = true;
                        }
                    }
                    if (!stop)
                    {
                        bool inside;
                        if (GRID == GRID_TREE)
                        {
                            const double step = ds + S.eps;
                            w.rx += w.kx * step;
                            w.ry += w.ky * step;
                            w.rz += w.kz * step;
                            inside = treeAdvance(S, L, w);
                        }
                        else
                            inside = cartAdvance(S, L, w);
                        if (!inside)
                        {
                            if (mode == MODE_PEEL)
                            {
                                p.ptau = w.tau;
                                phase = PH_PEEL_DONE;
                            }
                            else if (mode == MODE_PASS1)
                                phase = PH_PASS1_DONE;
                            else
                            {
                                // beyond the last segment (SpatialGridPath.cpp:198-204); non-forced: escaped
                                if (S.force_scattering && w.lastm >= 0)
                                {
                                    p.mint = w.lastm;
                                    p.sint = w.s;
                                    p.nint = (GRID == GRID_TREE) ? S.leaves[w.lastm].density : S.cell_density[w.lastm];
                                }
                                phase = PH_PASS2_DONE;
                            }
                        }
                    }
                }
            }
        }

        // ---- epilogue: statistics of the last history, privatised SED blocks, counters
        if (p.alive && S.any_stats) flushStatistics(S, L, p, cnt, laneIndex);
        __syncthreads();
        for (int i = 0; i < S.num_instruments; ++i)
        {
            const DevInstrument& I = S.inst[i];
            if (!I.include_sed) continue;
            const double* sed = L.sed + I.sed_lds_offset;
            const int nflux = I.num_components * I.num_lambda;
            for (int q = tid; q < nflux; q += blockDim.x)
                if (sed[q] != 0.) atomicAddF64(S.frames + I.sed_offset + q, sed[q]);
            if (I.record_stats)
                for (int q = tid; q < 5 * I.num_lambda; q += blockDim.x)
                    if (sed[nflux + q] != 0.) atomicAddF64(S.frames + I.wsed_offset + q, sed[nflux + q]);
        }
        unsigned long long v;
        v = waveSum(cnt.histories);
        if (lane == 0 && v) atomicAdd(S.counters + 0, v);
        v = waveSum(cnt.paths);
        if (lane == 0 && v) atomicAdd(S.counters + 1, v);
        v = waveSum(cnt.visits);
        if (lane == 0 && v) atomicAdd(S.counters + 2, v);
        v = waveSum(cnt.updates);
        if (lane == 0 && v) atomicAdd(S.counters + 3, v);
        v = waveSum(cnt.scatterings);
        if (lane == 0 && v) atomicAdd(S.counters + 4, v);
        v = waveSum(cnt.overflows);
        if (lane == 0 && v) atomicAdd(S.counters + 5, v);
        v = waveSum(cnt.rewalks);
        if (lane == 0 && v) atomicAdd(S.counters + 6, v);
    }

    // ================================================================================================
    //  single-ray tracer: the same traversal code, one lane, (m, ds) written out
    // ================================================================================================
    template<int GRID> __global__ void traceRayKernel(const int slot, double rx, double ry, double rz, double kx, double ky,
                                                      double kz, int32_t* mOut, double* dsOut, int32_t cap, int32_t* nOut)
    {
        const DevScene& S = c_scene[slot];
        extern __shared__ double lds[];
        const int tid = threadIdx.x;
        {
            const int ngrid = (GRID == GRID_TREE) ? 3 * ((1 << S.lmax) + 1) : (S.nx + 1) + (S.ny + 1) + (S.nz + 1);
            double* g = lds + S.lds_grid_off;
            if (GRID == GRID_TREE)
                for (int i = tid; i < ngrid; i += blockDim.x) g[i] = S.coord_tab[i];
            else
            {
                for (int i = tid; i <= S.nx; i += blockDim.x) g[i] = S.xv[i];
                for (int i = tid; i <= S.ny; i += blockDim.x) g[(S.nx + 1) + i] = S.yv[i];
                for (int i = tid; i <= S.nz; i += blockDim.x) g[(S.nx + 1) + (S.ny + 1) + i] = S.zv[i];
            }
        }
        __syncthreads();
        if (tid != 0) return;
        Lds L;
        L.grid = lds + S.lds_grid_off;
        L.tabn = (1 << S.lmax) + 1;
        Walk w;
        w.rx = rx, w.ry = ry, w.rz = rz;
        w.kx = kx, w.ky = ky, w.kz = kz;
        int n = 0;
        w.tau = 0., w.s = 0., w.lastm = -1;
        double cumds;
        bool ok = moveInside(S, w, cumds);
        if (ok)
        {
            if (cumds > 0.)
            {
                if (n < cap)
                {
                    mOut[n] = -1;
                    dsOut[n] = cumds;
                }
                ++n;
            }
            if (GRID == GRID_CART)
            {
                w.ci = locateClip(L.grid, S.nx + 1, w.rx);
                w.cj = locateClip(L.grid + (S.nx + 1), S.ny + 1, w.ry);
                w.ck = locateClip(L.grid + (S.nx + 1) + (S.ny + 1), S.nz + 1, w.rz);
                cartEnter(S, L, w);
            }
            else
            {
                int m = topDown(S, L, w.rx, w.ry, w.rz);
                treeEnter<false>(S, L, w, m, 0);
            }
            bool inside = true;
            int guard = 0;
            while (inside && guard++ < 100000)
            {
                if (n < cap)
                {
                    mOut[n] = w.cell;
                    dsOut[n] = w.ds;
                }
                ++n;
                if (GRID == GRID_TREE)
                {
                    const double step = w.ds + S.eps;
                    w.rx += w.kx * step;
                    w.ry += w.ky * step;
                    w.rz += w.kz * step;
                    inside = treeAdvance(S, L, w);
                }
                else
                    inside = cartAdvance(S, L, w);
            }
        }
        *nOut = n;
    }
}

// ---------------------------------------------------------------------------------------------------
// launch wrappers used by pmc_api.hip

extern "C" hipError_t pmcUploadScene(int slot, const DevScene* scene, hipStream_t stream)
{
    return hipMemcpyToSymbolAsync(HIP_SYMBOL(c_scene), scene, sizeof(DevScene), size_t(slot) * sizeof(DevScene),
                                  hipMemcpyHostToDevice, stream);
}

extern "C" hipError_t pmcLaunchPrimary(int slot, int gridKind, uint64_t first, uint64_t count, uint64_t seed, int grid,
                                       int block, size_t ldsBytes, hipStream_t stream)
{
    hipError_t e;
    if (gridKind == PMC_GRID_OCTREE)
    {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&primaryEmissionKernel<GRID_TREE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(primaryEmissionKernel<GRID_TREE>, dim3(grid), dim3(block), ldsBytes, stream, slot, first, count, seed);
    }
    else
    {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&primaryEmissionKernel<GRID_CART>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(primaryEmissionKernel<GRID_CART>, dim3(grid), dim3(block), ldsBytes, stream, slot, first, count, seed);
    }
    return hipGetLastError();
}

extern "C" hipError_t pmcLaunchTrace(int slot, int gridKind, const double r[3], const double k[3], int32_t* m, double* ds,
                                     int32_t cap, int32_t* n, size_t ldsBytes, hipStream_t stream)
{
    hipError_t e;
    if (gridKind == PMC_GRID_OCTREE)
    {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&traceRayKernel<GRID_TREE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(traceRayKernel<GRID_TREE>, dim3(1), dim3(64), ldsBytes, stream, slot, r[0], r[1], r[2], k[0], k[1],
                           k[2], m, ds, cap, n);
    }
    else
    {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&traceRayKernel<GRID_CART>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(traceRayKernel<GRID_CART>, dim3(1), dim3(64), ldsBytes, stream, slot, r[0], r[1], r[2], k[0], k[1],
                           k[2], m, ds, cap, n);
    }
    return hipGetLastError();
}
