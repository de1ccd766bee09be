// pmc_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) of the primary-emission photon loop.
//
// The life cycle of a photon history (MonteCarloSimulation.cpp:538-613 performLifeCycle) is a chain of grid walks
// separated by short "transitions":
//
//      launch -> [peel-off walk per observer] -> pass-1 walk (tau of the whole path) -> sample tau ->
//      pass-2 walk (same path again, up to the sampled tau) -> interaction -> [peel-off walk per observer] ->
//      scatter -> pass-1 walk ...
//
// >95 % of the work is in the walks: a latency-bound pointer chase through the cell records in HBM/L2 with one
// dependent load per step.  What hides that latency is occupancy, so the loop is split into TWO kernels that run
// alternately over a pool of `num_slots` concurrently live histories whose state lives in HBM (struct of arrays):
//
//   walkKernel        lean (few registers -> many waves per SIMD).  Persistent wavefronts, one walk per lane.  Lanes
//                     whose walk has ended write the result, and when enough lanes of the wave are idle the wave
//                     fetches the next slots from a global cursor and starts their walks (ballot/compaction of
//                     terminated walks).  No random numbers, no transcendental functions except none.
//   transitionKernel  one lane per slot: consumes the walk result, does the divergent physics (detection with
//                     atomics, sampling, HG scattering, launching the next history from the global history cursor)
//                     and leaves the next walk task in the slot.  All random draws happen here, from the slot's own
//                     Philox stream keyed by (seed, history index) (include/pmc_philox.h).
//
// One generation = walkKernel + transitionKernel; a history needs about 3 generations per scattering event.  The
// host enqueues generations until no slot is alive (pmc_api.hip).
//
// The reference stores the whole path (<= 1000 x 40 B per thread) and binary-searches the interaction point
// (SpatialGridPath.cpp:164-206).  Here the path is walked twice with bit-identical arithmetic instead: pass 1 yields
// tau_path, pass 2 stops in the segment whose cumulative tau exceeds the sampled value and interpolates exactly as
// findInteractionPoint does.
//
// Arithmetic is IEEE double with contraction off (-ffp-contract=off): the reference build has no FMA, and the
// traversal must produce the same (m, ds) sequence bit for bit (pmc_trace_ray).
//
// Octree traversal (TreeSpatialGrid.cpp:132-217): the reference hops through per-wall neighbour lists of heap
// nodes.  Here a cell is ONE 128-byte record (LeafRec) holding a dyadic box code, the density and four links per
// wall; wall coordinates come from a per-axis table staged in LDS (exactly the reference's doubles).  A link leads
// to the neighbour leaf covering that quadrant of the wall (same size, coarser or one level finer) or to an internal
// node (two or more levels finer), from which the position descends with the reference's child rule
// (OctTreeNode.cpp:36-41).  This gives the reference's answer whenever the new position lies strictly inside one
// neighbour; in every other case (position on a shared boundary, corner overshoot, rounding) the code falls back to
// the literal reference algorithm on the neighbour lists kept in HBM in the reference's order, followed by the
// reference's top-down search and next-after escape.
//
// Exit distances: the reference evaluates three divisions (wall - r)/k per step and takes the smallest.  The kernel
// first orders the three candidates with reciprocal multiplies (error < 4e-16 relative); if the smallest is separated
// from the others by more than 1e-14 relative it computes only that one quotient with a true IEEE division, otherwise
// all three.  The selected wall and the emitted ds are therefore bit-identical to the reference's.

#include "pmc_device.h"
#include "../../include/pmc_philox.h"
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>

#ifndef PMC_WALK_REFILL
    #define PMC_WALK_REFILL 8  // idle lanes in a wave that trigger a refill
#endif
#ifndef PMC_WALK_STEPS
    #define PMC_WALK_STEPS 4  // steps between two refill checks
#endif
#ifndef PMC_TRANSITION_BLOCK
    #define PMC_TRANSITION_BLOCK 1024  // lanes per workgroup of the transition kernel (regrouped by event type)
#endif
#define PMC_TASK_CHUNK 128   // slots a wave takes from the global cursor at a time

// the scene of every live context, in constant memory: all accesses are scalar loads
__constant__ DevScene c_scene[PMC_MAX_CONTEXTS];

namespace
{
    enum Mode : int { MODE_PASS1 = 0, MODE_PASS2 = 1, MODE_PEEL = 2, MODE_NONE = 3 };
    enum Grid : int { GRID_CART = PMC_GRID_CARTESIAN, GRID_TREE = PMC_GRID_OCTREE };
    constexpr int MODE_ALIVE = 1 << 5;

    // ------------------------------------------------------------------------------------------------
    struct Rng
    {
        uint32_t h0, h1, block, have;
        double spare;
    };
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
        double rx, ry, rz;     // generator position (PathSegmentGenerator::_rx..)
        double kx, ky, kz;     // direction
        double ikx, iky, ikz;  // reciprocals for the ordering of the exit distances (0: |k| <= 1e-15)
        double tau, s;         // cumulative optical depth and path length of the segments emitted so far
        double ds;             // length of the pending segment (exit distance of the current cell)
        double dens;           // number density of the current cell
        uint64_t code;         // octree: box code of the current cell
        int cell;              // octree: leaf m; Cartesian: m
        int ci, cj, ck;        // Cartesian indices
        int axis;              // exit axis 0,1,2 of the pending segment
        int lastm;             // cell of the last segment added to the path (ds > 0)
    };

    struct GridLds
    {
        const double* grid;  // octree: coordinate tables [3][tabn]; Cartesian: xv | yv | zv
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

    // selection of the exit wall: returns the axis and the exact exit distance (TreeSpatialGrid.cpp:160-185 /
    // CartesianSpatialGrid.cpp:112-119).  cartesianTies selects the Cartesian generator's tie rule.
    template<bool CARTESIAN_TIES>
    __device__ __forceinline__ void exitDistance(const Walk& w, double xnext, double ynext, double znext, double& ds, int& axis)
    {
        const double dx = xnext - w.rx, dy = ynext - w.ry, dz = znext - w.rz;
        // approximate quotients (DBL_MAX where the reference ignores the axis)
        const double ax = (w.ikx != 0.) ? dx * w.ikx : DBL_MAX;
        const double ay = (w.iky != 0.) ? dy * w.iky : DBL_MAX;
        const double az = (w.ikz != 0.) ? dz * w.ikz : DBL_MAX;
        const double mxy = 1e-14 * (fabs(ax) + fabs(ay)), mxz = 1e-14 * (fabs(ax) + fabs(az)), myz = 1e-14 * (fabs(ay) + fabs(az));
        const bool winx = ax < ay - mxy && ax < az - mxz;
        const bool winy = ay < ax - mxy && ay < az - myz;
        const bool winz = az < ax - mxz && az < ay - myz;
        if (winx || winy || winz)
        {
            double num = dz, den = w.kz;
            axis = 2;
            if (winy)
            {
                num = dy;
                den = w.ky;
                axis = 1;
            }
            if (winx)
            {
                num = dx;
                den = w.kx;
                axis = 0;
            }
            ds = num / den;  // ONE exact IEEE division
        }
        else
        {
            // near tie: the reference's three divisions and its tie order
            const double dsx = (fabs(w.kx) > 1e-15) ? dx / w.kx : DBL_MAX;
            const double dsy = (fabs(w.ky) > 1e-15) ? dy / w.ky : DBL_MAX;
            const double dsz = (fabs(w.kz) > 1e-15) ? dz / w.kz : DBL_MAX;
            if (dsx <= dsy && dsx <= dsz)
            {
                ds = dsx;
                axis = 0;
            }
            else if ((CARTESIAN_TIES ? (dsy < dsx) : (dsy <= dsx)) && dsy <= dsz)
            {
                ds = dsy;
                axis = 1;
            }
            else
            {
                ds = dsz;
                axis = 2;
            }
        }
    }

    __device__ __forceinline__ void setDirection(Walk& w, double kx, double ky, double kz)
    {
        w.kx = kx, w.ky = ky, w.kz = kz;
        w.ikx = (fabs(kx) > 1e-15) ? 1. / kx : 0.;
        w.iky = (fabs(ky) > 1e-15) ? 1. / ky : 0.;
        w.ikz = (fabs(kz) > 1e-15) ? 1. / kz : 0.;
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

    // std::nextafter(x, negative ? -DBL_MAX : DBL_MAX) for finite |x| < DBL_MAX, without the library call
    __device__ __forceinline__ double nextAfterToward(double x, bool negative)
    {
        if (x == 0.) return negative ? -4.9406564584124654e-324 : 4.9406564584124654e-324;
        long long b = __double_as_longlong(x);
        b += ((x > 0.) != negative) ? 1 : -1;
        return __longlong_as_double(b);
    }

    // TreeNode::leafChild from the given link downwards (TreeNode.cpp:66-76, OctTreeNode.cpp:36-41)
    __device__ __forceinline__ int descend(const DevScene& S, const GridLds& L, int link, double x, double y, double z)
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
    __device__ __forceinline__ int topDown(const DevScene& S, const GridLds& L, double x, double y, double z)
    {
        if (!(x >= S.gx0 && x <= S.gx1 && y >= S.gy0 && y <= S.gy1 && z >= S.gz0 && z <= S.gz1)) return -1;
        return descend(S, L, S.root_link, x, y, z);
    }
    __device__ __forceinline__ bool leafContains(const DevScene& S, const GridLds& L, int m, double x, double y, double z)
    {
        int fx, fy, fz, size;
        decodeBox(S.leaves[m].code, S.lmax, fx, fy, fz, size);
        return x >= L.grid[fx] && x <= L.grid[fx + size] && y >= L.grid[L.tabn + fy] && y <= L.grid[L.tabn + fy + size]
               && z >= L.grid[2 * L.tabn + fz] && z <= L.grid[2 * L.tabn + fz + size];
    }
    // TreeNode::neighbor on the reference's neighbour list (TreeNode.cpp:103-112)
    __device__ __forceinline__ int listNeighbor(const DevScene& S, const GridLds& L, int m, int wall, double x, double y, double z)
    {
        int b = S.nbr_start[6 * (int64_t)m + wall], e = S.nbr_start[6 * (int64_t)m + wall + 1];
        for (int q = b; q < e; ++q)
        {
            int cand = S.nbr_list[q];
            if (leafContains(S, L, cand, x, y, z)) return cand;
        }
        return -1;
    }

    // loads the record of leaf m and prepares the pending segment: exit distance and the links through the exit wall
    // (TreeSpatialGrid.cpp:160-185).  With CHECK, first verifies that the position lies inside the closed box and
    // strictly between the transverse walls (then no other neighbour can contain it); returns false otherwise.
    template<bool CHECK> __device__ __forceinline__ bool treeEnter(const DevScene& S, const GridLds& L, Walk& w, int m, int axis)
    {
        const char* rec = reinterpret_cast<const char*>(S.leaves + m);
        const bool nx = w.kx < 0.0, ny = w.ky < 0.0, nz = w.kz < 0.0;
        // 16 bytes: box code + density.  (Uncoalesced gathers are priced per byte returned to the lane -- the L1
        // return path, not HBM, bounds this kernel -- so the links are NOT fetched here: treeAdvance reads the one
        // 4-byte link it needs once the exit wall and its quadrant are known; that read hits the line fetched here.)
        const uint4 q0 = *reinterpret_cast<const uint4*>(rec);
        const uint64_t code = ((uint64_t)q0.y << 32) | q0.x;
        int fx, fy, fz, size;
        decodeBox(code, S.lmax, fx, fy, fz, size);
        const double X0 = L.grid[fx], X1 = L.grid[fx + size];
        const double Y0 = L.grid[L.tabn + fy], Y1 = L.grid[L.tabn + fy + size];
        const double Z0 = L.grid[2 * L.tabn + fz], Z1 = L.grid[2 * L.tabn + fz + size];
        if (CHECK)
        {
            // (values first, then selects on values: keeps the Walk fields in registers)
            const double rx = w.rx, ry = w.ry, rz = w.rz;
            const bool inside = rx >= X0 && rx <= X1 && ry >= Y0 && ry <= Y1 && rz >= Z0 && rz <= Z1;
            const bool tx = rx == X0 || rx == X1, ty = ry == Y0 || ry == Y1, tz = rz == Z0 || rz == Z1;
            // a tie in a transverse coordinate: another neighbour's closed box may contain the position as well
            const bool tie = (axis != 0 && tx) || (axis != 1 && ty) || (axis != 2 && tz);
            if (!inside || tie) return false;
        }
        double ds;
        int ax;
        exitDistance<false>(w, nx ? X0 : X1, ny ? Y0 : Y1, nz ? Z0 : Z1, ds, ax);
        w.ds = ds;
        w.axis = ax;
        w.code = code;
        w.cell = m;
        w.dens = __longlong_as_double(((long long)q0.w << 32) | q0.z);
        return true;
    }

    // after the pending segment has been emitted and the position advanced: find the next cell
    // (TreeSpatialGrid.cpp:186-207).  Returns false when the path leaves the grid (State::Outside).
    __device__ __forceinline__ bool treeAdvance(const DevScene& S, const GridLds& L, Walk& w)
    {
        const int old = w.cell;
        const int axis = w.axis;
        // quadrant of the exit wall: the new position against the centre of the old cell in the transverse axes
        int fx, fy, fz, size;
        decodeBox(w.code, S.lmax, fx, fy, fz, size);
        const int half = size >> 1;
        // (values first, then selects on values: keeps the Walk fields in registers)
        const double rx = w.rx, ry = w.ry, rz = w.rz;
        // upper-half flags against the cell centre; a finest-level cell (half == 0) has four equal links
        const int ux = rx < L.grid[fx + half] ? 0 : 1;
        const int uy = ry < L.grid[L.tabn + fy + half] ? 0 : 1;
        const int uz = rz < L.grid[2 * L.tabn + fz + half] ? 0 : 1;
        // transverse axes in x, y, z order: axis 0 -> (y, z), axis 1 -> (x, z), axis 2 -> (x, y)
        const int u1 = axis == 0 ? uy : ux;
        const int u2 = axis == 2 ? uy : uz;
        const int q = half == 0 ? 0 : u1 + 2 * u2;
        bool neg = w.kz < 0.0;
        if (axis == 1) neg = w.ky < 0.0;
        if (axis == 0) neg = w.kx < 0.0;
        const int wall = 2 * axis + (neg ? 0 : 1);
        const int link = S.leaves[old].link[wall][q];  // 4 bytes from the line of the cell just left
        int next = link;
        if (next <= -2) next = descend(S, L, next, w.rx, w.ry, w.rz);
        if (next >= 0 && treeEnter<true>(S, L, w, next, axis)) return true;  // the common case

        // ---- everything else: the reference algorithm, literally
        next = (link == PMC_LINK_NONE) ? -1 : listNeighbor(S, L, old, wall, w.rx, w.ry, w.rz);
        if (next < 0) next = topDown(S, L, w.rx, w.ry, w.rz);
        if (next == old)
        {
            // PathSegmentGenerator::propagateToNextAfter (PathSegmentGenerator.hpp:148-153)
            w.rx = nextAfterToward(w.rx, w.kx < 0.);
            w.ry = nextAfterToward(w.ry, w.ky < 0.);
            w.rz = nextAfterToward(w.rz, w.kz < 0.);
            next = topDown(S, L, w.rx, w.ry, w.rz);
        }
        if (next < 0 || next == old) return false;
        treeEnter<false>(S, L, w, next, axis);
        return true;
    }

    // ------------------------------------------------------------------------------------------------
    // Cartesian helpers (CartesianSpatialGrid.cpp:87-163)

    __device__ __forceinline__ void cartEnter(const DevScene& S, const GridLds& L, Walk& w)
    {
        const double* xv = L.grid;
        const double* yv = L.grid + (S.nx + 1);
        const double* zv = yv + (S.ny + 1);
        const int m = w.ck + S.nz * w.cj + S.nz * S.ny * w.ci;
        const double xE = (w.kx < 0.0) ? xv[w.ci] : xv[w.ci + 1];
        const double yE = (w.ky < 0.0) ? yv[w.cj] : yv[w.cj + 1];
        const double zE = (w.kz < 0.0) ? zv[w.ck] : zv[w.ck + 1];
        exitDistance<true>(w, xE, yE, zE, w.ds, w.axis);
        w.cell = m;
        w.dens = S.cell_density[m];
    }
    __device__ __forceinline__ bool cartAdvance(const DevScene& S, const GridLds& L, Walk& w)
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
    // has been added to w.s.  located receives the start cell if the start position itself lies inside the grid.
    template<int GRID>
    __device__ __forceinline__ bool startWalk(const DevScene& S, const GridLds& L, Walk& w, int hint, int& located)
    {
        w.tau = 0.;
        w.s = 0.;
        w.lastm = -1;
        located = -1;
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
            if (cumds == 0.) located = m;
            return true;
        }
    }

    __device__ __forceinline__ unsigned long long waveSum(uint32_t value)
    {
        unsigned long long v = value;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        return v;
    }

    template<int GRID> __device__ __forceinline__ void stageGrid(const DevScene& S, double* g, int tid, int nthreads)
    {
        if (GRID == GRID_TREE)
        {
            const int ngrid = 3 * ((1 << S.lmax) + 1);
            for (int i = tid; i < ngrid; i += nthreads) g[i] = S.coord_tab[i];
        }
        else
        {
            for (int i = tid; i <= S.nx; i += nthreads) g[i] = S.xv[i];
            for (int i = tid; i <= S.ny; i += nthreads) g[(S.nx + 1) + i] = S.yv[i];
            for (int i = tid; i <= S.nz; i += nthreads) g[(S.nx + 1) + (S.ny + 1) + i] = S.zv[i];
        }
    }

    // ================================================================================================
    //  walk kernel
    // ================================================================================================
    template<int GRID> __global__ __launch_bounds__(256) void walkKernel(const int sceneSlot, const int numSlots)
    {
        const DevScene& S = c_scene[sceneSlot];
        extern __shared__ double lds[];
        const int tid = threadIdx.x;
        const int lane = tid & 63;
        stageGrid<GRID>(S, lds, tid, blockDim.x);
        double* ldsExt = lds + S.lds_grid_len;
        if (S.dust_in_lds)
            for (int i = tid; i < S.num_lambda; i += blockDim.x) ldsExt[i] = S.sigma_ext[i];
        __syncthreads();
        GridLds L;
        L.grid = lds;
        L.tabn = (1 << S.lmax) + 1;
        const double* sigmaExt = S.dust_in_lds ? ldsExt : S.sigma_ext;
        const SlotArrays& A = S.slots;

        Walk w;
        w.cell = -1;
        w.ds = 0.;
        int slot = -1;
        int mode = MODE_NONE;
        double sext = 0., target = 0.;
        uint32_t visits = 0, rewalks = 0, paths = 0;
        bool exhausted = false;
        unsigned long long poolNext = 0, poolEnd = 0;  // wave-uniform
#ifdef PMC_PROFILE
        long long profRefill = 0, profStep = 0, profT0 = clock64();
        unsigned long long profRefills = 0, profWaveSteps = 0, profLaneSteps = 0;
#endif

        while (true)
        {
            // ---------------- refill: idle lanes fetch the next slots and start their walks
            const unsigned long long idle = __ballot(slot < 0);
#ifdef PMC_PROFILE
            {
                const long long now = clock64();
                profStep += now - profT0;
                profT0 = now;
            }
#endif
            const int nidle = __popcll(idle);
            if (nidle && !exhausted && (nidle >= PMC_WALK_REFILL || nidle == 64))
            {
                // slots are handed out from a wave-local pool that is refilled in chunks from the global cursor:
                // one device-scope atomic per PMC_TASK_CHUNK walks (a single word saturates near 90 atomics/us)
                if (poolNext >= poolEnd)
                {
                    unsigned long long got = 0;
                    if (lane == 0) got = atomicAdd(S.counters + PMC_CTR_TASK, (unsigned long long)PMC_TASK_CHUNK);
                    got = __shfl(got, 0, 64);
                    poolNext = got;
                    poolEnd = got + PMC_TASK_CHUNK;
                    if (poolEnd > (unsigned long long)numSlots) poolEnd = (unsigned long long)numSlots;
                    if (poolNext >= poolEnd)
                    {
                        poolNext = poolEnd = 0;
                        exhausted = true;
                    }
                }
                const unsigned long long base = poolNext;
                const unsigned long long avail = poolEnd - poolNext;
                poolNext += (unsigned long long)nidle < avail ? (unsigned long long)nidle : avail;
                if (slot < 0 && !exhausted)
                {
                    const unsigned long long rank = __popcll(idle & ((1ull << lane) - 1ull));
                    if (rank < avail)
                    {
                        const unsigned long long t = base + rank;
                        const int sl = (int)t;
                        const int mw = A.mode[sl];
                        if ((mw & MODE_ALIVE) && (mw & 3) != MODE_NONE)
                        {
                            slot = sl;
                            mode = mw & 3;
                            w.rx = A.rx[sl], w.ry = A.ry[sl], w.rz = A.rz[sl];
                            if (mode == MODE_PEEL)
                            {
                                const DevInstrument& I = S.inst[(mw >> 2) & 7];
                                setDirection(w, I.kx, I.ky, I.kz);
                            }
                            else
                                setDirection(w, A.kx[sl], A.ky[sl], A.kz[sl]);
                            sext = sigmaExt[A.dustIndex[sl]];
                            target = A.target[sl];
                            paths += 1;
                            int located;
                            const bool ok = startWalk<GRID>(S, L, w, (GRID == GRID_TREE) ? A.cellhint[sl] : -1, located);
                            if (GRID == GRID_TREE && located >= 0) A.cellhint[sl] = located;
                            bool dead = !ok;
                            if (mode == MODE_PEEL && target == -INFINITY)
                            {
                                // zero-luminosity peel-off packet (MediumSystem.cpp:1195-1196)
                                A.tau[sl] = INFINITY;
                                dead = true;
                            }
                            else if (!ok)
                            {
                                A.tau[sl] = 0.;
                                if (mode == MODE_PASS2) A.mint[sl] = -1;
                            }
                            if (dead) slot = -1;
                        }
                    }
                }
            }
#ifdef PMC_PROFILE
            {
                const long long now = clock64();
                profRefill += now - profT0;
                profT0 = now;
                if (nidle && (nidle >= PMC_WALK_REFILL || nidle == 64)) profRefills += 1;
            }
#endif
            if (!__ballot(slot >= 0))
            {
                if (exhausted) break;
                continue;
            }

            // ---------------- walk steps (convergent hot loop)
#pragma unroll 1
            for (int it = 0; it < PMC_WALK_STEPS; ++it)
            {
#ifdef PMC_PROFILE
                profWaveSteps += 1;
                profLaneSteps += __popcll(__ballot(slot >= 0));
#endif
                if (slot >= 0)
                {
                    // ---- emit the pending segment (m = w.cell, ds = w.ds)
                    const double ds = w.ds;
                    const double tau0 = w.tau, s0 = w.s;
                    bool done = false;
                    if (mode == MODE_PEEL)
                    {
                        // MediumSystem.cpp:1207-1219
                        visits += 1;
                        w.tau += sext * w.dens * ds;
                        if (w.tau >= target)  // target = taumax
                        {
                            A.tau[slot] = INFINITY;
                            done = true;
                        }
                    }
                    else if (ds > 0. || !S.force_scattering)
                    {
                        // SpatialGridPath::addSegment + MediumSystem.cpp:863-871 (forced) / :988-1008 (non-forced)
                        if (mode == MODE_PASS2 && S.force_scattering)
                            rewalks += 1;
                        else
                            visits += 1;
                        w.s += ds;
                        w.tau += sext * w.dens * ds;
                        w.lastm = w.cell;
                        if (mode == MODE_PASS2 && target < w.tau)
                        {
                            // findInteractionPoint (SpatialGridPath.cpp:177-196): first segment with tau > target
                            A.mint[slot] = w.cell;
                            A.nint[slot] = w.dens;
                            A.sint[slot] = s0 + ((target - tau0) / (w.tau - tau0)) * (w.s - s0);
                            done = true;
                        }
                    }
                    if (!done)
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
                            if (mode != MODE_PASS2)
                                A.tau[slot] = w.tau;
                            else if (S.force_scattering && w.lastm >= 0)
                            {
                                // beyond the last segment (SpatialGridPath.cpp:198-204)
                                A.mint[slot] = w.lastm;
                                A.sint[slot] = w.s;
                                A.nint[slot] = (GRID == GRID_TREE) ? S.leaves[w.lastm].density : S.cell_density[w.lastm];
                            }
                            else
                                A.mint[slot] = -1;  // non-forced: the packet escapes
                            done = true;
                        }
                    }
                    if (done) slot = -1;
                }
            }
        }
#ifdef PMC_PROFILE
        if (lane == 0)
        {
            atomicAdd(S.counters + 11, (unsigned long long)profRefill);
            atomicAdd(S.counters + 12, (unsigned long long)profStep);
            atomicAdd(S.counters + 13, profRefills);
            atomicAdd(S.counters + 14, profWaveSteps);
            atomicAdd(S.counters + 15, profLaneSteps);
        }
#endif
        unsigned long long v;
        v = waveSum(paths);
        if (lane == 0 && v) atomicAdd(S.counters + 1, v);
        v = waveSum(visits);
        if (lane == 0 && v) atomicAdd(S.counters + 2, v);
        v = waveSum(rewalks);
        if (lane == 0 && v) atomicAdd(S.counters + 6, v);
    }

#include "pmc_transition.inc"

    // ================================================================================================
    //  single-ray tracer: the same traversal code, one lane, (m, ds) written out
    // ================================================================================================
    template<int GRID> __global__ void traceRayKernel(const int sceneSlot, double rx, double ry, double rz, double kx, double ky,
                                                      double kz, int32_t* mOut, double* dsOut, int32_t cap, int32_t* nOut)
    {
        const DevScene& S = c_scene[sceneSlot];
        extern __shared__ double lds[];
        const int tid = threadIdx.x;
        stageGrid<GRID>(S, lds, tid, blockDim.x);
        __syncthreads();
        if (tid != 0) return;
        GridLds L;
        L.grid = lds;
        L.tabn = (1 << S.lmax) + 1;
        Walk w;
        w.rx = rx, w.ry = ry, w.rz = rz;
        setDirection(w, kx, ky, kz);
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

extern "C" hipError_t pmcConfigureKernels(size_t walkLds, size_t transitionLds)
{
    hipError_t e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&walkKernel<GRID_TREE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)walkLds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&walkKernel<GRID_CART>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)walkLds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&transitionKernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)transitionLds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&traceRayKernel<GRID_TREE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)walkLds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&traceRayKernel<GRID_CART>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)walkLds);
}

extern "C" int pmcWalkBlocksPerCU(int gridKind, int block, size_t ldsBytes)
{
    int n = 0;
    hipError_t e;
    if (gridKind == PMC_GRID_OCTREE)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&walkKernel<GRID_TREE>), block, ldsBytes);
    else
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&walkKernel<GRID_CART>), block, ldsBytes);
    return e == hipSuccess ? n : 0;
}

extern "C" hipError_t pmcLaunchWalk(int slot, int gridKind, int numSlots, int grid, int block, size_t ldsBytes, hipStream_t stream)
{
    if (gridKind == PMC_GRID_OCTREE)
        hipLaunchKernelGGL(walkKernel<GRID_TREE>, dim3(grid), dim3(block), ldsBytes, stream, slot, numSlots);
    else
        hipLaunchKernelGGL(walkKernel<GRID_CART>, dim3(grid), dim3(block), ldsBytes, stream, slot, numSlots);
    return hipGetLastError();
}

extern "C" hipError_t pmcLaunchTransition(int slot, int numSlots, uint64_t first, uint64_t count, uint64_t seed, int initial,
                                          size_t ldsBytes, hipStream_t stream)
{
    const int block = PMC_TRANSITION_BLOCK;
    const int grid = (numSlots + block - 1) / block;
    hipLaunchKernelGGL(transitionKernel, dim3(grid), dim3(block), ldsBytes, stream, slot, numSlots, first, count, seed, initial);
    return hipGetLastError();
}

extern "C" hipError_t pmcLaunchTrace(int slot, int gridKind, const double r[3], const double k[3], int32_t* m, double* ds,
                                     int32_t cap, int32_t* n, size_t ldsBytes, hipStream_t stream)
{
    if (gridKind == PMC_GRID_OCTREE)
        hipLaunchKernelGGL(traceRayKernel<GRID_TREE>, dim3(1), dim3(64), ldsBytes, stream, slot, r[0], r[1], r[2], k[0], k[1],
                           k[2], m, ds, cap, n);
    else
        hipLaunchKernelGGL(traceRayKernel<GRID_CART>, dim3(1), dim3(64), ldsBytes, stream, slot, r[0], r[1], r[2], k[0], k[1],
                           k[2], m, ds, cap, n);
    return hipGetLastError();
}
