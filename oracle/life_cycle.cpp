// oracle/life_cycle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement, in scalar double precision, of SKIRT 9's primary-emission photon life cycle over the POD
// scene of include/pmc.h.  It exists to CHECK the HIP engine; only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may build, load or call it.  The product (skirt9_amd/) never links it.
//
// Parity status: PINNED.  With the mt19937_64 stream continued after setup (rng_kind 0) this code reproduces the
// unmodified reference (oracle/_ref, single thread) bit for bit: every FITS frame, the SED and the statistics of
// tests/golden/cfg1*, cfg2* (tests/test_oracle_golden.py).  With rng_kind 1 it consumes the engine's per-history
// Philox streams (include/pmc_philox.h) in the same call order, which is what the GPU results are compared with.
//
// Each function cites the reference code it follows (paths relative to the SKIRT 9 tree).

#include "../include/pmc.h"
#include "../include/pmc_layout.h"
#include "../include/pmc_philox.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <random>
#include <tuple>
#include <vector>

namespace
{
    // ------------------------------------------------------------------ random streams

    struct Rng
    {
        virtual ~Rng() {}
        virtual double uniform() = 0;
        virtual void beginHistory(uint64_t /*history*/) {}
    };

    // Random::uniform with the parent-thread generator (Random.cpp:18-54,70-73), continued after `skip` draws
    struct MtRng : Rng
    {
        std::mt19937_64 generator;
        std::uniform_real_distribution<double> distribution{std::nextafter(0., 1.), 1.};
        MtRng(int seed, unsigned long long skip)
        {
            std::seed_seq seedseq{979364188u + seed, 871244425u + seed, 1693909487u + seed, 1290454318u + seed,
                                  210509498u + seed, 542237529u + seed, 3429911442u + seed, 3321294726u + seed};
            generator.seed(seedseq);
            generator.discard(skip);
        }
        double uniform() override { return distribution(generator); }
    };

    struct PhiloxRng : Rng
    {
        uint64_t seed;
        pmc_rng state;
        explicit PhiloxRng(uint64_t s) : seed(s) { pmc_rng_init(&state, s, 0); }
        void beginHistory(uint64_t history) override { pmc_rng_init(&state, seed, history); }
        double uniform() override { return pmc_rng_uniform(&state); }
    };

    // ------------------------------------------------------------------ small helpers (NR.hpp:130-190,328-362)

    int locateBasic(const double* xv, double x, int n)
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
    int locate(const double* xv, int n, double x)
    {
        if (x == xv[n - 1]) return n - 2;
        return locateBasic(xv, x, n);
    }
    int locateClip(const double* xv, int n, double x)
    {
        if (x < xv[0]) return 0;
        return locateBasic(xv, x, n - 1);
    }
    double interpolateLinLin(double x, double x1, double x2, double f1, double f2)
    {
        return f1 + ((x - x1) / (x2 - x1)) * (f2 - f1);
    }
    double interpolateLogLog(double x, double x1, double x2, double f1, double f2)
    {
        if (f1 <= 0 || f2 <= 0)
        {
            if (x == x1) return f1;
            if (x == x2) return f2;
            return 0;
        }
        return f1 * exp(log(x / x1) / log(x2 / x1) * (log(f2 / f1)));
    }
    // SpecialFunctions::gexp (SpecialFunctions.cpp:822-836)
    double gexp(double p, double x)
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

    struct V3
    {
        double x, y, z;
    };

    // SpecialFunctions::LambertW1 (SKIRT/utils/SpecialFunctions.cpp:578-625)
    double lambertW1(double z)
    {
        const double eps = 1.0e-12;
        const double em1 = 0.3678794411714423215955237701614608;
        const double c[12] = {-1.0,
                                     2.331643981597124203363536062168,
                                     -1.812187885639363490240191647568,
                                     1.936631114492359755363277457668,
                                     -2.353551201881614516821543561516,
                                     3.066858901050631912893148922704,
                                     -4.175335600258177138854984177460,
                                     5.858023729874774148815053846119,
                                     -8.401032217523977370984161688514,
                                     12.250753501314460424,
                                     -18.100697012472442755,
                                     27.029044799010561650};
        if (z == 0.0) return -DBL_MAX;
        double q = z + em1;
        double r = -sqrt(q);
        double t8 = c[8] + r * (c[9] + r * (c[10] + r * c[11]));
        double t5 = c[5] + r * (c[6] + r * (c[7] + r * t8));
        double t1 = c[1] + r * (c[2] + r * (c[3] + r * (c[4] + r * t5)));
        double w0 = c[0] + r * t1;
        if (q < 3.0e-3) return w0;
        double w;
        if (z < -1e-6)
            w = w0;
        else
        {
            double l1 = log(-z);
            double l2 = log(-l1);
            w = l1 - l2 + l2 / l1;
        }
        for (int i = 0; i < 10; i++)
        {
            double e = exp(w);
            double t = w * e - z;
            double p = w + 1.0;
            t /= e * p - 0.5 * (p + 1.0) * t / p;
            w -= t;
            if (fabs(t) < eps * (1.0 + fabs(w))) return w;
        }
        return w;  // (the reference raises "No convergence": unreachable for arguments in [-1/e, 0])
    }

    // Direction(theta, phi) (Direction.cpp:11-38)
    V3 directionFromAngles(double theta, double phi)
    {
        const double eps = 1e-8;
        if (theta <= eps) return V3{0, 0, 1};
        if (theta >= M_PI - eps) return V3{0, 0, -1};
        double sintheta = sin(theta);
        return V3{sintheta * cos(phi), sintheta * sin(phi), cos(theta)};
    }
    // Random::direction() (Random.cpp:121-126)
    V3 randomDirection(Rng& rng)
    {
        double theta = acos(2.0 * rng.uniform() - 1.0);
        double phi = 2.0 * M_PI * rng.uniform();
        return directionFromAngles(theta, phi);
    }
    // Random::direction(bfk, costheta) (Random.cpp:130-164)
    V3 randomDirectionAbout(Rng& rng, V3 k, double costheta)
    {
        double phi = 2.0 * M_PI * rng.uniform();
        double cosphi = cos(phi);
        double sinphi = sin(phi);
        double sintheta = sqrt(fabs((1.0 - costheta) * (1.0 + costheta)));
        double kx = k.x, ky = k.y, kz = k.z;
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
        return V3{kxnew, kynew, kznew};
    }
    // Random::exponCutoff (Random.cpp:105-116)
    double exponCutoff(Rng& rng, double xmax)
    {
        if (xmax == 0.0) return 0.0;
        if (xmax < 1e-10) return rng.uniform() * xmax;
        double x = -log(1.0 - rng.uniform() * (1.0 - exp(-xmax)));
        while (x > xmax) x = -log(1.0 - rng.uniform() * (1.0 - exp(-xmax)));
        return x;
    }

    // ------------------------------------------------------------------ path segment generators

    struct BoxD
    {
        double xmin, ymin, zmin, xmax, ymax, zmax;
        bool contains(double x, double y, double z) const
        {
            return x >= xmin && x <= xmax && y >= ymin && y <= ymax && z >= zmin && z <= zmax;
        }
    };

    // PathSegmentGenerator (PathSegmentGenerator.hpp:139-153, PathSegmentGenerator.cpp:11-112)
    struct Generator
    {
        enum class State { Unknown, Inside, Outside };
        const pmc_grid& g;
        State state{State::Unknown};
        double rx{0}, ry{0}, rz{0}, kx{0}, ky{0}, kz{0};
        int m{-1};
        double ds{0};

        explicit Generator(const pmc_grid& grid) : g(grid) {}
        virtual ~Generator() {}
        void start(V3 r, V3 k)
        {
            state = State::Unknown;
            rx = r.x, ry = r.y, rz = r.z;
            kx = k.x, ky = k.y, kz = k.z;
            m = -1;
            ds = 0.;
        }
        virtual bool next() = 0;

        void propagater(double s)
        {
            rx += kx * s;
            ry += ky * s;
            rz += kz * s;
        }
        void propagateToNextAfter()
        {
            rx = std::nextafter(rx, (kx < 0.) ? -DBL_MAX : DBL_MAX);
            ry = std::nextafter(ry, (ky < 0.) ? -DBL_MAX : DBL_MAX);
            rz = std::nextafter(rz, (kz < 0.) ? -DBL_MAX : DBL_MAX);
        }
        bool moveInside(const BoxD& box, double eps)
        {
            m = -1;
            ds = 0.;
            state = State::Outside;
            double cumds = 0.;
            if (rx <= box.xmin)
            {
                if (kx <= 0.0) return false;
                double d = (box.xmin - rx) / kx;
                rx = box.xmin + eps;
                ry += ky * d;
                rz += kz * d;
                cumds += d;
            }
            else if (rx >= box.xmax)
            {
                if (kx >= 0.0) return false;
                double d = (box.xmax - rx) / kx;
                rx = box.xmax - eps;
                ry += ky * d;
                rz += kz * d;
                cumds += d;
            }
            if (ry <= box.ymin)
            {
                if (ky <= 0.0) return false;
                double d = (box.ymin - ry) / ky;
                rx += kx * d;
                ry = box.ymin + eps;
                rz += kz * d;
                cumds += d;
            }
            else if (ry >= box.ymax)
            {
                if (ky >= 0.0) return false;
                double d = (box.ymax - ry) / ky;
                rx += kx * d;
                ry = box.ymax - eps;
                rz += kz * d;
                cumds += d;
            }
            if (rz <= box.zmin)
            {
                if (kz <= 0.0) return false;
                double d = (box.zmin - rz) / kz;
                rx += kx * d;
                ry += ky * d;
                rz = box.zmin + eps;
                cumds += d;
            }
            else if (rz >= box.zmax)
            {
                if (kz >= 0.0) return false;
                double d = (box.zmax - rz) / kz;
                rx += kx * d;
                ry += ky * d;
                rz = box.zmax - eps;
                cumds += d;
            }
            if (!box.contains(rx, ry, rz)) return false;
            m = -1;
            ds = cumds;
            state = State::Inside;
            return true;
        }
        BoxD extent() const { return BoxD{g.xmin, g.ymin, g.zmin, g.xmax, g.ymax, g.zmax}; }
    };

    // CartesianSpatialGrid::MySegmentGenerator (CartesianSpatialGrid.cpp:87-163)
    struct CartesianGenerator : Generator
    {
        int i{-1}, j{-1}, k{-1};
        using Generator::Generator;
        bool next() override
        {
            switch (state)
            {
                case State::Unknown:
                {
                    if (!moveInside(extent(), g.eps)) return false;
                    i = locateClip(g.xv, g.nx + 1, rx);
                    j = locateClip(g.yv, g.ny + 1, ry);
                    k = locateClip(g.zv, g.nz + 1, rz);
                    if (ds > 0.) return true;
                }
                // intentionally falls through
                case State::Inside:
                {
                    int mm = k + g.nz * j + g.nz * g.ny * i;
                    double xE = (kx < 0.0) ? g.xv[i] : g.xv[i + 1];
                    double yE = (ky < 0.0) ? g.yv[j] : g.yv[j + 1];
                    double zE = (kz < 0.0) ? g.zv[k] : g.zv[k + 1];
                    double dsx = (fabs(kx) > 1e-15) ? (xE - rx) / kx : DBL_MAX;
                    double dsy = (fabs(ky) > 1e-15) ? (yE - ry) / ky : DBL_MAX;
                    double dsz = (fabs(kz) > 1e-15) ? (zE - rz) / kz : DBL_MAX;
                    if (dsx <= dsy && dsx <= dsz)
                    {
                        m = mm, ds = dsx;
                        rx = xE;
                        ry += ky * dsx;
                        rz += kz * dsx;
                        i += (kx < 0.0) ? -1 : 1;
                        if (i >= g.nx || i < 0) state = State::Outside;
                    }
                    else if (dsy < dsx && dsy <= dsz)
                    {
                        m = mm, ds = dsy;
                        ry = yE;
                        rx += kx * dsy;
                        rz += kz * dsy;
                        j += (ky < 0.0) ? -1 : 1;
                        if (j >= g.ny || j < 0) state = State::Outside;
                    }
                    else
                    {
                        m = mm, ds = dsz;
                        rz = zE;
                        rx += kx * dsz;
                        ry += ky * dsz;
                        k += (kz < 0.0) ? -1 : 1;
                        if (k >= g.nz || k < 0) state = State::Outside;
                    }
                    return true;
                }
                case State::Outside:
                {
                }
            }
            return false;
        }
    };

    // VoronoiMeshSnapshot::MySegmentGenerator (VoronoiMeshSnapshot.cpp:1058-1188) with VoronoiMeshSnapshot::cellIndex
    // (:1006-1040: nearest site among the cells listed for the block of the position)
    struct VoronoiGenerator : Generator
    {
        int mr{-1};
        using Generator::Generator;
        int cellIndex(double x, double y, double z) const
        {
            if (!extent().contains(x, y, z)) return -1;
            const int nb = g.vblock_n;
            int i = std::max(0, std::min(nb - 1, static_cast<int>(nb * (x - g.xmin) / (g.xmax - g.xmin))));
            int j = std::max(0, std::min(nb - 1, static_cast<int>(nb * (y - g.ymin) / (g.ymax - g.ymin))));
            int k = std::max(0, std::min(nb - 1, static_cast<int>(nb * (z - g.zmin) / (g.zmax - g.zmin))));
            const size_t b = (static_cast<size_t>(i) * nb + j) * nb + k;
            int best = -1;
            double mdist = DBL_MAX;
            for (int q = g.vblock_start[b]; q != g.vblock_start[b + 1]; ++q)
            {
                const int id = g.vblock_list[q];
                double dx = x - g.site[3 * id], dy = y - g.site[3 * id + 1], dz = z - g.site[3 * id + 2];
                double idist = dx * dx + dy * dy + dz * dz;
                if (idist < mdist)
                {
                    best = id;
                    mdist = idist;
                }
            }
            return best;
        }
        bool next() override
        {
            switch (state)
            {
                case State::Unknown:
                {
                    if (!moveInside(extent(), g.eps)) return false;
                    mr = cellIndex(rx, ry, rz);
                    if (ds > 0.) return true;
                }
                // intentionally falls through
                case State::Inside:
                {
                    while (true)
                    {
                        const double prx = g.site[3 * mr], pry = g.site[3 * mr + 1], prz = g.site[3 * mr + 2];
                        double sq = DBL_MAX;
                        const int NO_INDEX = -99;
                        int mq = NO_INDEX;
                        for (int q = g.vnbr_start[mr]; q != g.vnbr_start[mr + 1]; ++q)
                        {
                            int mi = g.vnbr_list[q];
                            double si = 0;
                            if (mi >= 0)
                            {
                                const double pix = g.site[3 * mi], piy = g.site[3 * mi + 1], piz = g.site[3 * mi + 2];
                                const double nx = pix - prx, ny = piy - pry, nz = piz - prz;
                                double ndotk = nx * kx + ny * ky + nz * kz;
                                if (ndotk > 0)
                                {
                                    const double px = 0.5 * (pix + prx), py = 0.5 * (piy + pry), pz = 0.5 * (piz + prz);
                                    si = (nx * (px - rx) + ny * (py - ry) + nz * (pz - rz)) / ndotk;
                                }
                            }
                            else
                            {
                                switch (mi)
                                {
                                    case -1: si = (g.xmin - rx) / kx; break;
                                    case -2: si = (g.xmax - rx) / kx; break;
                                    case -3: si = (g.ymin - ry) / ky; break;
                                    case -4: si = (g.ymax - ry) / ky; break;
                                    case -5: si = (g.zmin - rz) / kz; break;
                                    case -6: si = (g.zmax - rz) / kz; break;
                                    default: break;
                                }
                            }
                            if (si > 0 && si < sq)
                            {
                                sq = si;
                                mq = mi;
                            }
                        }
                        if (mq == NO_INDEX)
                        {
                            propagater(g.eps);
                            mr = cellIndex(rx, ry, rz);
                            if (mr < 0)
                            {
                                state = State::Outside;
                                return false;
                            }
                        }
                        else
                        {
                            propagater(sq + g.eps);
                            m = mr;
                            ds = sq;
                            mr = mq;
                            if (mr < 0) state = State::Outside;
                            return true;
                        }
                    }
                }
                case State::Outside:
                {
                }
            }
            return false;
        }
    };

    // TreeSpatialGrid::MySegmentGenerator (TreeSpatialGrid.cpp:132-217) over the flattened node list
    struct TreeGenerator : Generator
    {
        int node{-1};
        using Generator::Generator;
        BoxD box(int id) const
        {
            const double* b = g.node_box + 6 * size_t(id);
            return BoxD{b[0], b[1], b[2], b[3], b[4], b[5]};
        }
        // TreeNode::leafChild (TreeNode.cpp:66-76) with OctTreeNode::child (OctTreeNode.cpp:36-41)
        int leafChild(int id, double x, double y, double z) const
        {
            if (!box(id).contains(x, y, z)) return -1;
            while (g.node_first_child[id] >= 0)
            {
                int first = g.node_first_child[id];
                const double* c0 = g.node_box + 6 * size_t(first);  // child 0: its rmax is the split point
                int l = (x < c0[3] ? 0 : 1) + (y < c0[4] ? 0 : 2) + (z < c0[5] ? 0 : 4);
                id = first + l;
            }
            return id;
        }
        // TreeNode::neighbor (TreeNode.cpp:103-112)
        int neighbor(int id, int wall, double x, double y, double z) const
        {
            int b = g.nbr_start[6 * size_t(id) + wall], e = g.nbr_start[6 * size_t(id) + wall + 1];
            for (int q = b; q < e; ++q)
                if (box(g.nbr_list[q]).contains(x, y, z)) return g.nbr_list[q];
            return -1;
        }
        bool next() override
        {
            switch (state)
            {
                case State::Unknown:
                {
                    if (!moveInside(extent(), g.eps)) return false;
                    node = leafChild(0, rx, ry, rz);
                    if (ds > 0.) return true;
                }
                // intentionally falls through
                case State::Inside:
                {
                    BoxD nb = box(node);
                    double xnext = (kx < 0.0) ? nb.xmin : nb.xmax;
                    double ynext = (ky < 0.0) ? nb.ymin : nb.ymax;
                    double znext = (kz < 0.0) ? nb.zmin : nb.zmax;
                    double dsx = (fabs(kx) > 1e-15) ? (xnext - rx) / kx : DBL_MAX;
                    double dsy = (fabs(ky) > 1e-15) ? (ynext - ry) / ky : DBL_MAX;
                    double dsz = (fabs(kz) > 1e-15) ? (znext - rz) / kz : DBL_MAX;
                    double d;
                    int wall;
                    if (dsx <= dsy && dsx <= dsz)
                    {
                        d = dsx;
                        wall = (kx < 0.0) ? PMC_WALL_BACK : PMC_WALL_FRONT;
                    }
                    else if (dsy <= dsx && dsy <= dsz)
                    {
                        d = dsy;
                        wall = (ky < 0.0) ? PMC_WALL_LEFT : PMC_WALL_RIGHT;
                    }
                    else
                    {
                        d = dsz;
                        wall = (kz < 0.0) ? PMC_WALL_BOTTOM : PMC_WALL_TOP;
                    }
                    propagater(d + g.eps);
                    m = g.node_cell[node];
                    ds = d;

                    int oldnode = node;
                    node = neighbor(node, wall, rx, ry, rz);
                    if (node < 0) node = leafChild(0, rx, ry, rz);
                    if (node == oldnode)
                    {
                        propagateToNextAfter();
                        node = leafChild(0, rx, ry, rz);
                    }
                    if (node < 0 || node == oldnode) state = State::Outside;
                    return true;
                }
                case State::Outside:
                {
                }
            }
            return false;
        }
    };

    // ------------------------------------------------------------------ photon packet and path (PhotonPacket.hpp:333-363)

    struct Segment
    {
        int m;
        double ds, s, tau;   // tau: cumulative extinction optical depth -- or, in the explicit-absorption cycle, scattering optical depth
        double tauAbs;       // cumulative absorption optical depth (explicit-absorption cycle), else zero (SpatialGridPath.hpp:98-108)
        double tauExt() const { return tau + tauAbs; }
    };

    struct Packet
    {
        V3 r{0, 0, 0}, k{0, 0, 1};
        double lambda{0}, W{0}, D{0};
        int nscatt{0};
        uint64_t historyIndex{0};
        bool hasObservedOpticalDepth{false};
        double observedOpticalDepth{0};
        std::vector<Segment> segments;
        double pathS{0};
        int interactionCell{-1};
        double interactionDistance{0};
        double interactionOpticalDepth{0};   // cumulative absorption optical depth at the interaction point (explicit absorption)
        const pmc_source* source{nullptr};   // the source that launched the packet (angular distribution of its emission peel-off packets)
        double luminosity() const { return W / lambda; }
    };

    // ---- axisymmetric angular distributions of a point source (AxAngularDistribution.cpp:27-41)
    struct NetzerTable
    {
        // NetzerAngularDistribution::setupSelfBefore (NetzerAngularDistribution.cpp:12-30; NR::buildLinearGrid, NR.hpp:203-209)
        double costheta[PMC_NETZER_POINTS + 1], X[PMC_NETZER_POINTS + 1];
        NetzerTable()
        {
            const int n = PMC_NETZER_POINTS;
            double dx = (+1. - -1.) / n;
            for (int i = 0; i <= n; i++) costheta[i] = -1. + i * dx;
            X[0] = 0;
            for (int i = 1; i < n; i++)
            {
                double ct = costheta[i];
                double sign = ct > 0 ? 1. : -1;
                X[i] = (1. / 2.) + (2. / 7.) * ct * ct * ct + sign * (3. / 14.) * ct * ct;
            }
            X[n] = 1.;
        }
    };
    // probabilityForInclinationCosine: LaserAngularDistribution.cpp:11-16, ConicalAngularDistribution.cpp:19-25,
    // NetzerAngularDistribution.cpp:34-38
    double angularProbability(const pmc_source& s, V3 k)
    {
        double costheta = s.angular_axis[0] * k.x + s.angular_axis[1] * k.y + s.angular_axis[2] * k.z;
        if (s.angular_kind == PMC_ANGULAR_LASER) return costheta > 0.99999 ? std::numeric_limits<double>::infinity() : 0.;
        if (s.angular_kind == PMC_ANGULAR_CONICAL) return std::abs(costheta) > s.angular_cos_delta ? 1.0 / (1.0 - s.angular_cos_delta) : 0.;
        double sign = costheta > 0 ? 1. : -1;
        return (6. / 7.) * costheta * (2. * costheta + sign);
    }

    struct Contribution
    {
        int ell, l;
        double w;
        bool operator<(const Contribution& c) const { return std::tie(ell, l) < std::tie(c.ell, c.l); }
    };

    // ------------------------------------------------------------------ the life cycle

    struct LifeCycle
    {
        const pmc_scene& sc;
        Rng& rng;
        double* frames;
        pmc_counter_values counters{};
        double* radiationField{nullptr};  // rf[m * num_lambda + ell], or null: not stored
        std::unique_ptr<Generator> generator;
        std::vector<pmc_frame_layout> layouts;
        // per instrument contribution lists for the statistics (FluxRecorder.hpp:308-343)
        struct ContributionList
        {
            uint64_t historyIndex{0};
            std::vector<Contribution> contributions;
        };
        std::vector<ContributionList> lists;

        LifeCycle(const pmc_scene& scene, Rng& r, double* f) : sc(scene), rng(r), frames(f)
        {
            if (sc.grid.kind == PMC_GRID_CARTESIAN)
                generator.reset(new CartesianGenerator(sc.grid));
            else if (sc.grid.kind == PMC_GRID_VORONOI)
                generator.reset(new VoronoiGenerator(sc.grid));
            else
                generator.reset(new TreeGenerator(sc.grid));
            layouts.resize(sc.num_instruments);
            for (int i = 0; i < sc.num_instruments; ++i) pmc_layout_compute(&sc, i, &layouts[i]);
            lists.resize(sc.num_instruments);
        }

        // DustMix::indexForLambda / sectionExt / ... (DustMix.cpp:276-279,319-367)
        // several medium components with constant cross sections (Configuration::hasMultipleConstantSectionMedia)
        int numMedia() const { return sc.num_media > 1 ? sc.num_media : 1; }
        const pmc_medium& med(int h) const { return sc.num_media > 1 ? sc.media[h] : sc.medium; }
        int indexForLambda(int h, double lambda) const { return locateClip(med(h).lambda_border, med(h).num_lambda, lambda); }
        // opacity of component h in cell m (MaterialMix::opacitySca / opacityExt: n sigma)
        double opacityOf(int h, const double* sigma, double lambda, int m) const
        {
            double n = med(h).number_density[m];
            return n > 0. ? n * sigma[indexForLambda(h, lambda)] : 0.;
        }
        int indexForLambda(double lambda) const { return locateClip(sc.medium.lambda_border, sc.medium.num_lambda, lambda); }
        double sectionExt(double lambda) const { return sc.medium.sigma_ext[indexForLambda(lambda)]; }
        double opacity(const double* sigma, double lambda, int m) const
        {
            double n = sc.medium.number_density[m];
            return n > 0. ? n * sigma[indexForLambda(lambda)] : 0.;
        }

        // ---- SourceSystem::launch -> NormalizedSource::launch -> Point/GeometricSource (SourceSystem.cpp:101-112,
        //      NormalizedSource.cpp:73-110, PointSource.cpp:32-43, GeometricSource.cpp:66-82, PhotonPacket.cpp:18-40)
        void launch(Packet& pp, uint64_t historyIndex)
        {
            // SourceSystem::launch (SourceSystem.cpp:100-107): upper_bound of the history index in the _Iv boundaries
            int si = 0;
            if (sc.num_sources > 1)
                while (si + 1 < sc.num_sources && historyIndex >= sc.source_first[si + 1]) ++si;
            const pmc_source& s = sc.num_sources > 1 ? sc.sources[si] : sc.source;
            double L = s.packet_luminosity;
            double lambda, w;
            if (s.lambda_mode == PMC_LAMBDA_OLIGO)
            {
                // xi = 1: `uniform() > xi` never holds, then OligoWavelengthDistribution::generateWavelength
                (void)(rng.uniform() > 1.);
                size_t index = static_cast<size_t>(rng.uniform() * s.num_oligo);
                lambda = s.oligo_lambda[index];
                w = s.oligo_weight[index];
            }
            else
            {
                double xi = s.lambda_bias;
                auto sedSample = [&]() {
                    // Random::cdfLogLog (Random.cpp:209-216)
                    double X = rng.uniform();
                    int i = locateClip(s.sed_P, s.num_sed, X);
                    double alpha = log(s.sed_p[i + 1] / s.sed_p[i]) / log(s.sed_lambda[i + 1] / s.sed_lambda[i]);
                    return s.sed_lambda[i] * gexp(-alpha, (X - s.sed_P[i]) / (s.sed_p[i] * s.sed_lambda[i]));
                };
                if (!xi)
                {
                    lambda = sedSample();
                    w = 1.;
                }
                else
                {
                    double logMin = log(s.bias_min), logWidth = log(s.bias_max) - log(s.bias_min);
                    if (rng.uniform() > xi)
                        lambda = sedSample();
                    else if (s.bias_kind == PMC_BIAS_LIN)
                        lambda = s.bias_min + (s.bias_max - s.bias_min) * rng.uniform();
                    else
                        lambda = exp(logMin + logWidth * rng.uniform());
                    // SED::specificLuminosity: analytic for a black body (BlackBodySED.cpp:36-39, PlanckFunction.cpp:24-27),
                    // otherwise log-log interpolation of the normalised table
                    double sl = 0.;
                    if (s.sed_kind == PMC_SED_BLACKBODY)
                        sl = s.sed_f2 / pow(lambda, 5) / (exp(s.sed_f1 / lambda) - 1.0) / s.sed_ltot;
                    else if (lambda >= s.sed_lambda[0] && lambda <= s.sed_lambda[s.num_sed - 1])
                    {
                        int i = locate(s.sed_lambda, s.num_sed, lambda);
                        if (i < 0) i = 0;
                        sl = interpolateLogLog(lambda, s.sed_lambda[i], s.sed_lambda[i + 1], s.sed_p[i], s.sed_p[i + 1]);
                    }
                    if (!sl)
                        w = 0.;
                    else
                    {
                        double b = 0.;
                        if (lambda >= s.bias_min && lambda <= s.bias_max)
                            b = s.bias_kind == PMC_BIAS_LIN ? 1. / (s.bias_max - s.bias_min) : 1. / (logWidth * lambda);
                        w = sl / ((1 - xi) * sl + xi * b);
                    }
                }
            }
            V3 r;
            if (s.kind == PMC_SOURCE_POINT)
                r = V3{s.position[0], s.position[1], s.position[2]};
            else if (s.kind == PMC_SOURCE_SERSIC)
            {
                // SersicGeometry::randomRadius + SpheGeometry::generatePosition (SersicGeometry.cpp:41-45,
                // SpheGeometry.cpp:25-32, SersicFunction.cpp:97-100)
                double X = rng.uniform();
                int n = s.sersic_n;
                int i = locate(s.sersic_M, n, X);
                double sv;
                if (i < 0)
                    sv = s.sersic_s[0];
                else if (i >= n - 1)
                    sv = s.sersic_s[n - 1];
                else
                    sv = interpolateLogLog(X, s.sersic_M[i], s.sersic_M[i + 1], s.sersic_s[i], s.sersic_s[i + 1]);
                double radius = s.reff * sv;
                V3 d = randomDirection(rng);
                r = V3{d.x * radius, d.y * radius, d.z * radius};
            }
            else if (s.kind == PMC_SOURCE_EXP_DISK)
            {
                // ExpDiskGeometry::randomCylRadius / randomZ with SepAxGeometry::generatePosition
                // (ExpDiskGeometry.cpp:46-68, SepAxGeometry.cpp:11-19)
                const double hR = s.box[0], hz = s.box[1], Rmin = s.box[2], Rmax = s.box[3], zmax = s.box[4];
                double R, X;
                do
                {
                    X = rng.uniform();
                    R = hR * (-1.0 - lambertW1((X - 1.0) / M_E));
                } while ((Rmax > 0.0 && R >= Rmax) || R <= Rmin);
                double phi = 2.0 * M_PI * rng.uniform();
                double z;
                do
                {
                    X = rng.uniform();
                    z = (X <= 0.5) ? hz * log(2.0 * X) : -hz * log(2.0 * (1.0 - X));
                } while (zmax > 0.0 && fabs(z) >= zmax);
                r = V3{R * cos(phi), R * sin(phi), z};
            }
            else if (s.kind == PMC_SOURCE_PLUMMER)
            {
                // PlummerGeometry::randomRadius + SpheGeometry::generatePosition (PlummerGeometry.cpp:29-33)
                double t = pow(rng.uniform(), 1.0 / 3.0);
                double radius = s.box[0] * t / sqrt((1.0 - t) * (1.0 + t));
                V3 d = randomDirection(rng);
                r = V3{radius * d.x, radius * d.y, radius * d.z};
            }
            else
            {
                double x = rng.uniform();
                double y = rng.uniform();
                double z = rng.uniform();
                r = V3{s.box[0] + x * (s.box[3] - s.box[0]), s.box[1] + y * (s.box[4] - s.box[1]),
                       s.box[2] + z * (s.box[5] - s.box[2])};
            }
            // SpheroidalGeometryDecorator::generatePosition (SpheroidalGeometryDecorator.cpp:19-25)
            if ((s.kind == PMC_SOURCE_SERSIC || s.kind == PMC_SOURCE_PLUMMER) && s.box[5] != 0.) r.z = s.box[5] * r.z;
            // OffsetGeometryDecorator::generatePosition (OffsetGeometryDecorator.cpp:33-39)
            if (s.kind != PMC_SOURCE_POINT && (s.position[0] != 0. || s.position[1] != 0. || s.position[2] != 0.))
                r = V3{r.x + s.position[0], r.y + s.position[1], r.z + s.position[2]};
            V3 k;
            if (s.kind == PMC_SOURCE_POINT && s.angular_kind != PMC_ANGULAR_ISOTROPIC)
            {
                // AxAngularDistribution::generateDirection (AxAngularDistribution.cpp:36-39) with generateInclinationCosine of
                // LaserAngularDistribution.cpp:20-23, ConicalAngularDistribution.cpp:29-36, NetzerAngularDistribution.cpp:42-45
                // (Random::cdfLinLin, Random.cpp:201-206)
                double costheta = 1.;
                if (s.angular_kind == PMC_ANGULAR_CONICAL)
                {
                    double X = rng.uniform();
                    if (X < 0.5)
                        costheta = 1.0 - 2.0 * X * (1.0 - s.angular_cos_delta);
                    else
                        costheta = 1.0 - 2.0 * s.angular_cos_delta - 2.0 * X * (1.0 - s.angular_cos_delta);
                }
                else if (s.angular_kind == PMC_ANGULAR_NETZER)
                {
                    static const NetzerTable T;
                    double X = rng.uniform();
                    int i = locateClip(T.X, PMC_NETZER_POINTS + 1, X);
                    costheta = interpolateLinLin(X, T.X[i], T.X[i + 1], T.costheta[i], T.costheta[i + 1]);
                }
                k = randomDirectionAbout(rng, V3{s.angular_axis[0], s.angular_axis[1], s.angular_axis[2]}, costheta);
            }
            else
                k = randomDirection(rng);
            pp.source = &s;
            double Lw = L * w;
            pp.lambda = lambda;
            pp.W = Lw * lambda;
            pp.D = 0;
            pp.historyIndex = historyIndex;
            pp.nscatt = 0;
            pp.r = r;
            pp.k = k;
            pp.hasObservedOpticalDepth = false;
            counters.histories++;
        }

        // ---- MediumSystem::setExtinctionOpticalDepths, single constant-section medium (MediumSystem.cpp:849-871)
        void setExtinctionOpticalDepths(Packet& pp)
        {
            generator->start(pp.r, pp.k);
            pp.segments.clear();
            pp.pathS = 0.;
            counters.paths++;
            while (generator->next())
            {
                // SpatialGridPath::addSegment (SpatialGridPath.cpp:41-48)
                if (generator->ds > 0.)
                {
                    pp.pathS += generator->ds;
                    pp.segments.push_back(Segment{generator->m, generator->ds, pp.pathS, 0., 0.});
                }
            }
            if (numMedia() > 1)
            {
                // several components: MediumSystem.cpp:874-887 (extinction), :934-955 (scattering and absorption apart)
                const int H = numMedia();
                double sext[PMC_MAX_MEDIA], ssca[PMC_MAX_MEDIA], sabs[PMC_MAX_MEDIA];
                for (int h = 0; h != H; ++h)
                {
                    const int ell = indexForLambda(h, pp.lambda);
                    sext[h] = med(h).sigma_ext[ell], ssca[h] = med(h).sigma_sca[ell], sabs[h] = med(h).sigma_abs[ell];
                }
                double tau = 0., tauSca = 0., tauAbs = 0.;
                for (auto& seg : pp.segments)
                {
                    if (seg.m >= 0)
                    {
                        counters.cell_visits++;
                        for (int h = 0; h != H; ++h)
                        {
                            if (sc.options.explicit_absorption)
                            {
                                double ns = med(h).number_density[seg.m] * seg.ds;
                                tauSca += ssca[h] * ns;
                                tauAbs += sabs[h] * ns;
                            }
                            else
                                tau += sext[h] * med(h).number_density[seg.m] * seg.ds;
                        }
                    }
                    seg.tau = sc.options.explicit_absorption ? tauSca : tau;
                    seg.tauAbs = tauAbs;
                }
                return;
            }
            if (sc.options.explicit_absorption)
            {
                // MediumSystem::setScatteringAndAbsorptionOpticalDepths, single constant-section medium (MediumSystem.cpp:905-932)
                double tauSca = 0., tauAbs = 0.;
                const int ell = indexForLambda(pp.lambda);
                const double sectionSca = sc.medium.sigma_sca[ell], sectionAbs = sc.medium.sigma_abs[ell];
                for (auto& seg : pp.segments)
                {
                    if (seg.m >= 0)
                    {
                        double ns = sc.medium.number_density[seg.m] * seg.ds;
                        tauSca += sectionSca * ns;
                        tauAbs += sectionAbs * ns;
                        counters.cell_visits++;
                    }
                    seg.tau = tauSca;
                    seg.tauAbs = tauAbs;
                }
                return;
            }
            double tau = 0.;
            double section = sectionExt(pp.lambda);
            for (auto& seg : pp.segments)
            {
                if (seg.m >= 0)
                {
                    tau += section * sc.medium.number_density[seg.m] * seg.ds;
                    counters.cell_visits++;
                }
                seg.tau = tau;
            }
        }

        // ---- MonteCarloSimulation::storeRadiationField, constant perceived wavelength (MonteCarloSimulation.cpp:638-662)
        //      with SpecialFunctions::lnmean (SpecialFunctions.cpp:860-880) and MediumSystem::storeRadiationField
        //      (MediumSystem.cpp:1294-1300): rf1(m, ell) += L * lnmean(extEnd, extBeg) * ds
        static double lnmean(double x1, double x2, double lnx1, double lnx2)
        {
            if (x1 > x2)
            {
                std::swap(x1, x2);
                std::swap(lnx1, lnx2);
            }
            if (x1 <= 0) return 0.;
            double x = x2 / x1 - 1.;
            if (x < 1e-3)
                return x1
                       / (1. - 1. / 2. * x + 1. / 3. * x * x - 1. / 4. * x * x * x + 1. / 5. * x * x * x * x
                          - 1. / 6. * x * x * x * x * x);
            return (x2 - x1) / (lnx2 - lnx1);
        }
        void storeRadiationField(const Packet& pp)
        {
            const pmc_radiation_field& R = sc.radiation_field;
            // DisjointWavelengthGrid::bin (DisjointWavelengthGrid.cpp:334-345)
            int ell = R.ellv[std::upper_bound(R.border, R.border + R.num_border, pp.lambda) - R.border];
            if (ell < 0) return;
            double luminosity = pp.luminosity();
            double lnExtBeg = 0.;
            double extBeg = 1.;
            for (const auto& segment : pp.segments)
            {
                double lnExtEnd = -segment.tauExt();
                double extEnd = exp(lnExtEnd);
                int m = segment.m;
                if (m >= 0)
                {
                    double extMean = lnmean(extEnd, extBeg, lnExtEnd, lnExtBeg);
                    double Lds = luminosity * extMean * segment.ds;
                    radiationField[static_cast<size_t>(m) * R.num_lambda + ell] += Lds;
                }
                lnExtBeg = lnExtEnd;
                extBeg = extEnd;
            }
        }

        // ---- MediumSystem::getExtinctionOpticalDepth (MediumSystem.cpp:1192-1223)
        double getExtinctionOpticalDepth(const Packet& pp, double distance)
        {
            double L = pp.luminosity();
            if (L <= 0) return std::numeric_limits<double>::infinity();
            double taumax = std::log(L) + 745;
            generator->start(pp.r, pp.k);
            counters.paths++;
            double tau = 0., s = 0.;
            if (numMedia() > 1)
            {
                // MediumSystem.cpp:1225-1242
                const int H = numMedia();
                double sectionv[PMC_MAX_MEDIA];
                for (int h = 0; h != H; ++h) sectionv[h] = med(h).sigma_ext[indexForLambda(h, pp.lambda)];
                while (generator->next())
                {
                    double ds = generator->ds;
                    int m = generator->m;
                    if (m >= 0)
                    {
                        counters.cell_visits++;
                        for (int h = 0; h != H; ++h) tau += sectionv[h] * med(h).number_density[m] * ds;
                        if (tau >= taumax) return std::numeric_limits<double>::infinity();
                    }
                    s += ds;
                    if (s > distance) break;
                }
                return tau;
            }
            double section = sectionExt(pp.lambda);
            while (generator->next())
            {
                if (generator->m >= 0)
                {
                    counters.cell_visits++;
                    tau += section * sc.medium.number_density[generator->m] * generator->ds;
                    if (tau >= taumax) return std::numeric_limits<double>::infinity();
                }
                s += generator->ds;
                if (s > distance) break;
            }
            return tau;
        }

        // ---- SpatialGridPath::findInteractionPoint (SpatialGridPath.cpp:164-206)
        void findInteractionPoint(Packet& pp, double tauinteract)
        {
            auto& segs = pp.segments;
            if (segs.empty())
            {
                pp.interactionCell = -1;
                pp.interactionDistance = 0.;
                pp.interactionOpticalDepth = 0.;
                return;
            }
            auto seg = std::upper_bound(segs.cbegin(), segs.cend(), tauinteract,
                                        [](double t, const Segment& sg) { return t < sg.tau; });
            if (seg == segs.cbegin())
            {
                pp.interactionCell = seg->m;
                pp.interactionDistance = interpolateLinLin(tauinteract, 0., seg->tau, 0., seg->s);
                pp.interactionOpticalDepth = interpolateLinLin(tauinteract, 0., seg->tau, 0., seg->tauAbs);
            }
            else if (seg < segs.cend())
            {
                pp.interactionCell = seg->m;
                pp.interactionDistance = interpolateLinLin(tauinteract, (seg - 1)->tau, seg->tau, (seg - 1)->s, seg->s);
                pp.interactionOpticalDepth = interpolateLinLin(tauinteract, (seg - 1)->tau, seg->tau, (seg - 1)->tauAbs, seg->tauAbs);
            }
            else
            {
                pp.interactionCell = (seg - 1)->m;
                pp.interactionDistance = (seg - 1)->s;
                pp.interactionOpticalDepth = (seg - 1)->tauAbs;
            }
        }

        // ---- MediumSystem::albedoForScattering (MediumSystem.cpp:678-693): scattering over extinction opacity of all components in the
        //      interaction cell
        double albedoForScattering(const Packet& pp) const
        {
            int m = pp.interactionCell;
            double ksca = 0., kext = 0.;
            for (int h = 0; h != numMedia(); ++h)
            {
                ksca += opacityOf(h, med(h).sigma_sca, pp.lambda, m);
                kext += opacityOf(h, med(h).sigma_ext, pp.lambda, m);
            }
            return kext > 0. ? ksca / kext : 0.;
        }

        // ---- MonteCarloSimulation::simulateForcedPropagation (MonteCarloSimulation.cpp:696-742)
        void simulateForcedPropagation(Packet& pp)
        {
            double taupath = pp.segments.empty() ? 0. : pp.segments.back().tau;
            if (taupath <= 0.)
            {
                pp.W *= 0.;
                return;
            }
            double xi = sc.options.path_length_bias;
            double tau = 0.;
            if (xi == 0.)
                tau = exponCutoff(rng, taupath);
            else
            {
                tau = rng.uniform() < xi ? rng.uniform() * taupath : exponCutoff(rng, taupath);
                double p = -exp(-tau) / expm1(-taupath);
                double q = (1.0 - xi) * p + xi / taupath;
                double weight = p / q;
                pp.W *= weight;
            }
            findInteractionPoint(pp, tau);
            if (sc.options.explicit_absorption)
            {
                // MonteCarloSimulation.cpp:729-733: the escape fraction and the absorption along the way to the interaction point
                double tauAbs = pp.interactionOpticalDepth;
                pp.W *= (-expm1(-taupath) * exp(-tauAbs));
            }
            else
            {
                // MediumSystem::albedoForScattering (MediumSystem.cpp:678-693)
                pp.W *= (-expm1(-taupath) * albedoForScattering(pp));
            }
            // PhotonPacket::propagate (PhotonPacket.cpp:107-111)
            double s = pp.interactionDistance;
            pp.r.x += s * pp.k.x;
            pp.r.y += s * pp.k.y;
            pp.r.z += s * pp.k.z;
            pp.D += s;
        }

        // ---- non-forced propagation (MonteCarloSimulation.cpp:746-780, MediumSystem.cpp:978-1012)
        bool simulateNonForcedPropagation(Packet& pp)
        {
            double tauinteract = -log(rng.uniform());  // Random::expon
            generator->start(pp.r, pp.k);
            counters.paths++;
            const bool explicitAbsorption = sc.options.explicit_absorption != 0;
            if (numMedia() > 1)
            {
                // several components: MediumSystem.cpp:1013-1037 (extinction), :1112-1153 (scattering and absorption apart)
                const int H = numMedia();
                double sext[PMC_MAX_MEDIA], ssca[PMC_MAX_MEDIA], sabs[PMC_MAX_MEDIA];
                for (int h = 0; h != H; ++h)
                {
                    const int ell = indexForLambda(h, pp.lambda);
                    sext[h] = med(h).sigma_ext[ell], ssca[h] = med(h).sigma_sca[ell], sabs[h] = med(h).sigma_abs[ell];
                }
                double tau = 0., tauAbs = 0., s = 0.;
                while (generator->next())
                {
                    double tau0 = tau, tauAbs0 = tauAbs, s0 = s;
                    double ds = generator->ds;
                    int m = generator->m;
                    if (m >= 0)
                    {
                        counters.cell_visits++;
                        for (int h = 0; h != H; ++h)
                        {
                            if (explicitAbsorption)
                            {
                                double ns = med(h).number_density[m] * ds;
                                tau += ssca[h] * ns;
                                tauAbs += sabs[h] * ns;
                            }
                            else
                                tau += sext[h] * med(h).number_density[m] * ds;
                        }
                    }
                    s += ds;
                    if (tauinteract < tau)
                    {
                        pp.interactionCell = m;
                        pp.interactionDistance = interpolateLinLin(tauinteract, tau0, tau, s0, s);
                        if (explicitAbsorption)
                            pp.W *= exp(-interpolateLinLin(tauinteract, tau0, tau, tauAbs0, tauAbs));
                        else
                            pp.W *= albedoForScattering(pp);
                        double sd = pp.interactionDistance;
                        pp.r.x += sd * pp.k.x;
                        pp.r.y += sd * pp.k.y;
                        pp.r.z += sd * pp.k.z;
                        pp.D += sd;
                        return true;
                    }
                }
                return false;
            }
            const int ellmix = indexForLambda(pp.lambda);
            // (explicit absorption: the interaction is drawn on the SCATTERING optical depth, MediumSystem.cpp:1075-1110)
            double section = explicitAbsorption ? sc.medium.sigma_sca[ellmix] : sectionExt(pp.lambda);
            double tau = 0., s = 0.;
            bool found = false;
            while (generator->next())
            {
                double tau0 = tau;
                double s0 = s;
                double ds = generator->ds;
                int m = generator->m;
                if (m >= 0)
                {
                    counters.cell_visits++;
                    tau += section * sc.medium.number_density[m] * ds;
                }
                s += ds;
                if (tauinteract < tau)
                {
                    pp.interactionCell = m;
                    pp.interactionDistance = interpolateLinLin(tauinteract, tau0, tau, s0, s);
                    found = true;
                    break;
                }
            }
            if (!found) return false;
            if (explicitAbsorption)
            {
                // MediumSystem.cpp:1105-1106 and MonteCarloSimulation.cpp:751-766
                double tauAbs = tauinteract * sc.medium.sigma_abs[ellmix] / sc.medium.sigma_sca[ellmix];
                pp.W *= exp(-tauAbs);
            }
            else
            {
                pp.W *= albedoForScattering(pp);
            }
            double sd = pp.interactionDistance;
            pp.r.x += sd * pp.k.x;
            pp.r.y += sd * pp.k.y;
            pp.r.z += sd * pp.k.z;
            pp.D += sd;
            return true;
        }

        // ---- FrameInstrument::pixelOnDetector (FrameInstrument.cpp:45-65)
        static int pixelOnDetector(const pmc_instrument& ins, V3 r)
        {
            double xpp = -ins.sinphi * r.x + ins.cosphi * r.y;
            double ypp = -ins.cosphi * ins.costheta * r.x - ins.sinphi * ins.costheta * r.y + ins.sintheta * r.z;
            double xp = ins.cosomega * xpp - ins.sinomega * ypp;
            double yp = ins.sinomega * xpp + ins.cosomega * ypp;
            int i = static_cast<int>(floor((xp - ins.xpmin) / ins.xpsiz));
            int j = static_cast<int>(floor((yp - ins.ypmin) / ins.ypsiz));
            if (i < 0 || i >= ins.nxp || j < 0 || j >= ins.nyp) return -1;
            return i + ins.nxp * j;
        }

        void add(int64_t index, double value)
        {
            frames[index] += value;
            counters.detector_updates++;
        }

        // ---- FluxRecorder::recordContributions (FluxRecorder.cpp:962-1014)
        void recordContributions(int instrument)
        {
            ContributionList& list = lists[instrument];
            const pmc_instrument& ins = sc.instruments[instrument];
            const pmc_frame_layout& L = layouts[instrument];
            std::sort(list.contributions.begin(), list.contributions.end());
            const auto& c = list.contributions;
            size_t n = c.size();
            if (ins.include_flux_density)
            {
                double w = 0;
                for (size_t i = 0; i != n; ++i)
                {
                    w += c[i].w;
                    if (i + 1 == n || c[i].ell != c[i + 1].ell)
                    {
                        int ell = c[i].ell;
                        double wn = 1.;
                        for (int k = 0; k <= 4; ++k)
                        {
                            add(L.wsed_offset + k * L.num_lambda + ell, wn);
                            wn *= w;
                        }
                        w = 0;
                    }
                }
            }
            if (ins.include_surface_brightness)
            {
                double w = 0;
                for (size_t i = 0; i != n; ++i)
                {
                    w += c[i].w;
                    if (i + 1 == n || c[i].ell != c[i + 1].ell || c[i].l != c[i + 1].l)
                    {
                        if (c[i].l >= 0)
                        {
                            int64_t lell = c[i].l + int64_t(c[i].ell) * L.npix;
                            double wn = 1.;
                            for (int k = 0; k <= 4; ++k)
                            {
                                add(L.wifu_offset + k * L.npix * L.num_lambda + lell, wn);
                                wn *= w;
                            }
                        }
                        w = 0;
                    }
                }
            }
        }

        // ---- FluxRecorder::detect (FluxRecorder.cpp:304-468) for a distant instrument
        void detect(int instrument, Packet& ppp, int l)
        {
            const pmc_instrument& ins = sc.instruments[instrument];
            const pmc_frame_layout& L = layouts[instrument];
            // ApertureInstrument::isInsideAperture (ApertureInstrument.cpp:22-43), tested by SEDInstrument::detect
            if (ins.aperture_radius2)
            {
                double xpp = -ins.sinphi * ppp.r.x + ins.cosphi * ppp.r.y;
                double ypp = -ins.cosphi * ins.costheta * ppp.r.x - ins.sinphi * ins.costheta * ppp.r.y + ins.sintheta * ppp.r.z;
                double radius2 = xpp * xpp + ypp * ypp;
                if (radius2 > ins.aperture_radius2) return;
            }
            if (!ins.include_flux_density && l < 0) return;
            double wavelength = ppp.lambda * (1. + ins.redshift);
            // DisjointWavelengthGrid::bins (DisjointWavelengthGrid.cpp:320-345)
            size_t index = std::upper_bound(ins.border, ins.border + ins.num_border, wavelength) - ins.border;
            int ell = ins.ellv[index];
            if (ell < 0) return;

            double Lum = ppp.luminosity() * 1.;  // transmission of a disjoint grid is 1
            double Lext = Lum;
            double tau;
            if (ppp.hasObservedOpticalDepth)
                tau = ppp.observedOpticalDepth;
            else
            {
                tau = getExtinctionOpticalDepth(ppp, std::numeric_limits<double>::infinity());
                ppp.observedOpticalDepth = tau;
                ppp.hasObservedOpticalDepth = true;
            }
            Lext *= exp(-tau);

            auto record = [&](int64_t base, int64_t len, int64_t idx) {
                int numScatt = ppp.nscatt;
                if (!ins.record_components)
                    add(base + idx, Lext);
                else if (numScatt == 0)
                {
                    add(base + 0 * len + idx, Lum);   // Transparent
                    add(base + 1 * len + idx, Lext);  // PrimaryDirect
                }
                else
                {
                    add(base + 2 * len + idx, Lext);  // PrimaryScattered
                    if (numScatt <= ins.num_scattering_levels) add(base + (3 + numScatt - 1) * len + idx, Lext);
                }
            };
            if (ins.include_flux_density) record(L.sed_offset, L.num_lambda, ell);
            if (ins.include_surface_brightness && l >= 0) record(L.ifu_offset, L.npix * L.num_lambda, l + int64_t(ell) * L.npix);

            if (ins.record_statistics)
            {
                ContributionList& list = lists[instrument];
                if (list.historyIndex != ppp.historyIndex)
                {
                    recordContributions(instrument);
                    list.historyIndex = ppp.historyIndex;
                    list.contributions.clear();
                }
                list.contributions.push_back(Contribution{ell, l, Lext});
            }
        }

        // ---- MonteCarloSimulation::peelOffEmission (MonteCarloSimulation.cpp:617-634, PhotonPacket.cpp:66-85)
        void peelOffEmission(const Packet& pp, Packet& ppp)
        {
            for (int i = 0; i < sc.num_instruments; ++i)
            {
                const pmc_instrument& ins = sc.instruments[i];
                if (!ins.same_observer_as_preceding)
                {
                    V3 k{ins.kobs[0], ins.kobs[1], ins.kobs[2]};
                    ppp.lambda = pp.lambda;
                    ppp.W = pp.W;
                    // PhotonPacket::launchEmissionPeelOff (PhotonPacket.cpp:66-85): the bias of an anisotropic emitter
                    if (pp.source && pp.source->kind == PMC_SOURCE_POINT && pp.source->angular_kind != PMC_ANGULAR_ISOTROPIC)
                        ppp.W *= angularProbability(*pp.source, k);
                    ppp.D = -(k.x * pp.r.x + k.y * pp.r.y + k.z * pp.r.z);
                    ppp.historyIndex = pp.historyIndex;
                    ppp.nscatt = 0;
                    ppp.r = pp.r;
                    ppp.k = k;
                    ppp.hasObservedOpticalDepth = false;
                }
                detect(i, ppp, pixelOnDetector(ins, ppp.r));
            }
        }

        // ---- HG phase function (DustMix.cpp:395-445)
        static double valueHG(double g, double costheta)
        {
            double t = 1. + g * g - 2. * g * costheta;
            return (1. - g) * (1. + g) / sqrt(t * t * t);
        }
        static double integralHG(double g, double cosalpha, double cosbeta)
        {
            double ta = sqrt(1. + g * g - 2. * g * cosalpha);
            double tb = sqrt(1. + g * g - 2. * g * cosbeta);
            double f1 = (1. - g) * (1. + g) / g;
            double f2 = (tb - ta) / (tb * ta);
            return f1 * f2;
        }
        static double meanHG(double g, double costheta)
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

        // ---- MonteCarloSimulation::peelOffScattering, consolidated branch (MonteCarloSimulation.cpp:784-842;
        //      MediumSystem.cpp:697-767; DustMix.cpp:430-445; PhotonPacket.cpp:89-103)
        void peelOffScattering(const Packet& pp, Packet& ppp)
        {
            double lambda = pp.lambda;
            // MediumSystem::weightsForScattering (MediumSystem.cpp:697-730): the components' shares of the scattering opacity in the
            // interaction cell; no peel-off at all if none of them scatters
            const int H = numMedia();
            double wv[PMC_MAX_MEDIA] = {1., 0., 0., 0.};
            if (H > 1)
            {
                double sum = 0.;
                for (int h = 0; h != H; ++h)
                {
                    wv[h] = opacityOf(h, med(h).sigma_sca, lambda, pp.interactionCell);
                    sum += wv[h];
                }
                if (!(sum > 0.)) return;
                for (int h = 0; h != H; ++h) wv[h] /= sum;
            }
            for (int i = 0; i < sc.num_instruments; ++i)
            {
                const pmc_instrument& ins = sc.instruments[i];
                if (!ins.same_observer_as_preceding)
                {
                    V3 k{ins.kobs[0], ins.kobs[1], ins.kobs[2]};
                    double costheta = pp.k.x * k.x + pp.k.y * k.y + pp.k.z * k.z;
                    double I = 0.;
                    for (int h = 0; h != H; ++h)
                    {
                        // (MediumSystem.cpp:745-757: components that do not scatter the packet are skipped)
                        if (wv[h] > 0.)
                        {
                            double g = med(h).asymmpar[indexForLambda(h, lambda)];
                            double value = std::abs(g) > 0.95 ? meanHG(g, costheta) : valueHG(g, costheta);
                            I += value * wv[h];
                        }
                    }
                    ppp.lambda = lambda;
                    ppp.W = pp.W * I;
                    ppp.D = pp.D - (k.x * pp.r.x + k.y * pp.r.y + k.z * pp.r.z);
                    ppp.historyIndex = pp.historyIndex;
                    ppp.nscatt = pp.nscatt + 1;
                    ppp.r = pp.r;
                    ppp.k = k;
                    ppp.hasObservedOpticalDepth = false;
                }
                detect(i, ppp, pixelOnDetector(ins, ppp.r));
            }
        }

        // ---- MediumSystem::simulateScattering -> DustMix::performScattering, HG (MediumSystem.cpp:796-823,
        //      DustMix.cpp:490-511, PhotonPacket.cpp:115-122)
        void simulateScattering(Packet& pp)
        {
            // the scattering component, drawn from the cumulative distribution of the components' scattering opacities in the
            // interaction cell with ONE uniform deviate (MediumSystem.cpp:806-817, NR::cdf, NR::locateClip)
            int hpick = 0;
            if (numMedia() > 1)
            {
                const int H = numMedia();
                double Xv[PMC_MAX_MEDIA + 1];
                Xv[0] = 0.;
                for (int h = 0; h != H; ++h) Xv[h + 1] = Xv[h] + opacityOf(h, med(h).sigma_sca, pp.lambda, pp.interactionCell);
                double norm = Xv[H];
                for (int h = 0; h <= H; ++h) Xv[h] /= norm;
                hpick = locateClip(Xv, H + 1, rng.uniform());
            }
            double g = med(hpick).asymmpar[indexForLambda(hpick, pp.lambda)];
            V3 knew;
            if (fabs(g) < 1e-6)
                knew = randomDirection(rng);
            else
            {
                double f = ((1.0 - g) * (1.0 + g)) / (1.0 - g + 2.0 * g * rng.uniform());
                double costheta = (1.0 + g * g - f * f) / (2.0 * g);
                knew = randomDirectionAbout(rng, pp.k, costheta);
            }
            pp.nscatt++;
            pp.k = knew;
            pp.hasObservedOpticalDepth = false;
            counters.scatterings++;
        }

        // ---- MonteCarloSimulation::performLifeCycle (MonteCarloSimulation.cpp:538-613), primary, peel-off, no RF
        void run(uint64_t first, uint64_t count)
        {
            Packet pp, ppp;
            pp.segments.reserve(1000);
            for (uint64_t historyIndex = first; historyIndex != first + count; ++historyIndex)
            {
                rng.beginHistory(historyIndex);
                launch(pp, historyIndex);
                if (pp.luminosity() > 0)
                {
                    peelOffEmission(pp, ppp);
                    if (sc.options.force_scattering)
                    {
                        double Lthreshold = pp.luminosity() / sc.options.min_weight_reduction;
                        int minScattEvents = sc.options.min_scatt_events;
                        while (true)
                        {
                            setExtinctionOpticalDepths(pp);
                            if (radiationField) storeRadiationField(pp);
                            simulateForcedPropagation(pp);
                            if (pp.luminosity() <= 0 || (pp.luminosity() <= Lthreshold && pp.nscatt >= minScattEvents)) break;
                            peelOffScattering(pp, ppp);
                            simulateScattering(pp);
                        }
                    }
                    else
                    {
                        while (true)
                        {
                            if (!simulateNonForcedPropagation(pp)) break;
                            if (pp.luminosity() <= 0) break;
                            peelOffScattering(pp, ppp);
                            simulateScattering(pp);
                        }
                    }
                }
            }
        }

        // ---- InstrumentSystem::flush (FluxRecorder.cpp:472-480)
        void flush()
        {
            for (int i = 0; i < sc.num_instruments; ++i)
                if (sc.instruments[i].record_statistics)
                {
                    recordContributions(i);
                    lists[i].historyIndex = 0;
                    lists[i].contributions.clear();
                }
        }
    };
}

extern "C" {

// rng_kind 0: mt19937_64 seeded from `seed` with `skip_draws` deviates discarded (the single-thread reference stream);
// rng_kind 1: per-history Philox streams keyed by `seed` (the engine's streams).
// frames: caller-allocated, accumulated into.  The per-history statistics lists are flushed before returning, so a
// history range must not be split across calls with rng_kind 0 if bit-exact statistics are wanted.
int oracle_run_primary(const pmc_scene* scene, uint64_t first, uint64_t count, int rng_kind, uint64_t seed,
                       uint64_t skip_draws, double* frames, pmc_counter_values* counters)
{
    if (!scene || !frames) return PMC_ERR_INVALID;
    std::unique_ptr<Rng> rng;
    if (rng_kind == 0)
        rng.reset(new MtRng(static_cast<int>(seed), skip_draws));
    else
        rng.reset(new PhiloxRng(seed));
    LifeCycle cycle(*scene, *rng, frames);
    // the reference's contribution list starts with history index 0 and an empty list (FluxRecorder.hpp:335)
    cycle.run(first, count);
    cycle.flush();
    if (counters) *counters = cycle.counters;
    return PMC_OK;
}

// the same with the radiation field table rf[m * num_lambda + ell] (scene->radiation_field.store must be set)
int oracle_run_primary_rf(const pmc_scene* scene, uint64_t first, uint64_t count, int rng_kind, uint64_t seed,
                          uint64_t skip_draws, double* frames, double* rf, pmc_counter_values* counters)
{
    if (!scene || !frames || !rf || !scene->radiation_field.store) return PMC_ERR_INVALID;
    std::unique_ptr<Rng> rng;
    if (rng_kind == 0)
        rng.reset(new MtRng(static_cast<int>(seed), skip_draws));
    else
        rng.reset(new PhiloxRng(seed));
    LifeCycle cycle(*scene, *rng, frames);
    cycle.radiationField = rf;
    cycle.run(first, count);
    cycle.flush();
    if (counters) *counters = cycle.counters;
    return PMC_OK;
}

int oracle_trace_ray(const pmc_scene* scene, const double r[3], const double k[3], int32_t* m, double* ds, int32_t cap,
                     int32_t* n)
{
    if (!scene) return PMC_ERR_INVALID;
    std::unique_ptr<Generator> gen;
    if (scene->grid.kind == PMC_GRID_CARTESIAN)
        gen.reset(new CartesianGenerator(scene->grid));
    else if (scene->grid.kind == PMC_GRID_VORONOI)
        gen.reset(new VoronoiGenerator(scene->grid));
    else
        gen.reset(new TreeGenerator(scene->grid));
    gen->start(V3{r[0], r[1], r[2]}, V3{k[0], k[1], k[2]});
    int32_t count = 0;
    while (gen->next())
    {
        if (count < cap)
        {
            m[count] = gen->m;
            ds[count] = gen->ds;
        }
        ++count;
    }
    *n = count;
    return PMC_OK;
}
}
