// particles.hpp -- import of a smoothed-particle (SPH) medium: the host-side setup of BASELINE configs[3]
// ("SPH-imported medium with adaptive octree").  Restates, with the reference's arithmetic and ORDER of operations so
// that the sampled cell densities come out bit-identical:
//   column text file with unit header                SKIRT/core/TextInFile.cpp:16-47,57-110,214-330
//   column roles and default units                   SKIRT/core/Snapshot.cpp:62-85,130-180
//   mass policy (massFraction, metallicity, Tmax)    SKIRT/core/ImportedMedium.cpp:12-55, ParticleSnapshot.cpp:79-151
//   block search grid                                SKIRT/utils/BoxSearch.cpp:14-150
//   kernel-weighted density at a position            SKIRT/core/ParticleSnapshot.cpp:233-243
//   smoothing kernels                                SKIRT/core/CubicSplineSmoothingKernel.cpp:40-50,
//                                                    ScaledGaussianSmoothingKernel.cpp:16-47, UniformSmoothingKernel.cpp
// Not restated: massInBox (needs the absent "CumulativeKernels" resource; the reference then samples the density as
// this code does, SmoothingKernel.cpp:87-90), velocity / magnetic field / variable-mix columns, .scol binary files.
#ifndef SKH_PARTICLES_HPP
#define SKH_PARTICLES_HPP

#include "../../include/pmc.h"
#include "mathutil.hpp"
#include <memory>
#include <string>
#include <vector>

namespace skh
{
    class SmoothingKernel
    {
    public:
        virtual ~SmoothingKernel() {}
        virtual std::string type() const = 0;
        virtual double density(double u) const = 0;
        static std::unique_ptr<SmoothingKernel> create(const std::string& type);
    };

    // BoxSearch.cpp:54-122: entities (bounding boxes) binned in a numBlocks^3 grid whose separation points balance the
    // number of entity centres per slab; the per-block lists hold entity indices in ascending order
    class BoxSearch
    {
    public:
        template<class Bounds, class Intersects> void loadEntities(int numEntities, Bounds bounds, Intersects intersects);
        const std::vector<int>& entitiesFor(Vec3 r) const;
        int numBlocks() const { return _numBlocks; }
        const Box& extent() const { return _extent; }
        size_t numReferences() const;
        // flattened: separation arrays and the per-block lists as CSR (block b = (i*n + j)*n + k)
        const Array& xgrid() const { return _xgrid; }
        const Array& ygrid() const { return _ygrid; }
        const Array& zgrid() const { return _zgrid; }
        void flatten(std::vector<int64_t>& start, std::vector<int32_t>& list) const;

    private:
        int blockIndex(int i, int j, int k) const { return ((i * _numBlocks) + j) * _numBlocks + k; }
        Box _extent;
        int _numBlocks{0};
        Array _xgrid, _ygrid, _zgrid;
        std::vector<std::vector<int>> _listv;
        std::vector<int> _empty;
    };

    struct ParticleImportOptions
    {
        std::string path;            // resolved file path
        bool holdsNumber{false};     // massType Number: the mass column holds a number of entities
        double massFraction{1.};
        bool importMetallicity{false};
        bool importTemperature{false};
        double maxTemperature{0.};
        bool isDust{true};
    };

    // entry points of the device sampler (include/pmc.h pmc_sampler_*), handed over at run time so that this library
    // keeps no link-time dependency on the HIP engine
    struct ParticleSamplerApi
    {
        int (*create)(const pmc_particles*, int32_t, pmc_sampler**){nullptr};
        int (*density)(pmc_sampler*, const double*, int64_t, double*){nullptr};
        void (*destroy)(pmc_sampler*){nullptr};
        const char* (*lastError)(){nullptr};
        int32_t device{0};
    };

    class ParticleSnapshot
    {
    public:
        ~ParticleSnapshot();
        struct Particle
        {
            double x, y, z, h, M;
            double density() const { return M / (h * h * h); }
        };
        // reads the file and builds the search grid; throws std::runtime_error on malformed input
        void load(const ParticleImportOptions& options, std::unique_ptr<SmoothingKernel> kernel);
        double density(Vec3 r) const;  // ParticleSnapshot.cpp:233-243
        // the same for many positions: on the MI355X if a device sampler has been set (bit-identical results for the
        // cubic-spline and uniform kernels), otherwise on the host cores
        void densities(const std::vector<Vec3>& positions, std::vector<double>& out) const;
        void useDeviceSampler(const ParticleSamplerApi& api) { _api = api; }
        bool onDevice() const { return _sampler != nullptr; }
        double mass() const { return _mass; }
        bool holdsNumber() const { return _holdsNumber; }
        size_t numParticles() const { return _pv.size(); }
        const BoxSearch& search() const { return _search; }
        const std::vector<Particle>& particles() const { return _pv; }
        // the positions of ALL imported entities in file order, also those that carry no mass (SiteListInterface of ImportedMedium,
        // ImportedMedium.cpp:268-278: the sites of a Voronoi grid with the ImportedSites policy)
        const std::vector<Vec3>& sitePositions() const { return _sites; }
        const SmoothingKernel& kernel() const { return *_kernel; }

    private:
        std::vector<Particle> _pv;
        std::vector<Vec3> _sites;
        BoxSearch _search;
        std::unique_ptr<SmoothingKernel> _kernel;
        double _mass{0};
        bool _holdsNumber{false};
        ParticleSamplerApi _api;
        pmc_sampler* _sampler{nullptr};
    };

    // column text file (TextInFile): "# column N: description (unit)" header lines are optional; without them every
    // column carries the default unit its role declares
    struct ColumnSpec
    {
        std::string description, quantity, defaultUnit;
    };
    std::vector<Array> readColumnFile(const std::string& path, const std::vector<ColumnSpec>& columns,
                                      const std::string& description = "smoothed particles");
}

#endif
