// model.hpp -- host model layer: the SKIRT-equivalent objects a ski file describes, for the classes the
// primary-emission path supports, and their flattening into the POD tables of include/pmc.h.
//
// Class names and semantics follow the reference (SKIRT/core); each implementation cites the code it restates.
// Unsupported ski classes raise a std::runtime_error naming the class ("not supported on the MI355X path").
#ifndef SKH_MODEL_HPP
#define SKH_MODEL_HPP

#include "../../include/pmc.h"
#include "mathutil.hpp"
#include "particles.hpp"
#include "voronoi.hpp"
#include "xml.hpp"
#include <memory>
#include <string>
#include <vector>

namespace skh
{
    // ---------------------------------------------------------------- units selected in the ski file

    struct OutputUnits
    {
        std::string system{"ExtragalacticUnits"};  // SIUnits | StellarUnits | ExtragalacticUnits
        std::string wavelengthStyle{"Wavelength"};
        std::string fluxStyle{"Frequency"};  // Neutral | Wavelength | Frequency
        double out(const std::string& qty, double value) const;
        std::string unit(const std::string& qty) const;
        // Units.cpp:111-160, 529-609
        std::string swavelength() const { return "lambda"; }
        std::string uwavelength() const { return unit("wavelength"); }
        double owavelength(double lambda) const { return out("wavelength", lambda); }
        std::string sfluxdensity() const;
        std::string ufluxdensity() const;
        double ofluxdensity(double lambda, double Flambda) const;
        std::string usurfacebrightness() const;
        double osurfacebrightness(double lambda, double flambda) const;
        std::string smeanintensity() const;  // Units.cpp:613-651
        std::string umeanintensity() const;
        double omeanintensity(double lambda, double Jlambda) const;
    };

    // ---------------------------------------------------------------- geometries

    class Geometry
    {
    public:
        virtual ~Geometry() {}
        virtual std::string type() const = 0;
        virtual double density(Vec3 r) const = 0;
        virtual double SigmaX() const = 0;
        virtual double SigmaY() const = 0;
        virtual double SigmaZ() const = 0;
        // Geometry::generatePosition: a random position drawn from the density (consumes the simulation's random stream)
        virtual Vec3 generatePosition(Random& random) const = 0;
    };

    // SKIRT/utils/SersicFunction.cpp:13-101
    class SersicFunction
    {
    public:
        explicit SersicFunction(double n);
        double operator()(double s) const;
        double inverseMass(double M) const;
        const Array& sv() const { return _sv; }
        const Array& Mv() const { return _Mv; }

    private:
        Array _sv, _Sv, _Mv;
    };

    class UniformBoxGeometry : public Geometry
    {
    public:
        explicit UniformBoxGeometry(const Box& box);
        std::string type() const override { return "UniformBoxGeometry"; }
        double density(Vec3 r) const override;
        double SigmaX() const override;
        double SigmaY() const override;
        double SigmaZ() const override;
        Vec3 generatePosition(Random& random) const override { return random.position(_box); }  // UniformBoxGeometry.cpp:37-40
        const Box& box() const { return _box; }

    private:
        Box _box;
        double _rho{0};
    };

    class ExpDiskGeometry : public Geometry
    {
    public:
        ExpDiskGeometry(double hR, double hz, double Rmin, double Rmax, double zmax);
        std::string type() const override { return "ExpDiskGeometry"; }
        double density(Vec3 r) const override;
        double SigmaX() const override { return 2.0 * SigmaR(); }
        double SigmaY() const override { return 2.0 * SigmaR(); }
        double SigmaZ() const override;
        double SigmaR() const;
        Vec3 generatePosition(Random& random) const override;  // SepAxGeometry.cpp:11-19, ExpDiskGeometry.cpp:46-68
        void parameters(double v[5]) const { v[0] = _hR, v[1] = _hz, v[2] = _Rmin, v[3] = _Rmax, v[4] = _zmax; }

    private:
        double _hR, _hz, _Rmin, _Rmax, _zmax, _rho0;
    };

    class SersicGeometry : public Geometry
    {
    public:
        SersicGeometry(double reff, double n);
        std::string type() const override { return "SersicGeometry"; }
        double density(Vec3 r) const override;
        double SigmaX() const override { return 2.0 * Sigmar(); }
        double SigmaY() const override { return 2.0 * Sigmar(); }
        double SigmaZ() const override { return 2.0 * Sigmar(); }
        double Sigmar() const;
        Vec3 generatePosition(Random& random) const override;  // SpheGeometry.cpp:25-32, SersicGeometry.cpp:41-45
        double reff() const { return _reff; }
        const SersicFunction& function() const { return *_function; }

    private:
        double _reff, _n, _rho0, _b;
        std::unique_ptr<SersicFunction> _function;
    };

    class PlummerGeometry : public Geometry
    {
    public:
        explicit PlummerGeometry(double c);
        std::string type() const override { return "PlummerGeometry"; }
        double density(Vec3 r) const override;
        double SigmaX() const override { return 2.0 * Sigmar(); }
        double SigmaY() const override { return 2.0 * Sigmar(); }
        double SigmaZ() const override { return 2.0 * Sigmar(); }
        double Sigmar() const;
        Vec3 generatePosition(Random& random) const override;  // SpheGeometry.cpp:25-32, PlummerGeometry.cpp:29-33
        double scaleLength() const { return _c; }

    private:
        double _c, _rho0;
    };

    // SpheroidalGeometryDecorator (SpheroidalGeometryDecorator.cpp:11-43): a spherical geometry flattened along z by q
    class SpheroidalGeometry : public Geometry
    {
    public:
        SpheroidalGeometry(std::unique_ptr<Geometry> spherical, double q) : _inner(std::move(spherical)), _q(q) {}
        std::string type() const override { return "SpheroidalGeometryDecorator"; }
        double density(Vec3 r) const override;
        double SigmaX() const override { return 2.0 * SigmaR(); }
        double SigmaY() const override { return 2.0 * SigmaR(); }
        double SigmaZ() const override { return 2.0 * (_inner->SigmaX() / 2.0); }
        double SigmaR() const { return 1.0 / _q * (_inner->SigmaX() / 2.0); }
        Vec3 generatePosition(Random& random) const override;
        const Geometry* inner() const { return _inner.get(); }
        double flattening() const { return _q; }

    private:
        std::unique_ptr<Geometry> _inner;
        double _q;
    };

    // OffsetGeometryDecorator (OffsetGeometryDecorator.cpp:18-52): any geometry shifted by a fixed vector
    class OffsetGeometry : public Geometry
    {
    public:
        OffsetGeometry(std::unique_ptr<Geometry> geometry, Vec3 offset) : _inner(std::move(geometry)), _offset(offset) {}
        std::string type() const override { return "OffsetGeometryDecorator"; }
        double density(Vec3 r) const override { return _inner->density(Vec3{r.x - _offset.x, r.y - _offset.y, r.z - _offset.z}); }
        double SigmaX() const override { return _inner->SigmaX(); }
        double SigmaY() const override { return _inner->SigmaY(); }
        double SigmaZ() const override { return _inner->SigmaZ(); }
        Vec3 generatePosition(Random& random) const override
        {
            Vec3 r = _inner->generatePosition(random);
            return Vec3{r.x + _offset.x, r.y + _offset.y, r.z + _offset.z};
        }
        const Geometry* inner() const { return _inner.get(); }
        Vec3 offset() const { return _offset; }

    private:
        std::unique_ptr<Geometry> _inner;
        Vec3 _offset;
    };

    // ShellGeometry (ShellGeometry.cpp:12-58): power-law shell A r^-p between two radii
    class ShellGeometry : public Geometry
    {
    public:
        ShellGeometry(double rmin, double rmax, double p);
        std::string type() const override { return "ShellGeometry"; }
        double density(Vec3 r) const override;
        double SigmaX() const override { return 2.0 * Sigmar(); }
        double SigmaY() const override { return 2.0 * Sigmar(); }
        double SigmaZ() const override { return 2.0 * Sigmar(); }
        double Sigmar() const;
        Vec3 generatePosition(Random& random) const override;  // SpheGeometry.cpp:25-32, ShellGeometry.cpp:38-51

    private:
        double _rmin, _rmax, _p, _smin, _sdiff, _tmin, _tmax, _A;
    };

    // TorusGeometry (TorusGeometry.cpp:12-108): A r^-p exp(-q |cos theta|) within the opening angle
    class TorusGeometry : public Geometry
    {
    public:
        TorusGeometry(double p, double q, double Delta, double rmin, double rmax, bool rani, double rcut);
        std::string type() const override { return "TorusGeometry"; }
        double density(Vec3 r) const override;
        double SigmaX() const override { return 2.0 * SigmaR(); }
        double SigmaY() const override { return 2.0 * SigmaR(); }
        double SigmaZ() const override { return 0.0; }
        double SigmaR() const;
        Vec3 generatePosition(Random& random) const override;

    private:
        double _p, _q, _Delta, _rmin, _rmax;
        bool _rani;
        double _rcut, _sinDelta, _smin, _sdiff, _tmin, _tmax, _A;
    };

    // RingGeometry (RingGeometry.cpp:13-77): Gaussian ring with an exponential vertical profile
    class RingGeometry : public Geometry
    {
    public:
        RingGeometry(double R0, double w, double hz);
        std::string type() const override { return "RingGeometry"; }
        double density(Vec3 r) const override;
        double SigmaX() const override { return 2.0 * SigmaR(); }
        double SigmaY() const override { return 2.0 * SigmaR(); }
        double SigmaZ() const override;
        double SigmaR() const;
        Vec3 generatePosition(Random& random) const override;  // SepAxGeometry.cpp:11-19, RingGeometry.cpp:50-61

    private:
        double _R0, _w, _hz, _A;
        Array _Rv, _Xv;
    };

    // ---------------------------------------------------------------- dust mix (tabulated mean properties)

    // DustMix tables for a TabulatedDustMix subclass (MeanListDustMix / MeanFileDustMix):
    // SKIRT/core/DustMix.cpp:47-162, TabulatedDustMix.cpp:12-45, MeanListDustMix.cpp:12-27
    class DustMix
    {
    public:
        std::string typeName;
        Array inLambda, inKappaExt, inAlbedo, inAsymmpar;  // as configured
        double mu{1.5e-29};
        // built by setup()
        Array lambdaSample;  // sampling wavelengths (DustMix.cpp local lambdav)
        Array lambdaBorder;  // DustMix::_lambdav (shifted borders used by indexForLambda)
        Array sigmaAbs, sigmaSca, sigmaExt, asymmpar;

        void setup(double rangeMin, double rangeMax, const std::vector<double>& simulationWavelengths);
        int indexForLambda(double lambda) const { return nr::locateClip(lambdaBorder, lambda); }
        double sectionExt(double lambda) const { return sigmaExt[indexForLambda(lambda)]; }
        double sectionSca(double lambda) const { return sigmaSca[indexForLambda(lambda)]; }
        double mass() const { return mu; }
    };

    // ---------------------------------------------------------------- medium

    // what the grid policy and the density sampling ask of a medium (SKIRT/core/Medium.hpp)
    class Medium
    {
    public:
        virtual ~Medium() {}
        std::unique_ptr<DustMix> mix;
        virtual std::string type() const = 0;
        virtual void setup() = 0;
        virtual double numberDensity(Vec3 r) const = 0;
        virtual double massDensity(Vec3 r) const = 0;
        virtual double totalMass() const = 0;
        virtual double totalNumber() const = 0;
        // Medium::generatePosition (sites of a Voronoi grid with the DustDensity policy)
        virtual Vec3 generatePosition(Random& random) const = 0;
        // many positions at once (the setup phase samples 100 per tree node and per cell): on the host cores by
        // default; a medium may hand the batch to the GPU
        virtual void massDensities(const std::vector<Vec3>& positions, std::vector<double>& out) const
        {
            out.resize(positions.size());
            parallelFor(positions.size(), [&](size_t b, size_t e) {
                for (size_t i = b; i != e; ++i) out[i] = massDensity(positions[i]);
            });
        }
        virtual void numberDensities(const std::vector<Vec3>& positions, std::vector<double>& out) const
        {
            out.resize(positions.size());
            parallelFor(positions.size(), [&](size_t b, size_t e) {
                for (size_t i = b; i != e; ++i) out[i] = numberDensity(positions[i]);
            });
        }
        // wavelength that the medium's normalisation adds to the simulation wavelengths (0: none)
        virtual double normalizationWavelength() const { return 0.; }
    };

    // GeometricMedium with OpticalDepth/Mass/Number material normalisation
    // SKIRT/core/GeometricMedium.cpp:13-18,128-131; OpticalDepthMaterialNormalization.cpp:13-27
    class GeometricMedium : public Medium
    {
    public:
        std::unique_ptr<Geometry> geometry;
        std::string normType;  // OpticalDepthMaterialNormalization | MassMaterialNormalization | NumberMaterialNormalization
        char normAxis{'Z'};
        double normWavelength{0.55e-6};
        double normOpticalDepth{0};
        double normMass{0};
        double normNumber{0};
        double number{0}, mass{0};  // results of setup()

        std::string type() const override { return "GeometricMedium"; }
        void setup() override;
        double numberDensity(Vec3 r) const override { return number * geometry->density(r); }
        double massDensity(Vec3 r) const override { return mass * geometry->density(r); }
        double totalMass() const override { return mass; }
        double totalNumber() const override { return number; }
        Vec3 generatePosition(Random& random) const override { return geometry->generatePosition(random); }
        double normalizationWavelength() const override
        {
            return normType == "OpticalDepthMaterialNormalization" ? normWavelength : 0.;
        }
    };

    // ParticleMedium: a dust medium imported from smoothed particles (SKIRT/core/ParticleMedium.cpp:12-63,
    // ImportedMedium.cpp:12-55,188-214); see particles.hpp
    class ParticleMedium : public Medium
    {
    public:
        ParticleImportOptions options;
        std::string kernelType{"CubicSplineSmoothingKernel"};
        ParticleSnapshot snapshot;

        std::string type() const override { return "ParticleMedium"; }
        void setup() override { snapshot.load(options, SmoothingKernel::create(kernelType)); }
        double numberDensity(Vec3 r) const override
        {
            double result = snapshot.density(r);
            if (!snapshot.holdsNumber()) result /= mix->mass();
            return result;
        }
        double massDensity(Vec3 r) const override
        {
            double result = snapshot.density(r);
            if (snapshot.holdsNumber()) result *= mix->mass();
            return result;
        }
        double totalMass() const override
        {
            double result = snapshot.mass();
            if (snapshot.holdsNumber()) result *= mix->mass();
            return result;
        }
        double totalNumber() const override
        {
            double result = snapshot.mass();
            if (!snapshot.holdsNumber()) result /= mix->mass();
            return result;
        }
        Vec3 generatePosition(Random&) const override
        {
            throw std::runtime_error("random positions from an imported particle medium are not supported on the MI355X path");
        }
        void massDensities(const std::vector<Vec3>& positions, std::vector<double>& out) const override
        {
            snapshot.densities(positions, out);
            if (snapshot.holdsNumber())
                for (double& v : out) v *= mix->mass();
        }
        void numberDensities(const std::vector<Vec3>& positions, std::vector<double>& out) const override
        {
            snapshot.densities(positions, out);
            if (!snapshot.holdsNumber())
                for (double& v : out) v /= mix->mass();
        }
    };

    // ---------------------------------------------------------------- spatial grids

    class SpatialGrid
    {
    public:
        virtual ~SpatialGrid() {}
        Box extent;
        virtual int numCells() const = 0;
        virtual Box cellBox(int m) const = 0;  // box cells: the cell; Voronoi: the bounding box of the cell
        virtual void fill(pmc_grid& g) const = 0;
        // SpatialGrid::volume / centralPositionInCell / randomPositionInCell
        virtual double volume(int m) const { return cellBox(m).volume(); }
        virtual Vec3 centralPositionInCell(int m) const { return cellBox(m).center(); }
        virtual Vec3 randomPositionInCell(int m, Random& random) const { return random.position(cellBox(m)); }
    };

    // CartesianSpatialGrid with LinMesh axes (CartesianSpatialGrid.cpp:14-28, LinMesh.cpp:11-16)
    class CartesianSpatialGrid : public SpatialGrid
    {
    public:
        int nx{0}, ny{0}, nz{0};
        // the Mesh of every axis: LinMesh | PowMesh | SymPowMesh | LogMesh | SymLogMesh (mesh() of the class, on [0,1])
        struct MeshSpec
        {
            std::string type{"LinMesh"};
            double ratio{1};                  // PowMesh, SymPowMesh: last/first resp. outermost/innermost bin width
            double centralBinFraction{1e-3};  // LogMesh, SymLogMesh
            std::vector<double> points;       // ListMesh: the normalised border points (TabulatedMesh.cpp:12-33)
        };
        MeshSpec meshSpec[3];
        Array xv, yv, zv;
        void setup();
        int numCells() const override { return nx * ny * nz; }
        Box cellBox(int m) const override;
        void fill(pmc_grid& g) const override;
    };

    // PolicyTreeSpatialGrid (treeType OctTree) with DensityTreePolicy
    // TreeSpatialGrid.cpp:23-78, TreeNode.cpp, OctTreeNode.cpp:22-138, DensityTreePolicy.cpp:117-309
    class OctreeSpatialGrid : public SpatialGrid
    {
    public:
        // policy configuration
        int minLevel{3}, maxLevel{7};
        double maxDustFraction{1e-6}, maxDustOpticalDepth{0}, policyWavelength{0.55e-6}, maxDustDensityDispersion{0};

        struct Node
        {
            Box box;
            int level{0};
            int parent{-1};
            int firstChild{-1};
            std::vector<int> neighbors[6];
        };
        std::vector<Node> nodes;
        std::vector<int> cellIndexOfNode;  // _cellindexv
        std::vector<int> nodeOfCell;       // _idv
        // flattened (built by setup)
        std::vector<double> flatBox;
        std::vector<int32_t> flatLevel, flatFirstChild, flatCell, flatNbrStart, flatNbrList;

        // builds the tree; consumes the random stream exactly like DensityTreePolicy::constructTree at one thread
        void setup(const Medium& medium, int numDensitySamples, Random& random);
        // builds the tree from a topology stream ("1"/"0" per node, depth first; TreeSpatialGrid.cpp:225-251) --
        // the node ids are then assigned breadth-first as the policy would have done for the same topology
        void setupFromTopology(const std::vector<char>& topology);
        int numCells() const override { return static_cast<int>(nodeOfCell.size()); }
        Box cellBox(int m) const override { return nodes[nodeOfCell[m]].box; }
        void fill(pmc_grid& g) const override;

    private:
        void subdivide(int id);
        void finish();
    };

    // VoronoiMeshSpatialGrid (VoronoiMeshSpatialGrid.cpp:42-121) with the policies Uniform (random sites) and File
    // (sites from a column text file); the tessellation itself is voronoi.hpp
    class VoronoiSpatialGrid : public SpatialGrid
    {
    public:
        std::string policy{"Uniform"};
        int numSites{500};
        std::string sitesPath;  // resolved file path (policy File)
        VoronoiMesh mesh;

        void setup(Random& random, const Medium& medium);
        int numCells() const override { return mesh.numCells(); }
        Box cellBox(int m) const override { return mesh.cellBox(m); }
        void fill(pmc_grid& g) const override;
        double volume(int m) const override { return mesh.volume(m); }
        Vec3 centralPositionInCell(int m) const override { return mesh.site(m); }
        // VoronoiMeshSnapshot::generatePosition(m) (:976-989): rejection sampling in the bounding box
        Vec3 randomPositionInCell(int m, Random& random) const override;
    };

    // ---------------------------------------------------------------- source

    struct SourceModel
    {
        std::string type;  // PointSource | GeometricSource
        Vec3 position;
        std::unique_ptr<Geometry> geometry;
        double sourceWeight{1.};
        double wavelengthBias{0.5};
        std::string biasDistType{"DefaultWavelengthDistribution"};
        double biasMin{0}, biasMax{0};
        // BlackBodySED, or a tabulated SED (ListSED / FileSED: wavelengths and specific luminosities per unit of wavelength,
        // arbitrarily scaled; TabulatedSED.cpp:12-25)
        std::string sedType{"BlackBodySED"};
        double temperature{5000.};
        Array sedInLambda, sedInP;
        // IntegratedLuminosityNormalization, or SpecificLuminosityNormalization (normType; specific luminosity per unit of
        // wavelength at normWavelength after Units::fromFluxStyle)
        std::string normType{"IntegratedLuminosityNormalization"};
        double normWavelength{0}, specificLuminosity{0};
        std::string normRange{"Source"};
        double normMinWavelength{0.09e-6}, normMaxWavelength{100e-6}, integratedLuminosity{0};
    };

    // ---------------------------------------------------------------- wavelength grid (disjoint bins)

    // DisjointWavelengthGrid.cpp:22-135,300-345
    struct WavelengthGrid
    {
        Array lambdav, lambdaleftv, lambdarightv, dlambdav, borderv;
        std::vector<int32_t> ellv;
        void setWavelengthRange(Array lambda, bool logScale);                              // Log/Lin/List grids
        void setWavelengthBins(Array lambda, double relativeHalfWidth, bool constantWidth);  // Oligo grid
        int numBins() const { return static_cast<int>(lambdav.size()); }
        double wavelength(int ell) const { return lambdav[ell]; }
        double effectiveWidth(int ell) const { return dlambdav[ell]; }
        double rangeMin() const { return lambdaleftv.front(); }
        double rangeMax() const { return lambdarightv.back(); }
        int bin(double lambda) const;
    };

    // ---------------------------------------------------------------- instruments

    struct InstrumentModel
    {
        std::string type;  // FrameInstrument | FullInstrument
        std::string name;
        double distance{0}, inclination{0}, azimuth{0}, roll{0};
        double fieldOfViewX{0}, fieldOfViewY{0}, centerX{0}, centerY{0};
        int numPixelsX{250}, numPixelsY{250};
        double radius{0};  // SEDInstrument: aperture radius (0: none)
        bool recordComponents{false}, recordPolarization{false}, recordStatistics{false};
        int numScatteringLevels{0};
        std::unique_ptr<WavelengthGrid> ownGrid;  // instrument-specific grid (panchromatic only)
        const WavelengthGrid* grid{nullptr};      // the grid in effect
        bool sameObserverAsPreceding{false};
    };
}

#endif
