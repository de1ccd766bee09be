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

    // A normalised density distribution: value, column densities along the coordinate axes through the origin (what the
    // material normalisations need), and random positions drawn from it.  Every class evaluates its formulas in the
    // operation order of the SKIRT class it stands for, so that setup-time tables come out bit-identical.
    class Geometry
    {
    public:
        virtual ~Geometry() {}
        virtual std::string type() const = 0;
        virtual double density(Vec3 r) const = 0;
        virtual double columnX() const = 0;
        virtual double columnY() const = 0;
        virtual double columnZ() const = 0;
        // a random position drawn from the density; consumes the simulation's random stream in the reference's order
        virtual Vec3 samplePosition(Random& random) const = 0;
    };

    // deprojected Sersic profile and its cumulative mass, tabulated on 101 logarithmic radii (SersicFunction.cpp:13-101)
    class SersicProfile
    {
    public:
        explicit SersicProfile(double index);
        double value(double s) const;          // S(s)
        double radiusOfMass(double M) const;   // inverse of the cumulative mass
        const Array& radii() const { return radius_; }
        const Array& masses() const { return mass_; }
        static double shapeB(double index);    // the b(n) series of the Sersic law

    private:
        double deproject(double index, double b, double central, double s) const;
        Array radius_, profile_, mass_;
    };

    class UniformBoxGeometry : public Geometry
    {
    public:
        explicit UniformBoxGeometry(const Box& box);
        std::string type() const override { return "UniformBoxGeometry"; }
        double density(Vec3 r) const override;
        double columnX() const override;
        double columnY() const override;
        double columnZ() const override;
        Vec3 samplePosition(Random& random) const override { return random.position(bounds_); }  // UniformBoxGeometry.cpp:37-40
        const Box& box() const { return bounds_; }

    private:
        Box bounds_;
        double level_{0};
    };

    // double-exponential disk with optional inner cavity and truncations (ExpDiskGeometry.cpp:13-90)
    class ExpDiskGeometry : public Geometry
    {
    public:
        ExpDiskGeometry(double radialScale, double verticalScale, double innerRadius, double outerRadius, double maxHeight);
        std::string type() const override { return "ExpDiskGeometry"; }
        double density(Vec3 r) const override;
        double columnX() const override { return 2.0 * radialColumn(); }
        double columnY() const override { return 2.0 * radialColumn(); }
        double columnZ() const override;
        double radialColumn() const;
        Vec3 samplePosition(Random& random) const override;  // SepAxGeometry.cpp:11-19, ExpDiskGeometry.cpp:46-68
        void parameters(double v[5]) const { v[0] = scaleR_, v[1] = scaleZ_, v[2] = innerR_, v[3] = outerR_, v[4] = maxZ_; }

    private:
        double scaleR_, scaleZ_, innerR_, outerR_, maxZ_, central_;
    };

    class SersicGeometry : public Geometry
    {
    public:
        SersicGeometry(double effectiveRadius, double index);
        std::string type() const override { return "SersicGeometry"; }
        double density(Vec3 r) const override;
        double columnX() const override { return 2.0 * radialColumn(); }
        double columnY() const override { return 2.0 * radialColumn(); }
        double columnZ() const override { return 2.0 * radialColumn(); }
        double radialColumn() const;
        Vec3 samplePosition(Random& random) const override;  // SpheGeometry.cpp:25-32, SersicGeometry.cpp:41-45
        double effectiveRadius() const { return reff_; }
        const SersicProfile& profile() const { return *profile_; }

    private:
        double reff_, index_, central_, b_;
        std::unique_ptr<SersicProfile> profile_;
    };

    class PlummerGeometry : public Geometry
    {
    public:
        explicit PlummerGeometry(double scale);
        std::string type() const override { return "PlummerGeometry"; }
        double density(Vec3 r) const override;
        double columnX() const override { return 2.0 * radialColumn(); }
        double columnY() const override { return 2.0 * radialColumn(); }
        double columnZ() const override { return 2.0 * radialColumn(); }
        double radialColumn() const;
        Vec3 samplePosition(Random& random) const override;  // SpheGeometry.cpp:25-32, PlummerGeometry.cpp:29-33
        double scaleLength() const { return scale_; }

    private:
        double scale_, central_;
    };

    // GaussianGeometry (GaussianGeometry.cpp:11-60): rho(r) = rho0 exp(-r^2 / 2 sigma^2); radii drawn from a 401-point table of the cumulative mass
    class GaussianGeometry : public Geometry
    {
    public:
        explicit GaussianGeometry(double dispersion);
        std::string type() const override { return "GaussianGeometry"; }
        double density(Vec3 r) const override;
        double columnX() const override { return 2.0 * radialColumn(); }
        double columnY() const override { return 2.0 * radialColumn(); }
        double columnZ() const override { return 2.0 * radialColumn(); }
        double radialColumn() const { return 1.0 / (4.0 * M_PI * sigma_ * sigma_); }
        Vec3 samplePosition(Random& random) const override;  // SpheGeometry.cpp:25-32, Random::cdfLinLin (Random.cpp:201-206)

    private:
        double sigma_, central_;
        std::vector<double> rv_, Xv_;
    };

    // SpheroidalGeometryDecorator (SpheroidalGeometryDecorator.cpp:11-43): a spherical geometry flattened along z
    class SpheroidalGeometry : public Geometry
    {
    public:
        SpheroidalGeometry(std::unique_ptr<Geometry> spherical, double flattening) : sphere_(std::move(spherical)), flat_(flattening) {}
        std::string type() const override { return "SpheroidalGeometryDecorator"; }
        double density(Vec3 r) const override;
        double columnX() const override { return 2.0 * radialColumn(); }
        double columnY() const override { return 2.0 * radialColumn(); }
        double columnZ() const override { return 2.0 * (sphere_->columnX() / 2.0); }
        double radialColumn() const { return 1.0 / flat_ * (sphere_->columnX() / 2.0); }
        Vec3 samplePosition(Random& random) const override;
        const Geometry* inner() const { return sphere_.get(); }
        double flattening() const { return flat_; }

    private:
        std::unique_ptr<Geometry> sphere_;
        double flat_;
    };

    // OffsetGeometryDecorator (OffsetGeometryDecorator.cpp:18-52): any geometry shifted by a fixed vector
    class OffsetGeometry : public Geometry
    {
    public:
        OffsetGeometry(std::unique_ptr<Geometry> geometry, Vec3 shift) : base_(std::move(geometry)), shift_(shift) {}
        std::string type() const override { return "OffsetGeometryDecorator"; }
        double density(Vec3 r) const override { return base_->density(Vec3{r.x - shift_.x, r.y - shift_.y, r.z - shift_.z}); }
        double columnX() const override { return base_->columnX(); }
        double columnY() const override { return base_->columnY(); }
        double columnZ() const override { return base_->columnZ(); }
        Vec3 samplePosition(Random& random) const override
        {
            const Vec3 p = base_->samplePosition(random);
            return Vec3{p.x + shift_.x, p.y + shift_.y, p.z + shift_.z};
        }
        const Geometry* inner() const { return base_.get(); }
        Vec3 offset() const { return shift_; }

    private:
        std::unique_ptr<Geometry> base_;
        Vec3 shift_;
    };

    // radial power law A r^-p between two radii: normalisation and radius sampling shared by the shell and the torus
    // (ShellGeometry.cpp:12-51, TorusGeometry.cpp:12-35,75-85)
    struct PowerLawRadius
    {
        double inner{0}, outer{0}, exponent{0};
        double logInner{0}, logSpan{0}, powInner{0}, powOuter{0};
        void prepare(double rmin, double rmax, double p);
        double sample(double X) const;
        double column(double amplitude) const;  // A * integral of r^-p over the radial range
    };

    // ShellGeometry (ShellGeometry.cpp:12-58): power-law shell A r^-p between two radii
    class ShellGeometry : public Geometry
    {
    public:
        ShellGeometry(double rmin, double rmax, double p);
        std::string type() const override { return "ShellGeometry"; }
        double density(Vec3 r) const override;
        double columnX() const override { return 2.0 * radialColumn(); }
        double columnY() const override { return 2.0 * radialColumn(); }
        double columnZ() const override { return 2.0 * radialColumn(); }
        double radialColumn() const { return radial_.column(amplitude_); }
        Vec3 samplePosition(Random& random) const override;  // SpheGeometry.cpp:25-32, ShellGeometry.cpp:38-51

    private:
        PowerLawRadius radial_;
        double amplitude_;
    };

    // TorusGeometry (TorusGeometry.cpp:12-108): A r^-p exp(-q |cos theta|) within the opening angle
    class TorusGeometry : public Geometry
    {
    public:
        TorusGeometry(double p, double q, double halfOpening, double rmin, double rmax, bool anisotropicInner, double cutoffRadius);
        std::string type() const override { return "TorusGeometry"; }
        double density(Vec3 r) const override;
        double columnX() const override { return 2.0 * radialColumn(); }
        double columnY() const override { return 2.0 * radialColumn(); }
        double columnZ() const override { return 0.0; }
        double radialColumn() const { return radial_.column(amplitude_); }
        Vec3 samplePosition(Random& random) const override;

    private:
        PowerLawRadius radial_;
        double polar_, sinOpening_, cutoff_, amplitude_;
        bool anisotropic_;
    };

    // RingGeometry (RingGeometry.cpp:13-77): Gaussian ring with an exponential vertical profile
    class RingGeometry : public Geometry
    {
    public:
        RingGeometry(double ringRadius, double width, double verticalScale);
        std::string type() const override { return "RingGeometry"; }
        double density(Vec3 r) const override;
        double columnX() const override { return 2.0 * radialColumn(); }
        double columnY() const override { return 2.0 * radialColumn(); }
        double columnZ() const override;
        double radialColumn() const;
        Vec3 samplePosition(Random& random) const override;  // SepAxGeometry.cpp:11-19, RingGeometry.cpp:50-61

    private:
        double centre_, width_, scaleZ_, amplitude_;
        Array tableR_, tableCdf_;
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
        int indexForLambda(double lambda) const { return tab::bracketClipped(lambdaBorder, lambda); }
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
        // DensityTreePolicy::_dustKappa (DensityTreePolicy.cpp:74-84): extinction cross section over mass per dust entity
        virtual double dustKappa(double lambda) const { return mix->sectionExt(lambda) / mix->mass(); }
    };

    // Several medium components seen as ONE dust distribution by the setup of the spatial grid: DensityTreePolicy sums the mass
    // densities of all dust media at a sample position (DensityTreePolicy.cpp:141), their masses (:72) and, for the optical depth
    // criterion, their cross sections and entity masses (:78-83).  The photon loop never sees this object: every component keeps
    // its own cell densities and material mix (MediumSystem.cpp:874-887).
    class CompositeMedium : public Medium
    {
    public:
        std::vector<Medium*> parts;
        std::string type() const override { return "media"; }
        void setup() override {}
        double numberDensity(Vec3) const override { throw std::runtime_error("number density of a composite medium"); }
        double massDensity(Vec3 r) const override
        {
            double rho = 0.;
            for (const Medium* part : parts) rho += part->massDensity(r);
            return rho;
        }
        double totalMass() const override
        {
            double sum = 0.;
            for (const Medium* part : parts) sum += part->totalMass();
            return sum;
        }
        double totalNumber() const override { return 0.; }
        Vec3 generatePosition(Random&) const override
        {
            throw std::runtime_error("unsupported: Voronoi sites drawn from the density of several medium components");
        }
        void massDensities(const std::vector<Vec3>& positions, std::vector<double>& out) const override
        {
            out.assign(positions.size(), 0.);
            std::vector<double> one;
            for (const Medium* part : parts)
            {
                part->massDensities(positions, one);
                for (size_t i = 0; i != out.size(); ++i) out[i] += one[i];
            }
        }
        double dustKappa(double lambda) const override
        {
            double sigma = 0., mu = 0.;
            for (const Medium* part : parts)
            {
                sigma += part->mix->sectionExt(lambda);
                mu += part->mix->mass();
            }
            return sigma / mu;
        }
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
        Vec3 generatePosition(Random& random) const override { return geometry->samplePosition(random); }
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
        bool relaxSites{false};  // VoronoiMeshSpatialGrid::relaxSites: one relaxation step of the sites before the final tessellation
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
        // emission direction of a point source: isotropic, or axisymmetric about `angularAxis` (AxAngularDistribution.cpp:12-18: normalised)
        int angularKind{0};  // PMC_ANGULAR_*
        Vec3 angularAxis{0., 0., 1.};
        double angularCosDelta{0.};
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
        // observer frame (DistantInstrument.cpp:24-36): an instrument at distance zero in a model at redshift z > 0
        double redshift{0}, luminosityDistance{0}, angularDiameterDistance{0};
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
