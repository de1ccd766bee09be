// simulation.hpp -- the MonteCarloSimulation of the host layer: ski file -> objects -> pmc_scene -> output files.
//
// Counterpart of SKIRT/core/MonteCarloSimulation (setupSimulation / runSimulation, MonteCarloSimulation.cpp:20-100)
// for the primary-emission path.  The photon loop itself is NOT here: the caller hands scene() to an engine
// (the HIP library behind include/pmc.h) and passes the resulting detector arrays to write().
#ifndef SKH_SIMULATION_HPP
#define SKH_SIMULATION_HPP

#include "model.hpp"
#include <cstdint>

namespace skh
{
    class Simulation
    {
    public:
        // parse the ski file and construct the item tree (XmlHierarchyCreator::readFile)
        static std::unique_ptr<Simulation> fromFile(const std::string& path);
        static std::unique_ptr<Simulation> fromString(const std::string& text, const std::string& prefix,
                                                      const std::string& inputPath = ".");

        // Simulation::setupSimulation: builds grids, densities, tables; consumes the parent random stream
        // exactly as the reference does with one thread.  If treeTopologyFile is non-empty the octree is rebuilt
        // from that TreeSpatialGridTopologyProbe file instead of by density sampling (no random draws for the tree).
        void setup();

        const pmc_scene& scene() const { return _scene; }
        uint64_t numPackets() const { return _numPackets; }
        int seed() const { return _seed; }
        // number of uniform deviates drawn from the parent stream during setup (the photon loop of the
        // single-thread reference continues the same stream from here)
        unsigned long long setupDraws() const { return _random.draws(); }
        const std::string& prefix() const { return _prefix; }

        int numInstruments() const { return static_cast<int>(_instruments.size()); }
        const InstrumentModel& instrument(int i) const { return _instruments[i]; }
        int64_t frameSize() const { return _frameSize; }
        const pmc_frame_layout& layout(int i) const { return _layouts[i]; }

        // FluxRecorder::calibrateAndWrite for every instrument (FluxRecorder.cpp:484-846); frames holds the raw
        // detector arrays in the pmc_frame_layout order and is calibrated IN PLACE.  Returns the files written.
        // writeStatistics = false: the flux files only (a segment whose statistics arrays are incomplete: the engine's list pool ran out)
        std::vector<std::string> write(double* frames, const std::string& outdir, bool writeStatistics = true) const;

        // Radiation field (RadiationFieldOptions::storeRadiationField): number of doubles of the table rf[m * nbins + ell]
        // that pmc_download_radiation_field fills (0 if not stored), and the RadiationFieldProbe / PerCellForm output
        // files "<prefix>_<probe>_J.dat" (RadiationFieldProbe.cpp:27-78, PerCellForm.cpp:14-32) written from it
        int64_t radiationFieldSize() const;
        std::vector<std::string> writeRadiationField(const double* rf, const std::string& outdir) const;

        // human-readable summary (grid size, tables, ...) for logs and tests
        std::string summary() const;

        // model objects, exposed for tests
        const SpatialGrid& grid() const { return *_grid; }
        const Medium& medium() const { return *_medium; }
        const Array& numberDensity(size_t h = 0) const { return _density[h]; }
        const SourceModel& source(int h = 0) const { return _sources[h]; }
        int numSources() const { return static_cast<int>(_sources.size()); }

        // optional overrides applied before setup()
        void setNumPackets(uint64_t n) { _numPackets = n; }
        // evaluate the densities of an imported particle medium on the GPU during setup (include/pmc.h pmc_sampler_*);
        // ignored for other media
        void setParticleSampler(const ParticleSamplerApi& api) { _samplerApi = api; }
        void setTreeTopology(std::vector<char> topology) { _topology = std::move(topology); }

    private:
        Simulation() {}
        void parse(const XmlElement& root);
        void buildScene();
        void flattenSourceGeometry(const SourceModel& source, pmc_source& flat) const;

        std::string _prefix;
        std::string _inputPath{"."};
        ParticleSamplerApi _samplerApi;
        OutputUnits _units;
        int _seed{0};
        // cosmology (Configuration.cpp:52-55): redshift of the model and the two distances of an observer at that redshift
        double _modelRedshift{0}, _cosmoAngularDiameterDistance{0}, _cosmoLuminosityDistance{0};
        Random _random;
        bool _oligo{false};
        uint64_t _numPackets{0};
        // source system
        double _ssMinWavelength{0.09e-6}, _ssMaxWavelength{100e-6}, _sourceBias{0.5};
        Array _oligoWavelengths;
        std::vector<SourceModel> _sources;
        // medium system
        pmc_options _options{};
        int _numDensitySamples{100};
        bool _hasMedium{true};                              // Configuration::hasMedium(): false in the NoMedium simulation modes
        std::unique_ptr<XmlElement> _standInMediumSystem;   // (those modes: the empty one-cell medium system the engine runs with)
        std::vector<std::unique_ptr<Medium>> _media;   // the medium components, in ski order (MediumSystem::_media)
        CompositeMedium _composite;                    // all of them as one dust distribution (grid setup)
        Medium* _medium{nullptr};                      // the only component, or the composite
        std::unique_ptr<SpatialGrid> _grid;
        std::vector<char> _topology;
        std::vector<Array> _density;                   // [component][cell] number densities
        std::vector<pmc_medium> _sceneMedia;
        // radiation field
        bool _storeRadiationField{false};
        std::unique_ptr<WavelengthGrid> _rfGridOwn;       // radiationFieldWLG as configured (panchromatic)
        const WavelengthGrid* _rfGrid{nullptr};           // Configuration::radiationFieldWLG()
        std::vector<std::string> _rfProbeNames;           // RadiationFieldProbe items with a PerCellForm
        // instruments
        std::unique_ptr<WavelengthGrid> _defaultGrid;     // as configured in the ski (panchromatic)
        std::unique_ptr<WavelengthGrid> _oligoGrid;       // OligoWavelengthGrid
        std::vector<InstrumentModel> _instruments;
        // derived tables referenced by the scene
        // per source: wavelength sampling tables and luminosity; flattened sources and the history index boundaries
        struct SourceTables
        {
            Array oligoWeight, sedLambda, sedp, sedP;
            double luminosity{0};
        };
        std::vector<SourceTables> _sourceTables;
        std::vector<pmc_source> _sceneSources;
        std::vector<uint64_t> _sourceFirst;
        std::vector<pmc_instrument> _pmcInstruments;
        std::vector<pmc_frame_layout> _layouts;
        int64_t _frameSize{0};
        pmc_scene _scene{};
    };

    // FITS primary image (BITPIX -32) + ASCII table of the third axis, byte-compatible with what the reference
    // produces through CFITSIO (FITSInOut.cpp:127-215)
    struct FitsObserverInfo
    {
        double inclination, azimuth, roll, redshift, luminosityDistance, angularDiameterDistance;
        std::string distanceUnits;
    };
    void writeFitsCube(const std::string& filepath, const double* data, const std::string& dataUnits, int nx, int ny,
                       double incx, double incy, double xc, double yc, const std::string& xyUnits, const Array& z,
                       const std::string& zUnits, const FitsObserverInfo* obsInfo);
}

#endif
