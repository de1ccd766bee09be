// simulation.cpp -- ski parsing, setup and scene flattening (see simulation.hpp).  Citations: SKIRT 9 tree.

#include "simulation.hpp"
#include "../../include/pmc_layout.h"
#include "units.hpp"
#include <cstring>
#include <fstream>
#include <set>
#include <sstream>

namespace skh
{
    namespace
    {
        [[noreturn]] void unsupported(const std::string& what)
        {
            throw std::runtime_error("ski: " + what + " is not supported on the MI355X primary-emission path");
        }

        // attribute access with the reference's defaults (ATTRIBUTE_DEFAULT_VALUE in the class headers)
        struct Reader
        {
            std::string unitSystem;
            double quantity(const XmlElement& e, const char* key, const char* qty, const char* fallback = nullptr) const
            {
                if (e.has(key)) return parseQuantity(e.attr(key), qty, unitSystem);
                if (!fallback)
                    throw std::runtime_error("ski: <" + e.name + "> lacks the required attribute '" + key + "'");
                return parseQuantity(fallback, qty, unitSystem);
            }
            double number(const XmlElement& e, const char* key, const char* fallback = nullptr) const
            {
                return quantity(e, key, "", fallback);
            }
            int integer(const XmlElement& e, const char* key, int fallback) const
            {
                if (!e.has(key)) return fallback;
                return std::stoi(squeeze(e.attr(key)));
            }
            bool boolean(const XmlElement& e, const char* key, bool fallback) const
            {
                if (!e.has(key)) return fallback;
                std::string v = squeeze(e.attr(key));
                if (v == "true" || v == "True" || v == "1" || v == "yes") return true;
                if (v == "false" || v == "False" || v == "0" || v == "no") return false;
                throw std::runtime_error("ski: invalid boolean '" + v + "' for attribute " + key);
            }
            std::vector<double> list(const XmlElement& e, const char* key, const char* qty, const char* fallback) const
            {
                return parseQuantityList(e.has(key) ? e.attr(key) : std::string(fallback), qty, unitSystem);
            }
        };

        std::unique_ptr<Geometry> makeGeometry(const XmlElement& e, const Reader& rd)
        {
            if (e.name == "UniformBoxGeometry")
                return std::make_unique<UniformBoxGeometry>(
                    Box(rd.quantity(e, "minX", "length"), rd.quantity(e, "minY", "length"), rd.quantity(e, "minZ", "length"),
                        rd.quantity(e, "maxX", "length"), rd.quantity(e, "maxY", "length"), rd.quantity(e, "maxZ", "length")));
            if (e.name == "ExpDiskGeometry")
                return std::make_unique<ExpDiskGeometry>(rd.quantity(e, "scaleLength", "length"),
                                                         rd.quantity(e, "scaleHeight", "length"),
                                                         rd.quantity(e, "minRadius", "length", "0"),
                                                         rd.quantity(e, "maxRadius", "length", "0"),
                                                         rd.quantity(e, "maxZ", "length", "0"));
            if (e.name == "SersicGeometry")
                return std::make_unique<SersicGeometry>(rd.quantity(e, "effectiveRadius", "length"), rd.number(e, "index", "1"));
            if (e.name == "PlummerGeometry") return std::make_unique<PlummerGeometry>(rd.quantity(e, "scaleLength", "length"));
            if (e.name == "GaussianGeometry") return std::make_unique<GaussianGeometry>(rd.quantity(e, "dispersion", "length"));
            if (e.name == "OffsetGeometryDecorator")
            {
                const XmlElement* inner = e.item("geometry");
                if (!inner) throw std::runtime_error("ski: OffsetGeometryDecorator lacks a geometry");
                return std::make_unique<OffsetGeometry>(makeGeometry(*inner, rd),
                                                        Vec3{rd.quantity(e, "offsetX", "length", "0"), rd.quantity(e, "offsetY", "length", "0"),
                                                             rd.quantity(e, "offsetZ", "length", "0")});
            }
            if (e.name == "SpheroidalGeometryDecorator")
            {
                const XmlElement* inner = e.item("geometry");
                if (!inner) throw std::runtime_error("ski: SpheroidalGeometryDecorator lacks a geometry");
                if (inner->name != "SersicGeometry" && inner->name != "PlummerGeometry" && inner->name != "ShellGeometry" && inner->name != "GaussianGeometry")
                    unsupported("SpheroidalGeometryDecorator of " + inner->name);
                return std::make_unique<SpheroidalGeometry>(makeGeometry(*inner, rd), rd.number(e, "flattening", "1"));
            }
            if (e.name == "ShellGeometry")
                return std::make_unique<ShellGeometry>(rd.quantity(e, "minRadius", "length"), rd.quantity(e, "maxRadius", "length"),
                                                       rd.number(e, "exponent", "2"));
            if (e.name == "TorusGeometry")
                return std::make_unique<TorusGeometry>(rd.number(e, "exponent", "1"), rd.number(e, "index", "1"),
                                                       rd.quantity(e, "openingAngle", "posangle"), rd.quantity(e, "minRadius", "length"),
                                                       rd.quantity(e, "maxRadius", "length"), rd.boolean(e, "reshapeInnerRadius", false),
                                                       rd.quantity(e, "cutoffRadius", "length", "0"));
            if (e.name == "RingGeometry")
                return std::make_unique<RingGeometry>(rd.quantity(e, "ringRadius", "length"), rd.quantity(e, "width", "length"),
                                                      rd.quantity(e, "height", "length"));
            unsupported("geometry " + e.name);
        }

        std::unique_ptr<WavelengthGrid> makeWavelengthGrid(const XmlElement& e, const Reader& rd)
        {
            auto grid = std::make_unique<WavelengthGrid>();
            if (e.name == "LogWavelengthGrid")
            {
                // LogWavelengthGrid.cpp:12-24
                double lo = rd.quantity(e, "minWavelength", "wavelength");
                double hi = rd.quantity(e, "maxWavelength", "wavelength");
                int n = rd.integer(e, "numWavelengths", 25);
                if (hi <= lo) throw std::runtime_error("the longest wavelength should be larger than the shortest");
                Array lambdav;
                tab::logGrid(lambdav, lo, hi, n - 1);
                grid->setWavelengthRange(lambdav, true);
            }
            else if (e.name == "LinWavelengthGrid")
            {
                double lo = rd.quantity(e, "minWavelength", "wavelength");
                double hi = rd.quantity(e, "maxWavelength", "wavelength");
                int n = rd.integer(e, "numWavelengths", 25);
                if (hi <= lo) throw std::runtime_error("the longest wavelength should be larger than the shortest");
                Array lambdav;
                tab::linearGrid(lambdav, lo, hi, n - 1);
                grid->setWavelengthRange(lambdav, false);
            }
            else if (e.name == "ListWavelengthGrid")
            {
                Array lambdav = rd.list(e, "wavelengths", "wavelength", "");
                double rhw = rd.number(e, "relativeHalfWidth", "0");
                if (rhw)
                    grid->setWavelengthBins(lambdav, rhw, false);
                else
                    grid->setWavelengthRange(lambdav, rd.boolean(e, "log", true));
            }
            else
                unsupported("wavelength grid " + e.name);
            return grid;
        }
    }

    // ================================================================ construction

    std::unique_ptr<Simulation> Simulation::fromFile(const std::string& path)
    {
        std::ifstream in(path, std::ios::binary);
        if (!in) throw std::runtime_error("Cannot open ski file " + path);
        std::stringstream ss;
        ss << in.rdbuf();
        // output prefix = file name without directory and extension (StringUtils::filenameBase)
        std::string base = path;
        size_t slash = base.find_last_of('/');
        if (slash != std::string::npos) base = base.substr(slash + 1);
        size_t dot = base.find_last_of('.');
        if (dot != std::string::npos) base = base.substr(0, dot);
        // input files named in the ski are looked up in the ski file's directory (the reference: FilePaths::input, i.e.
        // the -i directory or the working directory; SKH_INPUT_PATH overrides)
        std::string dir = slash != std::string::npos ? path.substr(0, slash) : std::string(".");
        if (const char* env = getenv("SKH_INPUT_PATH")) dir = env;
        return fromString(ss.str(), base, dir);
    }

    std::unique_ptr<Simulation> Simulation::fromString(const std::string& text, const std::string& prefix, const std::string& inputPath)
    {
        XmlParser parser(text);
        auto root = parser.parseDocument();
        std::unique_ptr<Simulation> sim(new Simulation());
        sim->_prefix = prefix;
        sim->_inputPath = inputPath;
        sim->parse(*root);
        return sim;
    }

    void Simulation::parse(const XmlElement& root)
    {
        if (root.name != "skirt-simulation-hierarchy" || root.attr("type", "") != "MonteCarloSimulation")
            throw std::runtime_error("ski: the root element is not a skirt-simulation-hierarchy of type MonteCarloSimulation");
        if (root.children.empty() || root.children[0]->name != "MonteCarloSimulation")
            throw std::runtime_error("ski: missing MonteCarloSimulation element");
        const XmlElement& sim = *root.children[0];

        // units first: they determine the default unit of unit-less attribute values
        Reader rd;
        rd.unitSystem = "ExtragalacticUnits";
        if (const XmlElement* u = sim.item("units"))
        {
            if (!unitTable().hasSystem(u->name)) unsupported("unit system " + u->name);
            _units.system = rd.unitSystem = u->name;
            _units.wavelengthStyle = u->attr("wavelengthOutputStyle", "Wavelength");
            _units.fluxStyle = u->attr("fluxOutputStyle", "Frequency");
            if (_units.wavelengthStyle != "Wavelength") unsupported("wavelengthOutputStyle " + _units.wavelengthStyle);
            if (_units.fluxStyle != "Frequency" && _units.fluxStyle != "Wavelength" && _units.fluxStyle != "Neutral")
                unsupported("fluxOutputStyle " + _units.fluxStyle);
        }

        std::string mode = sim.attr("simulationMode", "ExtinctionOnly");
        // (the modes without a medium: Configuration::hasMedium() is false, the medium system of the ski file -- if any -- is not
        // relevant, and the photon life cycle ends with the emission peel-off, MonteCarloSimulation.cpp:557)
        if (mode == "OligoExtinctionOnly")
            _oligo = true;
        else if (mode == "ExtinctionOnly")
            _oligo = false;
        else if (mode == "OligoNoMedium")
            _oligo = true, _hasMedium = false;
        else if (mode == "NoMedium")
            _oligo = false, _hasMedium = false;
        else
            unsupported("simulationMode " + mode);
        if (rd.boolean(sim, "iteratePrimaryEmission", false)) unsupported("iteratePrimaryEmission");
        _numPackets = static_cast<uint64_t>(rd.number(sim, "numPackets", "1e6"));  // Configuration.cpp:50

        if (const XmlElement* r = sim.item("random"))
        {
            if (r->name != "Random") unsupported("random generator " + r->name);
            _seed = rd.integer(*r, "seed", 0);
        }
        if (const XmlElement* c = sim.item("cosmology"))
        {
            if (c->name == "FlatUniverseCosmology")
            {
                // the model sits at redshift z in a flat universe (FlatUniverseCosmology.cpp:18-53): comoving distance
                // c/H0 * integral of dz' / sqrt(Om (1+z')^3 + 1 - Om) by the midpoint rule on max(2000, 10000 z) intervals
                const double z = rd.number(*c, "redshift", "1");
                const double h = rd.number(*c, "reducedHubbleConstant", "0.675");
                const double Om = rd.number(*c, "matterDensityFraction", "0.310");
                if (!(z > 0.)) unsupported("FlatUniverseCosmology with redshift zero");
                const double front = 10. * constants::pc * constants::c / h;
                const int n = std::max(2000, static_cast<int>(z * 10000));
                const double dz = z / n;
                double sum = 0.;
                for (int i = 0; i != n; ++i)
                {
                    const double zp1 = (i + 0.5) * dz + 1.;
                    sum += 1. / sqrt(Om * zp1 * zp1 * zp1 + (1. - Om));
                }
                const double comoving = front * z * sum / n;
                _modelRedshift = z;
                _cosmoAngularDiameterDistance = comoving / (1. + z);
                _cosmoLuminosityDistance = comoving * (1. + z);
            }
            else if (c->name != "LocalUniverseCosmology")
                unsupported("cosmology " + c->name);
        }

        // ---- source system (SourceSystem.hpp properties)
        const XmlElement* ss = sim.item("sourceSystem");
        if (!ss) throw std::runtime_error("ski: missing sourceSystem");
        _ssMinWavelength = rd.quantity(*ss, "minWavelength", "wavelength", "0.09 micron");
        _ssMaxWavelength = rd.quantity(*ss, "maxWavelength", "wavelength", "100 micron");
        _oligoWavelengths = rd.list(*ss, "wavelengths", "wavelength", "0.55 micron");
        _sourceBias = rd.number(*ss, "sourceBias", "0.5");
        auto sources = ss->items("sources");
        if (sources.empty() || sources.size() > 16) unsupported("a source system with " + std::to_string(sources.size()) + " sources");
        for (const XmlElement* srcElement : sources)
        {
        const XmlElement& src = *srcElement;
        _sources.emplace_back();
        SourceModel& _source = _sources.back();
        _source.type = src.name;
        _source.sourceWeight = rd.number(src, "sourceWeight", "1");
        _source.wavelengthBias = rd.number(src, "wavelengthBias", "0.5");
        if (src.name == "PointSource")
        {
            _source.position = Vec3{rd.quantity(src, "positionX", "length", "0"), rd.quantity(src, "positionY", "length", "0"),
                                    rd.quantity(src, "positionZ", "length", "0")};
            bool moving = rd.quantity(src, "velocityX", "velocity", "0") || rd.quantity(src, "velocityY", "velocity", "0")
                          || rd.quantity(src, "velocityZ", "velocity", "0");
            if (moving && !_oligo) unsupported("a source with a bulk velocity");
            if (const XmlElement* ad = src.item("angularDistribution"))
            {
                if (ad->name == "IsotropicAngularDistribution")
                    _source.angularKind = PMC_ANGULAR_ISOTROPIC;
                else if (ad->name == "LaserAngularDistribution" || ad->name == "ConicalAngularDistribution" || ad->name == "NetzerAngularDistribution")
                {
                    _source.angularKind = ad->name == "LaserAngularDistribution"     ? PMC_ANGULAR_LASER
                                          : ad->name == "ConicalAngularDistribution" ? PMC_ANGULAR_CONICAL
                                                                                     : PMC_ANGULAR_NETZER;
                    // AxAngularDistribution.cpp:12-18, Direction.cpp:61-79: the symmetry axis, normalised
                    double x = rd.number(*ad, "symmetryX", "0"), y = rd.number(*ad, "symmetryY", "0"), z = rd.number(*ad, "symmetryZ", "1");
                    const double norm = std::sqrt(x * x + y * y + z * z);
                    if (!(norm > 0.)) throw std::runtime_error("Symmetry axis direction cannot be null vector");
                    x /= norm, y /= norm, z /= norm;
                    _source.angularAxis = Vec3{x, y, z};
                    if (ad->name == "ConicalAngularDistribution")
                        _source.angularCosDelta = std::cos(rd.quantity(*ad, "openingAngle", "posangle", ""));
                }
                else
                    unsupported("angular distribution " + ad->name);
            }
            if (const XmlElement* pp = src.item("polarizationProfile"))
                if (pp->name != "NoPolarizationProfile") unsupported("polarization profile " + pp->name);
        }
        else if (src.name == "GeometricSource")
        {
            const XmlElement* g = src.item("geometry");
            if (!g) throw std::runtime_error("ski: GeometricSource lacks a geometry");
            if (g->name != "SersicGeometry" && g->name != "UniformBoxGeometry" && g->name != "ExpDiskGeometry" && g->name != "PlummerGeometry"
                && g->name != "SpheroidalGeometryDecorator" && g->name != "OffsetGeometryDecorator")
                unsupported("source geometry " + g->name);
            _source.geometry = makeGeometry(*g, rd);
            const Geometry* undecorated = _source.geometry.get();
            if (auto off = dynamic_cast<const OffsetGeometry*>(undecorated)) undecorated = off->inner();
            const std::string ut = undecorated->type();
            if (ut != "SersicGeometry" && ut != "UniformBoxGeometry" && ut != "ExpDiskGeometry" && ut != "PlummerGeometry"
                && ut != "SpheroidalGeometryDecorator")
                unsupported("source geometry " + ut);
            if (auto sph = dynamic_cast<const SpheroidalGeometry*>(undecorated))
                if (sph->inner()->type() != "SersicGeometry" && sph->inner()->type() != "PlummerGeometry")
                    unsupported("source geometry SpheroidalGeometryDecorator of " + sph->inner()->type());
            if (src.item("velocityDistribution") && rd.quantity(src, "velocityMagnitude", "velocity", "0") && !_oligo)
                unsupported("a source with a velocity field");
        }
        else
            unsupported("source " + src.name);
        if (const XmlElement* sed = src.item("sed"))
        {
            _source.sedType = sed->name;
            if (sed->name == "BlackBodySED")
                _source.temperature = rd.quantity(*sed, "temperature", "temperature", "5000 K");
            else if (sed->name == "ListSED")
            {
                // ListSED.cpp:11-18 with the default unit style (per unit of wavelength)
                if (sed->attr("unitStyle", "wavelengthmonluminosity") != "wavelengthmonluminosity")
                    unsupported("ListSED unitStyle " + sed->attr("unitStyle", ""));
                _source.sedInLambda = rd.list(*sed, "wavelengths", "wavelength", "");
                _source.sedInP = rd.list(*sed, "specificLuminosities", "wavelengthmonluminosity", "");
                if (_source.sedInLambda.size() != _source.sedInP.size())
                    throw std::runtime_error("Number of listed luminosities does not match number of listed wavelengths");
            }
            else if (sed->name == "FileSED")
            {
                // FileSED.cpp:11-18
                std::string filename = sed->attr("filename", "");
                if (filename.empty()) throw std::runtime_error("ski: FileSED lacks a filename");
                std::string path = (filename[0] == '/') ? filename : _inputPath + "/" + filename;
                for (const Array& row : readColumnFile(path, {{"wavelength", "wavelength", "micron"}, {"specific luminosity", "specific", "W/m"}},
                                                       "spectral energy distribution"))
                {
                    _source.sedInLambda.push_back(row[0]);
                    _source.sedInP.push_back(row[1]);
                }
            }
            else
                unsupported("SED " + sed->name);
            if (sed->name != "BlackBodySED")
            {
                // TabulatedSED::setupSelfBefore (TabulatedSED.cpp:14-21)
                if (_source.sedInLambda.size() < 2) throw std::runtime_error("SED must have at least two wavelength/luminosity pairs");
                if (_source.sedInLambda.front() > _source.sedInLambda.back())
                {
                    std::reverse(_source.sedInLambda.begin(), _source.sedInLambda.end());
                    std::reverse(_source.sedInP.begin(), _source.sedInP.end());
                }
            }
        }
        if (const XmlElement* norm = src.item("normalization"))
        {
            _source.normType = norm->name;
            if (norm->name == "SpecificLuminosityNormalization")
            {
                // SpecificLuminosityNormalization.cpp:12-21 with Units::fromFluxStyle (Units.cpp:41-51)
                std::string style = norm->attr("unitStyle", "wavelengthmonluminosity");
                if (style != "wavelengthmonluminosity" && style != "neutralmonluminosity" && style != "frequencymonluminosity")
                    unsupported("SpecificLuminosityNormalization unitStyle " + style);
                _source.normWavelength = rd.quantity(*norm, "wavelength", "wavelength");
                double L = rd.quantity(*norm, "specificLuminosity", style.c_str());
                const double lambda = _source.normWavelength;
                if (style == "neutralmonluminosity") L = L / lambda;
                if (style == "frequencymonluminosity") L = L * constants::c / lambda / lambda;
                _source.specificLuminosity = L;
            }
            else if (norm->name != "IntegratedLuminosityNormalization")
                unsupported("luminosity normalization " + norm->name);
            _source.normRange = norm->attr("wavelengthRange", "Source");
            _source.normMinWavelength = rd.quantity(*norm, "minWavelength", "wavelength", "0.09 micron");
            _source.normMaxWavelength = rd.quantity(*norm, "maxWavelength", "wavelength", "100 micron");
            if (norm->name == "IntegratedLuminosityNormalization")
                _source.integratedLuminosity = rd.quantity(*norm, "integratedLuminosity", "bolluminosity");
        }
        else
            throw std::runtime_error("ski: source lacks a luminosity normalization");
        if (const XmlElement* bd = src.item("wavelengthBiasDistribution"))
        {
            _source.biasDistType = bd->name;
            if (bd->name == "LogWavelengthDistribution" || bd->name == "LinWavelengthDistribution")
            {
                _source.biasMin = rd.quantity(*bd, "minWavelength", "wavelength", "1 pm");
                _source.biasMax = rd.quantity(*bd, "maxWavelength", "wavelength", "1 m");
            }
            else if (bd->name != "DefaultWavelengthDistribution")
                unsupported("wavelength bias distribution " + bd->name);
        }
        }  // sources

        // ---- medium system
        const XmlElement* ms = _hasMedium ? sim.item("mediumSystem") : nullptr;
        if (_hasMedium && !ms) throw std::runtime_error("ski: simulationMode " + mode + " needs a medium system");
        if (!ms)
        {
            // A simulation without a medium (MonteCarloSimulation.cpp:557; FluxRecorder.cpp:199: the total flux only).  The engine's
            // life cycle is that of a medium system whose one cell holds no matter: the emission peel-off sees optical depth zero, the
            // path of the packet has optical depth zero, and the history ends (MonteCarloSimulation.cpp:705-709).  One density sample
            // at the cell centre: the setup draws no random number, as the reference's setup without a medium draws none.
            static const char* const standIn =
                "<mediumSystem type=\"MediumSystem\"><MediumSystem>"
                "<photonPacketOptions type=\"PhotonPacketOptions\"><PhotonPacketOptions explicitAbsorption=\"false\" forceScattering=\"true\" "
                "minWeightReduction=\"1e4\" minScattEvents=\"0\" pathLengthBias=\"0.5\"/></photonPacketOptions>"
                "<media type=\"Medium\"><GeometricMedium>"
                "<geometry type=\"Geometry\"><UniformBoxGeometry minX=\"-1 m\" maxX=\"1 m\" minY=\"-1 m\" maxY=\"1 m\" minZ=\"-1 m\" maxZ=\"1 m\"/></geometry>"
                "<materialMix type=\"MaterialMix\"><MeanListDustMix wavelengths=\"1e-6 micron, 1e6 micron\" extinctionCoefficients=\"1 m2/kg, 1 m2/kg\" "
                "albedos=\"0.5, 0.5\" asymmetryParameters=\"0, 0\"/></materialMix>"
                "<normalization type=\"MaterialNormalization\"><NumberMaterialNormalization number=\"0\"/></normalization>"
                "</GeometricMedium></media>"
                "<samplingOptions type=\"SamplingOptions\"><SamplingOptions numDensitySamples=\"1\"/></samplingOptions>"
                "<grid type=\"SpatialGrid\"><CartesianSpatialGrid minX=\"-1 m\" maxX=\"1 m\" minY=\"-1 m\" maxY=\"1 m\" minZ=\"-1 m\" maxZ=\"1 m\">"
                "<meshX type=\"Mesh\"><LinMesh numBins=\"1\"/></meshX><meshY type=\"Mesh\"><LinMesh numBins=\"1\"/></meshY>"
                "<meshZ type=\"Mesh\"><LinMesh numBins=\"1\"/></meshZ></CartesianSpatialGrid></grid>"
                "</MediumSystem></mediumSystem>";
            _standInMediumSystem = XmlParser(standIn).parseDocument();
            ms = _standInMediumSystem->children.front().get();
        }
        _options.force_scattering = 1;
        _options.min_weight_reduction = 1e4;
        _options.min_scatt_events = 0;
        _options.path_length_bias = 0.5;
        _options.explicit_absorption = 0;
        if (const XmlElement* po = ms->item("photonPacketOptions"))
        {
            _options.explicit_absorption = rd.boolean(*po, "explicitAbsorption", false) ? 1 : 0;
            _options.force_scattering = rd.boolean(*po, "forceScattering", true);
            _options.min_weight_reduction = rd.number(*po, "minWeightReduction", "1e4");
            _options.min_scatt_events = rd.integer(*po, "minScattEvents", 0);
            _options.path_length_bias = rd.number(*po, "pathLengthBias", _options.force_scattering ? "0.5" : "0");
        }
        if (const XmlElement* rf = ms->item("radiationFieldOptions"))
            if (rd.boolean(*rf, "storeRadiationField", false))
            {
                // Configuration.cpp:244-250,476-482: the field is stored on the oligochromatic grid, or on the configured
                // radiationFieldWLG; storing it requires forced scattering (the reference switches it on with a warning)
                _storeRadiationField = true;
                if (!_options.force_scattering) unsupported("storeRadiationField without forceScattering");
                if (!_oligo)
                {
                    const XmlElement* wg = rf->item("radiationFieldWLG");
                    if (!wg) throw std::runtime_error("ski: RadiationFieldOptions lacks a radiationFieldWLG");
                    _rfGridOwn = makeWavelengthGrid(*wg, rd);
                }
            }
        if (const XmlElement* so = ms->item("samplingOptions")) _numDensitySamples = rd.integer(*so, "numDensitySamples", 100);
        auto media = ms->items("media");
        if (media.empty()) unsupported("a medium system without media");
        if (media.size() > PMC_MAX_MEDIA) unsupported("a medium system with more than " + std::to_string(PMC_MAX_MEDIA) + " media");
        // (several components: every one a dust medium with spatially constant cross sections --
        // Configuration::hasMultipleConstantSectionMedia, MediumSystem.cpp:874-887)
        for (const XmlElement* mediumElement : media)
        {
        const XmlElement& med = *mediumElement;
        std::unique_ptr<Medium> one;
        if (med.name != "GeometricMedium" && med.name != "ParticleMedium") unsupported("medium " + med.name);
        if (med.item("velocityDistribution") && rd.quantity(med, "velocityMagnitude", "velocity", "0") && !_oligo)
            unsupported("a medium with a velocity field");
        if (med.item("magneticFieldDistribution") && rd.quantity(med, "magneticFieldStrength", "magneticfield", "0"))
            unsupported("a medium with a magnetic field");
        const XmlElement* mm = med.item("materialMix");
        if (!mm) throw std::runtime_error("ski: " + med.name + " lacks a material mix");
        if (mm->name != "MeanListDustMix" && mm->name != "MeanFileDustMix") unsupported("material mix " + mm->name);
        auto mix = std::make_unique<DustMix>();
        mix->typeName = mm->name;
        if (mm->name == "MeanListDustMix")
        {
            mix->inLambda = rd.list(*mm, "wavelengths", "wavelength", "");
            mix->inKappaExt = rd.list(*mm, "extinctionCoefficients", "masscoefficient", "");
            mix->inAlbedo = rd.list(*mm, "albedos", "", "");
            mix->inAsymmpar = rd.list(*mm, "asymmetryParameters", "", "");
        }
        else
        {
            // MeanFileDustMix.cpp:11-22: four columns of a text file (wavelength, kappa_ext, albedo, g)
            std::string filename = mm->attr("filename", "");
            if (filename.empty()) throw std::runtime_error("ski: MeanFileDustMix lacks a filename");
            std::string path = (filename[0] == '/') ? filename : _inputPath + "/" + filename;
            auto rows = readColumnFile(path, {{"wavelength", "wavelength", "micron"},
                                              {"extinction mass coefficient", "masscoefficient", "m2/kg"},
                                              {"scattering albedo", "", ""},
                                              {"scattering asymmetry parameter", "", ""}},
                                       "optical dust properties");
            for (const Array& row : rows)
            {
                mix->inLambda.push_back(row[0]);
                mix->inKappaExt.push_back(row[1]);
                mix->inAlbedo.push_back(row[2]);
                mix->inAsymmpar.push_back(row[3]);
            }
        }
        if (med.name == "GeometricMedium")
        {
            auto gm = std::make_unique<GeometricMedium>();
            const XmlElement* mg = med.item("geometry");
            if (!mg) throw std::runtime_error("ski: GeometricMedium lacks a geometry");
            gm->geometry = makeGeometry(*mg, rd);
            const XmlElement* mn = med.item("normalization");
            if (!mn) throw std::runtime_error("ski: GeometricMedium lacks a normalization");
            gm->normType = mn->name;
            if (mn->name == "OpticalDepthMaterialNormalization")
            {
                std::string axis = mn->attr("axis", "Z");
                gm->normAxis = axis.empty() ? 'Z' : axis[0];
                gm->normWavelength = rd.quantity(*mn, "wavelength", "wavelength");
                gm->normOpticalDepth = rd.number(*mn, "opticalDepth");
            }
            else if (mn->name == "MassMaterialNormalization")
                gm->normMass = rd.quantity(*mn, "mass", "mass");
            else if (mn->name == "NumberMaterialNormalization")
                gm->normNumber = rd.number(*mn, "number");
            else
                unsupported("material normalization " + mn->name);
            one = std::move(gm);
        }
        else
        {
            // ParticleMedium (ParticleMedium.hpp, ImportedMedium.hpp): a text column file of smoothed particles
            auto pm = std::make_unique<ParticleMedium>();
            std::string filename = med.attr("filename", "");
            if (filename.empty()) throw std::runtime_error("ski: ParticleMedium lacks a filename");
            pm->options.path = (filename[0] == '/') ? filename : _inputPath + "/" + filename;
            std::string massType = med.attr("massType", "Mass");
            if (massType != "Mass" && massType != "Number") unsupported("massType " + massType);
            pm->options.holdsNumber = massType == "Number";
            pm->options.massFraction = rd.number(med, "massFraction", "1");
            pm->options.importMetallicity = rd.boolean(med, "importMetallicity", false);
            pm->options.importTemperature = rd.boolean(med, "importTemperature", false);
            pm->options.maxTemperature = rd.quantity(med, "maxTemperature", "temperature", "0 K");
            if (rd.boolean(med, "importVelocity", false) && !_oligo) unsupported("an imported medium with a velocity field");
            if (rd.boolean(med, "importMagneticField", false)) unsupported("an imported medium with a magnetic field");
            if (rd.boolean(med, "importVariableMixParams", false)) unsupported("an imported medium with a variable material mix");
            if (!med.attr("useColumns", "").empty()) unsupported("useColumns (column remapping)");
            if (const XmlElement* sk = med.item("smoothingKernel")) pm->kernelType = sk->name;
            one = std::move(pm);
        }
        one->mix = std::move(mix);
        _media.push_back(std::move(one));
        }  // media
        for (auto& part : _media) _composite.parts.push_back(part.get());
        _medium = _media.size() == 1 ? _media[0].get() : &_composite;

        const XmlElement* ge = ms->item("grid");
        if (!ge) throw std::runtime_error("ski: MediumSystem lacks a spatial grid");
        Box extent(rd.quantity(*ge, "minX", "length"), rd.quantity(*ge, "minY", "length"), rd.quantity(*ge, "minZ", "length"),
                   rd.quantity(*ge, "maxX", "length"), rd.quantity(*ge, "maxY", "length"), rd.quantity(*ge, "maxZ", "length"));
        if (ge->name == "CartesianSpatialGrid")
        {
            auto grid = std::make_unique<CartesianSpatialGrid>();
            grid->extent = extent;
            int axisIndex = 0;
            auto bins = [&](const char* prop) {
                CartesianSpatialGrid::MeshSpec& spec = grid->meshSpec[axisIndex++];
                const XmlElement* mesh = ge->item(prop);
                if (!mesh) return 100;  // Mesh default numBins
                if (mesh->name == "ListMesh")
                {
                    // TabulatedMesh::setupSelfBefore (TabulatedMesh.cpp:12-33) on ListMesh::getMeshBorderPoints
                    std::vector<double> points;
                    std::string text = mesh->attr("points", "");
                    for (char& ch : text)
                        if (ch == ',') ch = ' ';
                    std::istringstream in(text);
                    for (double v; in >> v;) points.push_back(v);
                    std::sort(points.begin(), points.end());
                    points.erase(std::unique(points.begin(), points.end()), points.end());
                    if (points.size() < 1) throw std::runtime_error("The mesh data file has no points");
                    if (points.front() < 0.) throw std::runtime_error("The mesh data file has negative points");
                    if (points.front() != 0.) points.insert(points.begin(), 0.);
                    if (points.size() < 2 || points.back() == 0.) throw std::runtime_error("The mesh data file has no positive points");
                    const double last = points.back();
                    for (double& v : points) v /= last;
                    spec.type = mesh->name;
                    spec.points = points;
                    return static_cast<int>(points.size()) - 1;
                }
                if (mesh->name != "LinMesh" && mesh->name != "PowMesh" && mesh->name != "SymPowMesh" && mesh->name != "LogMesh"
                    && mesh->name != "SymLogMesh")
                    unsupported("mesh " + mesh->name);
                spec.type = mesh->name;
                spec.ratio = rd.number(*mesh, "ratio", "1");
                spec.centralBinFraction = rd.number(*mesh, "centralBinFraction", "1e-3");
                return rd.integer(*mesh, "numBins", 100);
            };
            grid->nx = bins("meshX");
            grid->ny = bins("meshY");
            grid->nz = bins("meshZ");
            // (the engine stages the three border arrays in LDS: 160 KB hold about 20 000 borders; the cell index is an int32)
            if (grid->nx + grid->ny + grid->nz > 20000 || double(grid->nx) * grid->ny * grid->nz >= 2147483648.)
                unsupported("a Cartesian grid with more than 20000 bins on its three axes together, or 2^31 cells");
            _grid = std::move(grid);
        }
        else if (ge->name == "PolicyTreeSpatialGrid")
        {
            if (ge->attr("treeType", "OctTree") != "OctTree") unsupported("treeType " + ge->attr("treeType"));
            auto grid = std::make_unique<OctreeSpatialGrid>();
            grid->extent = extent;
            if (const XmlElement* pol = ge->item("policy"))
            {
                if (pol->name != "DensityTreePolicy") unsupported("tree policy " + pol->name);
                grid->minLevel = rd.integer(*pol, "minLevel", 3);
                grid->maxLevel = rd.integer(*pol, "maxLevel", 7);
                grid->maxDustFraction = rd.number(*pol, "maxDustFraction", "1e-6");
                grid->maxDustOpticalDepth = rd.number(*pol, "maxDustOpticalDepth", "0");
                grid->policyWavelength = rd.quantity(*pol, "wavelength", "wavelength", "0.55 micron");
                grid->maxDustDensityDispersion = rd.number(*pol, "maxDustDensityDispersion", "0");
            }
            _grid = std::move(grid);
        }
        else if (ge->name == "VoronoiMeshSpatialGrid")
        {
            // VoronoiMeshSpatialGrid.hpp: policies Uniform (random sites), CentralPeak, DustDensity, File (sites from a column text file),
            // ImportedSites (the positions of an imported medium's entities)
            auto grid = std::make_unique<VoronoiSpatialGrid>();
            grid->extent = extent;
            grid->policy = ge->attr("policy", "DustDensity");
            if (grid->policy != "Uniform" && grid->policy != "File" && grid->policy != "DustDensity" && grid->policy != "CentralPeak"
                && grid->policy != "ImportedSites")
                unsupported("Voronoi site policy " + grid->policy);
            grid->numSites = rd.integer(*ge, "numSites", 500);
            if (grid->policy == "File")
            {
                std::string filename = ge->attr("filename", "");
                if (filename.empty()) throw std::runtime_error("ski: VoronoiMeshSpatialGrid lacks a filename");
                grid->sitesPath = (filename[0] == '/') ? filename : _inputPath + "/" + filename;
            }
            grid->relaxSites = rd.boolean(*ge, "relaxSites", false);
            _grid = std::move(grid);
        }
        else
            unsupported("spatial grid " + ge->name);

        // ---- instrument system
        const XmlElement* is = sim.item("instrumentSystem");
        if (!is) throw std::runtime_error("ski: missing instrumentSystem");
        if (const XmlElement* dg = is->item("defaultWavelengthGrid")) _defaultGrid = makeWavelengthGrid(*dg, rd);
        for (const XmlElement* ie : is->items("instruments"))
        {
            InstrumentModel ins;
            ins.type = ie->name;
            if (ie->name != "FrameInstrument" && ie->name != "FullInstrument" && ie->name != "SEDInstrument")
                unsupported("instrument " + ie->name);
            // SEDInstrument (SEDInstrument.cpp:11-22, ApertureInstrument.cpp:11-43): the flux density only, optionally within
            // an aperture radius around the line of sight
            if (ie->name == "SEDInstrument") ins.radius = rd.quantity(*ie, "radius", "length", "0");
            ins.name = ie->attr("instrumentName");
            ins.inclination = rd.quantity(*ie, "inclination", "posangle", "0 deg");
            ins.azimuth = rd.quantity(*ie, "azimuth", "posangle", "0 deg");
            ins.roll = rd.quantity(*ie, "roll", "posangle", "0 deg");
            if (ie->name != "SEDInstrument")
            {
                ins.fieldOfViewX = rd.quantity(*ie, "fieldOfViewX", "length");
                ins.fieldOfViewY = rd.quantity(*ie, "fieldOfViewY", "length");
            }
            ins.centerX = rd.quantity(*ie, "centerX", "length", "0");
            ins.centerY = rd.quantity(*ie, "centerY", "length", "0");
            ins.numPixelsX = rd.integer(*ie, "numPixelsX", 250);
            ins.numPixelsY = rd.integer(*ie, "numPixelsY", 250);
            ins.recordComponents = rd.boolean(*ie, "recordComponents", false);
            ins.numScatteringLevels = rd.integer(*ie, "numScatteringLevels", 0);
            ins.recordPolarization = rd.boolean(*ie, "recordPolarization", false);
            ins.recordStatistics = rd.boolean(*ie, "recordStatistics", false);
            if (ins.recordPolarization) unsupported("recordPolarization");
            // DistantInstrument.cpp:24-36: a distance puts the instrument in the model's rest frame; distance zero means the
            // observer frame of a model at redshift z > 0 (distances from the cosmology, wavelengths shifted by 1 + z)
            ins.distance = rd.quantity(*ie, "distance", "distance", "0");
            if (ins.distance > 0.)
                ins.luminosityDistance = ins.angularDiameterDistance = ins.distance;
            else if (_modelRedshift > 0.)
            {
                ins.redshift = _modelRedshift;
                ins.luminosityDistance = _cosmoLuminosityDistance;
                ins.angularDiameterDistance = _cosmoAngularDiameterDistance;
            }
            else
                throw std::runtime_error("Instrument distance and model redshift are both zero");
            if (const XmlElement* wg = ie->item("wavelengthGrid")) ins.ownGrid = makeWavelengthGrid(*wg, rd);
            _instruments.push_back(std::move(ins));
        }
        if (_instruments.empty()) unsupported("a simulation without instruments");

        // ---- probes: the radiation field per cell (RadiationFieldProbe + PerCellForm); nothing else is on this path
        if (const XmlElement* ps = sim.item("probeSystem"))
            for (const XmlElement* pe : ps->items("probes"))
            {
                if (pe->name != "RadiationFieldProbe") unsupported("probe " + pe->name);
                const XmlElement* form = pe->item("form");
                if (!form || form->name != "PerCellForm") unsupported("probe form " + (form ? form->name : std::string("(none)")));
                if (rd.boolean(*pe, "writeWavelengthGrid", false)) unsupported("RadiationFieldProbe writeWavelengthGrid");
                std::string after = pe->attr("probeAfter", "Run");
                if (after != "Run" && after != "Primary") unsupported("RadiationFieldProbe probeAfter " + after);
                _rfProbeNames.push_back(pe->attr("probeName", ""));
            }
    }

    // ================================================================ setup

    void Simulation::setup()
    {
        _random.setSeed(_seed);

        // ---- Configuration: wavelength regime (Configuration.cpp:58-73)
        double sourceMin, sourceMax;
        if (_oligo)
        {
            _oligoGrid = std::make_unique<WavelengthGrid>();
            _oligoGrid->setWavelengthBins(_oligoWavelengths, 1e-3, true);  // OligoWavelengthGrid.cpp:20-26
            sourceMin = _oligoGrid->rangeMin();
            sourceMax = _oligoGrid->rangeMax();
        }
        else
        {
            sourceMin = _ssMinWavelength;
            sourceMax = _ssMaxWavelength;
        }
        const WavelengthGrid* defaultGrid = _oligo ? _oligoGrid.get() : _defaultGrid.get();
        if (_storeRadiationField) _rfGrid = _oligo ? _oligoGrid.get() : _rfGridOwn.get();


        // instruments: grid in effect (Configuration::wavelengthGrid, Configuration.cpp:666-671) and observer sharing
        // (DistantInstrument::determineSameObserverAsPreceding, DistantInstrument.cpp:55-63)
        for (size_t i = 0; i < _instruments.size(); ++i)
        {
            InstrumentModel& ins = _instruments[i];
            ins.grid = (ins.ownGrid && !_oligo) ? ins.ownGrid.get() : defaultGrid;
            if (!ins.grid) throw std::runtime_error("Cannot find a wavelength grid for instrument or probe");
            if (i > 0)
            {
                const InstrumentModel& o = _instruments[i - 1];
                ins.sameObserverAsPreceding = ins.distance == o.distance && ins.inclination == o.inclination
                                              && ins.azimuth == o.azimuth && ins.roll == o.roll;
            }
        }

        // ---- Configuration::simulationWavelengthRange / simulationWavelengths (Configuration.cpp:566-661)
        double rangeMin = sourceMin, rangeMax = sourceMax;
        auto extend = [&](double lo, double hi) {
            if (lo < rangeMin) rangeMin = lo;
            if (hi > rangeMax) rangeMax = hi;
        };
        std::set<double> simWavelengths;
        auto addGrid = [&](const WavelengthGrid* g) {
            extend(g->rangeMin(), g->rangeMax());
            for (double w : g->lambdav) simWavelengths.insert(w);
        };
        if (defaultGrid) addGrid(defaultGrid);
        if (_rfGrid && !_oligo) addGrid(_rfGrid);  // Configuration.cpp:576-579,644
        for (auto& ins : _instruments)
            if (ins.ownGrid) addGrid(ins.ownGrid.get());
        // MaterialWavelengthRangeInterface items: the normalisation wavelength and the tree policy wavelength
        for (auto& part : _media)
            if (part->normalizationWavelength() > 0)
            {
                extend(part->normalizationWavelength(), part->normalizationWavelength());
                simWavelengths.insert(part->normalizationWavelength());
            }
        if (auto tree = dynamic_cast<OctreeSpatialGrid*>(_grid.get()))
            if (tree->maxDustOpticalDepth > 0 && tree->policyWavelength > 0)
            {
                extend(tree->policyWavelength, tree->policyWavelength);
                simWavelengths.insert(tree->policyWavelength);
            }
        rangeMin /= (1. + 1. / 100.);  // Range::extendWithRedshift
        rangeMax *= (1. + 1. / 100.);

        // ---- dust mix, medium normalisation
        for (auto& part : _media)
        {
            part->mix->setup(rangeMin, rangeMax, std::vector<double>(simWavelengths.begin(), simWavelengths.end()));
            if (auto pm = dynamic_cast<ParticleMedium*>(part.get())) pm->snapshot.useDeviceSampler(_samplerApi);
            part->setup();
        }

        // ---- spatial grid (tree construction draws from the random stream) then cell densities
        if (auto cart = dynamic_cast<CartesianSpatialGrid*>(_grid.get()))
            cart->setup();
        else if (auto tree = dynamic_cast<OctreeSpatialGrid*>(_grid.get()))
        {
            if (!_topology.empty())
                tree->setupFromTopology(_topology);
            else
                tree->setup(*_medium, _numDensitySamples, _random);
        }
        else if (auto voro = dynamic_cast<VoronoiSpatialGrid*>(_grid.get()))
            voro->setup(_random, *_medium);

        // MediumSystem::setupSelfAfter density sampling (MediumSystem.cpp:80-106,308-321)
        // (the sample positions of a cell are drawn once and serve every component: PropertySampler::prepareForCell, :80-96)
        int numCells = _grid->numCells();
        const size_t H = _media.size();
        _density.assign(H, Array(numCells, 0.));
        if (_numDensitySamples == 1)
        {
            for (size_t h = 0; h != H; ++h)
                for (int m = 0; m != numCells; ++m) _density[h][m] = _media[h]->numberDensity(_grid->centralPositionInCell(m));
        }
        else
        {
            // positions from the random stream in cell order, densities on all host cores, sums in sample order
            const int batchCells = std::max(1, (1 << 22) / _numDensitySamples);
            std::vector<Vec3> pos;
            std::vector<double> samples;
            for (int m0 = 0; m0 < numCells; m0 += batchCells)
            {
                const int m1 = std::min(numCells, m0 + batchCells);
                pos.clear();
                for (int m = m0; m != m1; ++m)
                {
                    for (int n = 0; n != _numDensitySamples; ++n) pos.push_back(_grid->randomPositionInCell(m, _random));
                }
                for (size_t h = 0; h != H; ++h)
                {
                    _media[h]->numberDensities(pos, samples);
                    size_t at = 0;
                    for (int m = m0; m != m1; ++m)
                    {
                        double sum = 0.;
                        for (int n = 0; n != _numDensitySamples; ++n) sum += samples[at++];
                        _density[h][m] = sum / _numDensitySamples;
                    }
                }
            }
        }

        // ---- sources: luminosity and wavelength sampling tables
        buildScene();
        const int Ns = static_cast<int>(_sources.size());
        _sourceTables.assign(Ns, SourceTables());
        _sceneSources.assign(Ns, pmc_source());
        for (int hs = 0; hs < Ns; ++hs)
        {
        SourceModel& _source = _sources[hs];
        Array& _sedLambda = _sourceTables[hs].sedLambda;
        Array& _sedp = _sourceTables[hs].sedp;
        Array& _sedP = _sourceTables[hs].sedP;
        Array& _oligoWeight = _sourceTables[hs].oligoWeight;
        double& _sourceLuminosity = _sourceTables[hs].luminosity;
        pmc_source& flat = _sceneSources[hs];
        flattenSourceGeometry(_source, flat);
        // BlackBodySED (BlackBodySED.cpp:12-39) over the normalisation range = source range
        const double h = constants::h, c = constants::c, k = constants::k;
        double f1 = h * c / (k * _source.temperature);
        double f2 = 2.0 * h * c * c;
        auto planck = [&](double lambda) { return f2 / pow(lambda, 5) / (exp(f1 / lambda) - 1.0); };
        const bool tabulated = _source.sedType != "BlackBodySED";
        // NR::cdf<NR::interpolateLogLog>(xv, pv, Pv, inxv, inpv, range) (NR.hpp:494-520): the tabulated function restricted to
        // a range, its end points interpolated, and the normalised cumulative distribution
        auto tableCdf = [&](Array& xv, Array& pv, Array& Pv, const Array& inxv, const Array& inpv, double lo, double hi) {
            size_t minRight = std::upper_bound(inxv.begin(), inxv.end(), lo) - inxv.begin();
            size_t maxRight = std::lower_bound(inxv.begin(), inxv.end(), hi) - inxv.begin();
            size_t n = 1 + maxRight - minRight;
            xv.assign(n + 1, 0.);
            size_t i = 0;
            xv[i++] = lo;
            for (size_t j = minRight; j < maxRight;) xv[i++] = inxv[j++];
            xv[i++] = hi;
            pv.assign(n + 1, 0.);
            pv[0] = minRight == 0 ? 0. : tab::logLog(xv[0], inxv[minRight - 1], inxv[minRight], inpv[minRight - 1], inpv[minRight]);
            for (size_t q = 1; q < n; ++q) pv[q] = inpv[minRight + q - 1];
            pv[n] = maxRight == inxv.size()
                        ? 0.
                        : tab::logLog(xv[n], inxv[maxRight - 1], inxv[maxRight], inpv[maxRight - 1], inpv[maxRight]);
            return tab::cumulative(true, xv, pv, Pv);
        };
        auto planckCdf = [&](Array& lambdav, Array& pv, Array& Pv, double lo, double hi) {
            if (tabulated) return tableCdf(lambdav, pv, Pv, _source.sedInLambda, _source.sedInP, lo, hi);
            size_t n = std::max(static_cast<size_t>(100), static_cast<size_t>(1000. * log10(hi / lo)));
            tab::logGrid(lambdav, lo, hi, static_cast<int>(n));
            pv.resize(n + 1);
            for (size_t i = 0; i <= n; ++i) pv[i] = planck(lambdav[i]);
            return tab::cumulative(true, lambdav, pv, Pv);
        };
        double Ltot = planckCdf(_sedLambda, _sedp, _sedP, sourceMin, sourceMax);
        if (tabulated)
            for (double& v : _source.sedInP) v /= Ltot;  // TabulatedSED.cpp:20-21 (the later integrals use the normalised table)
        auto specificLuminosity = [&](double lambda) {
            if (!tabulated) return planck(lambda) / Ltot;
            // NR::value<NR::interpolateLogLog> (NR.hpp:372-378)
            int i = tab::bracketOrMiss(_source.sedInLambda, lambda);
            if (i < 0 || lambda < _source.sedInLambda.front()) return 0.;
            return tab::logLog(lambda, _source.sedInLambda[i], _source.sedInLambda[i + 1], _source.sedInP[i], _source.sedInP[i + 1]);
        };

        // IntegratedLuminosityNormalization::luminosityForSED (IntegratedLuminosityNormalization.cpp:12-31)
        if (_source.normType == "SpecificLuminosityNormalization")
        {
            double LlambdaSED = specificLuminosity(_source.normWavelength);
            if (LlambdaSED <= 0) throw std::runtime_error("The normalization wavelength is outside of the SED's wavelength range");
            _sourceLuminosity = _source.specificLuminosity / LlambdaSED;
        }
        else if (_source.normRange == "Source")
            _sourceLuminosity = _source.integratedLuminosity;
        else
        {
            double lo = _source.normRange == "Custom" ? _source.normMinWavelength : 1e-10;
            double hi = _source.normRange == "Custom" ? _source.normMaxWavelength : 1;
            if (lo >= hi) throw std::runtime_error("the normalization wavelength range is empty");
            Array a, b, cc;
            double L = planckCdf(a, b, cc, lo, hi) / (tabulated ? 1. : Ltot);
            if (L <= 0) throw std::runtime_error("the normalization luminosity is zero");
            _sourceLuminosity = _source.integratedLuminosity / L;
        }

        if (_oligo)
        {
            // NormalizedSource::launch with xi = 1 and OligoWavelengthDistribution (NormalizedSource.cpp:26-29,73-110;
            // OligoWavelengthDistribution.cpp:13-41)
            const Array& lam = _oligoGrid->lambdav;
            double probability = 1. / _oligoGrid->numBins() / _oligoGrid->effectiveWidth(0);
            const double xil = 1.;
            _oligoWeight.assign(lam.size(), 0.);
            for (size_t i = 0; i < lam.size(); ++i)
            {
                double s = specificLuminosity(lam[i]);
                if (!s)
                    _oligoWeight[i] = 0.;
                else
                {
                    double b = probability;
                    _oligoWeight[i] = s / ((1 - xil) * s + xil * b);
                }
            }
            flat.lambda_mode = PMC_LAMBDA_OLIGO;
            flat.num_oligo = static_cast<int32_t>(lam.size());
            flat.oligo_lambda = lam.data();
            flat.oligo_weight = _oligoWeight.data();
        }
        else
        {
            flat.lambda_mode = PMC_LAMBDA_TABULATED;
            flat.lambda_bias = _source.wavelengthBias;
            flat.num_sed = static_cast<int32_t>(_sedLambda.size());
            flat.sed_lambda = _sedLambda.data();
            flat.sed_p = _sedp.data();
            flat.sed_P = _sedP.data();
            flat.sed_kind = tabulated ? PMC_SED_TABULATED : PMC_SED_BLACKBODY;
            flat.sed_f1 = f1;
            flat.sed_f2 = f2;
            flat.sed_ltot = Ltot;
            // bias distribution range: Default = source range; Log/Lin = configured range intersected with source range
            double lo = sourceMin, hi = sourceMax;
            if (_source.biasDistType != "DefaultWavelengthDistribution")
            {
                lo = std::max(lo, _source.biasMin);
                hi = std::min(hi, _source.biasMax);
                if (!(lo < hi)) throw std::runtime_error("Wavelength distribution range does not overlap source wavelength range");
            }
            flat.bias_kind = _source.biasDistType == "LinWavelengthDistribution" ? PMC_BIAS_LIN : PMC_BIAS_LOG;
            flat.bias_min = lo;
            flat.bias_max = hi;
        }
        }  // sources

        // SourceSystem::setupSelfAfter / prepareForLaunch / launch (SourceSystem.cpp:14-40,75-107): normalised luminosities
        // _Lv, launch weights _Wv (composite bias), history index boundaries _Iv, luminosity per packet
        {
            double Lsys = 0.;
            for (const SourceTables& t : _sourceTables) Lsys += t.luminosity;
            if (!(Lsys > 0.)) throw std::runtime_error("The total luminosity of the source system is zero");
            std::vector<double> Lv(Ns), wv(Ns), wLv(Ns), Wv(Ns);
            for (int hs = 0; hs < Ns; ++hs) Lv[hs] = _sourceTables[hs].luminosity;
            for (int hs = 0; hs < Ns; ++hs) Lv[hs] /= Lsys;
            for (int hs = 0; hs < Ns; ++hs) wv[hs] = _sources[hs].sourceWeight;
            double wLsum = 0., wsum = 0.;
            for (int hs = 0; hs < Ns; ++hs)
            {
                wLv[hs] = wv[hs] * Lv[hs];
                wLsum += wLv[hs];
                wsum += wv[hs];
            }
            const double xi = _sourceBias;
            for (int hs = 0; hs < Ns; ++hs) Wv[hs] = (1 - xi) * wLv[hs] / wLsum + xi * wv[hs] / wsum;
            _sourceFirst.assign(Ns + 1, 0);
            double W = 0.;
            for (int hs = 1; hs < Ns; ++hs)
            {
                // track the cumulative normalised weight as a floating point number and limit the index to numPackets
                W += Wv[hs - 1];
                _sourceFirst[hs] = std::min(_numPackets, static_cast<uint64_t>(std::round(W * _numPackets)));
            }
            _sourceFirst[Ns] = _numPackets;
            const double Lpp = Lsys / _numPackets;
            for (int hs = 0; hs < Ns; ++hs) _sceneSources[hs].packet_luminosity = _numPackets ? Lpp * (Lv[hs] / Wv[hs]) : 0.;
        }
        _scene.source = _sceneSources[0];
        if (Ns > 1)
        {
            _scene.num_sources = Ns;
            _scene.sources = _sceneSources.data();
            _scene.source_first = _sourceFirst.data();
        }
    }

    // ================================================================ scene flattening

    // spatial part of one source (PointSource.cpp:32-43, GeometricSource.cpp:66-82 with the geometry's generatePosition)
    void Simulation::flattenSourceGeometry(const SourceModel& _source, pmc_source& s) const
    {
        if (_source.type == "PointSource")
        {
            s.kind = PMC_SOURCE_POINT;
            s.position[0] = _source.position.x;
            s.position[1] = _source.position.y;
            s.position[2] = _source.position.z;
            s.angular_kind = _source.angularKind;
            s.angular_axis[0] = _source.angularAxis.x, s.angular_axis[1] = _source.angularAxis.y, s.angular_axis[2] = _source.angularAxis.z;
            s.angular_cos_delta = _source.angularCosDelta;
        }
        const Geometry* shape = _source.geometry.get();
        if (auto off = dynamic_cast<const OffsetGeometry*>(shape))
        {
            // OffsetGeometryDecorator::generatePosition: the offset travels in the position member
            shape = off->inner();
            s.position[0] = off->offset().x, s.position[1] = off->offset().y, s.position[2] = off->offset().z;
        }
        double flattening = 0.;
        if (auto sph = dynamic_cast<const SpheroidalGeometry*>(shape))
        {
            shape = sph->inner();
            flattening = sph->flattening();
        }
        if (_source.type == "PointSource") {}
        else if (auto sersic = dynamic_cast<const SersicGeometry*>(shape))
        {
            s.kind = PMC_SOURCE_SERSIC;
            s.reff = sersic->effectiveRadius();
            s.sersic_n = static_cast<int32_t>(sersic->profile().radii().size());
            s.sersic_s = sersic->profile().radii().data();
            s.sersic_M = sersic->profile().masses().data();
        }
        else if (auto ubox = dynamic_cast<const UniformBoxGeometry*>(shape))
        {
            s.kind = PMC_SOURCE_UNIFORM_BOX;
            const Box& b = ubox->box();
            double v[6] = {b.xmin, b.ymin, b.zmin, b.xmax, b.ymax, b.zmax};
            std::memcpy(s.box, v, sizeof(v));
        }
        else if (auto disk = dynamic_cast<const ExpDiskGeometry*>(shape))
        {
            s.kind = PMC_SOURCE_EXP_DISK;
            disk->parameters(s.box);
        }
        else if (auto plummer = dynamic_cast<const PlummerGeometry*>(shape))
        {
            s.kind = PMC_SOURCE_PLUMMER;
            s.box[0] = plummer->scaleLength();
        }
        if (flattening) s.box[5] = flattening;  // SpheroidalGeometryDecorator of a Sersic or Plummer source: z -> q z
    }

    void Simulation::buildScene()
    {
        std::memset(&_scene, 0, sizeof(_scene));
        _scene.abi_version = PMC_ABI_VERSION;

        pmc_grid& g = _scene.grid;
        g.xmin = _grid->extent.xmin;
        g.ymin = _grid->extent.ymin;
        g.zmin = _grid->extent.zmin;
        g.xmax = _grid->extent.xmax;
        g.ymax = _grid->extent.ymax;
        g.zmax = _grid->extent.zmax;
        g.eps = 1e-12 * _grid->extent.diagonal();
        g.num_cells = _grid->numCells();
        _grid->fill(g);

        _sceneMedia.assign(_media.size(), pmc_medium());
        for (size_t h = 0; h != _media.size(); ++h)
        {
            pmc_medium& m = _sceneMedia[h];
            const DustMix& mix = *_media[h]->mix;
            m.number_density = _density[h].data();
            m.num_lambda = static_cast<int32_t>(mix.lambdaBorder.size());
            m.lambda_border = mix.lambdaBorder.data();
            m.sigma_ext = mix.sigmaExt.data();
            m.sigma_sca = mix.sigmaSca.data();
            m.sigma_abs = mix.sigmaAbs.data();
            m.asymmpar = mix.asymmpar.data();
        }
        _scene.medium = _sceneMedia[0];
        _scene.num_media = static_cast<int32_t>(_media.size());
        _scene.media = _media.size() > 1 ? _sceneMedia.data() : nullptr;

        _scene.options = _options;

        // instruments (DistantInstrument.cpp:13-51, FrameInstrument.cpp:12-33, FullInstrument.cpp:11-17)
        _pmcInstruments.assign(_instruments.size(), pmc_instrument{});
        for (size_t i = 0; i < _instruments.size(); ++i)
        {
            const InstrumentModel& ins = _instruments[i];
            pmc_instrument& p = _pmcInstruments[i];
            p.costheta = cos(ins.inclination);
            p.sintheta = sin(ins.inclination);
            p.cosphi = cos(ins.azimuth);
            p.sinphi = sin(ins.azimuth);
            p.cosomega = cos(ins.roll);
            p.sinomega = sin(ins.roll);
            // Direction(theta, phi) (Direction.cpp:11-38)
            const double eps = 1e-8;
            double theta = ins.inclination, phi = ins.azimuth;
            if (theta < -eps || theta > M_PI + eps) throw std::runtime_error("Theta should be between 0 and pi.");
            if (theta <= eps)
                p.kobs[0] = 0, p.kobs[1] = 0, p.kobs[2] = 1;
            else if (theta >= M_PI - eps)
                p.kobs[0] = 0, p.kobs[1] = 0, p.kobs[2] = -1;
            else
            {
                double sintheta = sin(theta);
                p.kobs[0] = sintheta * cos(phi);
                p.kobs[1] = sintheta * sin(phi);
                p.kobs[2] = cos(theta);
            }
            p.nxp = ins.numPixelsX;
            p.nyp = ins.numPixelsY;
            p.xpmin = ins.centerX - 0.5 * ins.fieldOfViewX;
            p.xpsiz = ins.fieldOfViewX / ins.numPixelsX;
            p.ypmin = ins.centerY - 0.5 * ins.fieldOfViewY;
            p.ypsiz = ins.fieldOfViewY / ins.numPixelsY;
            p.same_observer_as_preceding = ins.sameObserverAsPreceding;
            p.include_flux_density = ins.type != "FrameInstrument";
            p.include_surface_brightness = ins.type != "SEDInstrument";
            if (ins.type == "SEDInstrument")
            {
                // no frame: one pixel that covers everything, so that the detection code finds bin 0 (FluxRecorder::detect is
                // called with l = 0, SEDInstrument.cpp:19-22)
                p.nxp = p.nyp = 1;
                p.xpmin = p.ypmin = -0.25 * DBL_MAX;
                p.xpsiz = p.ypsiz = 0.5 * DBL_MAX;
            }
            p.aperture_radius2 = ins.radius * ins.radius;
            // (FluxRecorder.cpp:199: without a medium the total flux is all there is to record)
            p.record_components = ins.recordComponents && _hasMedium;
            p.num_scattering_levels = p.record_components ? ins.numScatteringLevels : 0;
            p.record_statistics = ins.recordStatistics;
            p.redshift = ins.redshift;
            p.num_lambda = ins.grid->numBins();
            p.num_border = static_cast<int32_t>(ins.grid->borderv.size());
            p.border = ins.grid->borderv.data();
            p.ellv = ins.grid->ellv.data();
        }
        _scene.num_instruments = static_cast<int32_t>(_pmcInstruments.size());
        _scene.instruments = _pmcInstruments.data();
        _scene.radiation_field = pmc_radiation_field{};
        if (_storeRadiationField)
        {
            pmc_radiation_field& R = _scene.radiation_field;
            R.store = 1;
            R.num_lambda = _rfGrid->numBins();
            R.num_border = static_cast<int32_t>(_rfGrid->borderv.size());
            R.border = _rfGrid->borderv.data();
            R.ellv = _rfGrid->ellv.data();
        }

        _layouts.resize(_instruments.size());
        for (size_t i = 0; i < _instruments.size(); ++i) _frameSize = pmc_layout_compute(&_scene, static_cast<int32_t>(i), &_layouts[i]);
    }

    std::string Simulation::summary() const
    {
        std::ostringstream s;
        s << "simulation " << _prefix << ": " << (_oligo ? "oligochromatic" : "panchromatic") << ", " << _numPackets
          << " packets, seed " << _seed << "\n";
        s << "  grid: " << (_scene.grid.kind == PMC_GRID_CARTESIAN ? "Cartesian" : _scene.grid.kind == PMC_GRID_VORONOI ? "Voronoi" : "octree") << " with " << _scene.grid.num_cells
          << " cells";
        if (_scene.grid.kind == PMC_GRID_OCTREE) s << " (" << _scene.grid.num_nodes << " nodes)";
        s << "\n  dust table: " << _scene.medium.num_lambda << " wavelengths; setup draws: " << _random.draws() << "\n";
        s << "  instruments: " << _instruments.size() << "; frame buffer: " << _frameSize << " doubles\n";
        return s.str();
    }
}
