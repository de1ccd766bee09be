// particles.cpp -- see particles.hpp
#include "particles.hpp"
#include "units.hpp"
#include <cstring>
#include <fstream>
#include <limits>
#include <regex>
#include <sstream>
#include <stdexcept>

namespace skh
{
    // ================================================================ smoothing kernels

    namespace
    {
        // CubicSplineSmoothingKernel.cpp:40-50
        class CubicSplineSmoothingKernel : public SmoothingKernel
        {
        public:
            std::string type() const override { return "CubicSplineSmoothingKernel"; }
            double density(double u) const override
            {
                if (!(u >= 0.0 && u < 1.0)) return 0.0;
                const double norm = 8.0 / M_PI;
                const double rest = 1.0 - u;
                if (u < 0.5) return norm * (1.0 - 6.0 * u * u * rest);  // inner half: 1 - 6 u^2 (1 - u)
                return norm * 2.0 * rest * rest * rest;                 // outer half: 2 (1 - u)^3
            }
        };
        // ScaledGaussianSmoothingKernel.cpp:16-47
        class ScaledGaussianSmoothingKernel : public SmoothingKernel
        {
        public:
            std::string type() const override { return "ScaledGaussianSmoothingKernel"; }
            double density(double u) const override
            {
                // front factor and exponent of the Gaussian truncated at the smoothing length
                static const double front = 2.56810060330949540082, exponent = -5.85836755024609305208;
                if (!(u >= 0. && u <= 1.)) return 0.;
                return front * exp(exponent * u * u);
            }
        };
        // UniformSmoothingKernel.cpp:13-17
        class UniformSmoothingKernel : public SmoothingKernel
        {
        public:
            std::string type() const override { return "UniformSmoothingKernel"; }
            double density(double u) const override
            {
                return (u >= 0.0 && u <= 1.0) ? 0.75 / M_PI : 0.0;
            }
        };
    }

    std::unique_ptr<SmoothingKernel> SmoothingKernel::create(const std::string& type)
    {
        if (type == "CubicSplineSmoothingKernel") return std::make_unique<CubicSplineSmoothingKernel>();
        if (type == "ScaledGaussianSmoothingKernel") return std::make_unique<ScaledGaussianSmoothingKernel>();
        if (type == "UniformSmoothingKernel") return std::make_unique<UniformSmoothingKernel>();
        throw std::runtime_error("smoothing kernel " + type + " is not supported on the MI355X path");
    }

    // ================================================================ block search grid (BoxSearch.cpp:14-127)

    namespace
    {
        // Separators of one axis that put about the same number of entity centres into every slab: a histogram of the
        // centres with 100 bins per slab, then a separator at the upper edge of the bin in which the running count passes
        // the next multiple of (entities / slabs) -- at most one separator per bin; separators that are never reached stay
        // zero, the outermost ones are infinite.
        Array balancedSeparators(const Array& centres, int slabs, double lower, double upper)
        {
            if (lower == upper)
            {
                // a degenerate extent is widened by a relative 1e-12
                const double margin = 1e-12 * (lower ? std::abs(lower) : 1.);
                lower -= margin;
                upper += margin;
            }
            const int bins = slabs * 100;
            const double width = (upper - lower) / bins;
            std::vector<int> histogram(bins);
            for (double centre : centres) histogram[static_cast<int>((centre - lower) / width)] += 1;

            const double quota = static_cast<double>(centres.size()) / slabs;
            const double infinite = std::numeric_limits<double>::infinity();
            Array separators(slabs + 1, 0.);
            separators.front() = -infinite;
            separators.back() = infinite;
            int running = 0, next = 1;
            for (int bin = 0; bin < bins && next < slabs; ++bin)
            {
                running += histogram[bin];
                if (running > quota * next) separators[next++] = lower + (bin + 1) * width;
            }
            return separators;
        }

        // does the sphere (centre c, radius r) reach into the box?  The squared radius minus the squared distance from
        // the centre to the box, accumulated axis by axis (Box.cpp:91-109)
        bool sphereReachesBox(const Box& box, double cx, double cy, double cz, double r)
        {
            auto outside = [](double c, double low, double high) { return c < low ? c - low : c > high ? c - high : 0.; };
            const double ex = outside(cx, box.xmin, box.xmax);
            const double ey = outside(cy, box.ymin, box.ymax);
            const double ez = outside(cz, box.zmin, box.zmax);
            double remaining = r * r;
            remaining -= ex * ex;
            remaining -= ey * ey;
            remaining -= ez * ez;
            return remaining >= 0.;
        }
    }

    template<class Bounds, class Intersects> void BoxSearch::loadEntities(int numEntities, Bounds bounds, Intersects intersects)
    {
        _listv.clear();
        _extent = Box();
        _numBlocks = 0;
        if (numEntities <= 0) return;

        // bounding boxes, their union, and their centres per axis
        std::vector<Box> boxes(numEntities);
        Array midX(numEntities), midY(numEntities), midZ(numEntities);
        for (int m = 0; m != numEntities; ++m)
        {
            const Box box = bounds(m);
            boxes[m] = box;
            midX[m] = 0.5 * (box.xmin + box.xmax);
            midY[m] = 0.5 * (box.ymin + box.ymax);
            midZ[m] = 0.5 * (box.zmin + box.zmax);
            if (m == 0) _extent = box;
            _extent = Box(std::min(_extent.xmin, box.xmin), std::min(_extent.ymin, box.ymin), std::min(_extent.zmin, box.zmin),
                          std::max(_extent.xmax, box.xmax), std::max(_extent.ymax, box.ymax), std::max(_extent.zmax, box.zmax));
        }
        _numBlocks = std::max(10, static_cast<int>(std::cbrt(numEntities)));
        _xgrid = balancedSeparators(midX, _numBlocks, _extent.xmin, _extent.xmax);
        _ygrid = balancedSeparators(midY, _numBlocks, _extent.ymin, _extent.ymax);
        _zgrid = balancedSeparators(midZ, _numBlocks, _extent.zmin, _extent.zmax);

        // every entity is listed in the blocks its bounding box overlaps and it actually intersects; entities are visited in
        // index order, so every list is ascending
        _listv.resize(static_cast<size_t>(_numBlocks) * _numBlocks * _numBlocks);
        struct Span
        {
            int first, last;
        };
        auto span = [](const Array& separators, double low, double high) {
            return Span{tab::bracketClipped(separators, low), tab::bracketClipped(separators, high)};
        };
        for (int m = 0; m != numEntities; ++m)
        {
            const Box& box = boxes[m];
            const Span sx = span(_xgrid, box.xmin, box.xmax), sy = span(_ygrid, box.ymin, box.ymax), sz = span(_zgrid, box.zmin, box.zmax);
            for (int i = sx.first; i <= sx.last; ++i)
                for (int j = sy.first; j <= sy.last; ++j)
                    for (int k = sz.first; k <= sz.last; ++k)
                    {
                        const Box block(_xgrid[i], _ygrid[j], _zgrid[k], _xgrid[i + 1], _ygrid[j + 1], _zgrid[k + 1]);
                        if (intersects(m, block)) _listv[blockIndex(i, j, k)].push_back(m);
                    }
        }
    }

    const std::vector<int>& BoxSearch::entitiesFor(Vec3 r) const
    {
        if (!_numBlocks) return _empty;
        int i = tab::bracketClipped(_xgrid, r.x);
        int j = tab::bracketClipped(_ygrid, r.y);
        int k = tab::bracketClipped(_zgrid, r.z);
        return _listv[blockIndex(i, j, k)];
    }

    void BoxSearch::flatten(std::vector<int64_t>& start, std::vector<int32_t>& list) const
    {
        start.assign(_listv.size() + 1, 0);
        list.clear();
        for (size_t b = 0; b != _listv.size(); ++b)
        {
            start[b] = static_cast<int64_t>(list.size());
            list.insert(list.end(), _listv[b].begin(), _listv[b].end());
        }
        start[_listv.size()] = static_cast<int64_t>(list.size());
    }

    size_t BoxSearch::numReferences() const
    {
        size_t n = 0;
        for (const auto& l : _listv) n += l.size();
        return n;
    }

    // ================================================================ column text file

    namespace
    {
        std::string squeezeText(const std::string& text)
        {
            std::string out;
            bool haveSpace = true;
            for (char c : text)
            {
                if (c == ' ' || c == '\t' || c == '\n' || c == '\r')
                {
                    if (!haveSpace)
                    {
                        out.push_back(' ');
                        haveSpace = true;
                    }
                }
                else
                {
                    out.push_back(c);
                    haveSpace = false;
                }
            }
            if (haveSpace && !out.empty()) out.pop_back();
            return out;
        }
    }

    std::vector<Array> readColumnFile(const std::string& path, const std::vector<ColumnSpec>& columns, const std::string& description)
    {
        std::ifstream in(path);
        if (!in) throw std::runtime_error("Could not open the " + description + " text file " + path);
        // ---- header: "# column N: description (unit)" lines (TextInFile.cpp:16-47,87-101); other '#' lines are comments
        struct FileColumn
        {
            std::string title, unit;
        };
        std::vector<FileColumn> fileCols;
        static const std::regex syntax("#\\s*column\\s*(\\d*)\\s*:\\s*([^()]*)\\(\\s*([a-zA-Z0-9/]*)\\s*\\)\\s*", std::regex::icase);
        while (true)
        {
            while (true)
            {
                int ch = in.peek();
                if (ch != ' ' && ch != '\t' && ch != '\n' && ch != '\r') break;
                in.get();
            }
            if (in.peek() != '#') break;
            std::string line;
            std::getline(in, line);
            std::smatch matches;
            if (std::regex_match(line, matches, syntax) && matches.size() == 4)
            {
                std::string index = matches[1].str();
                fileCols.push_back(FileColumn{squeezeText(matches[2].str()), matches[3].str()});
                if (!index.empty() && std::stoul(index) != fileCols.size())
                    throw std::runtime_error("Incorrect column index in file header for column " + std::to_string(fileCols.size()));
            }
        }
        // ---- logical columns in file order (TextInFile::addColumn without column remapping, :214-262)
        const size_t ncol = columns.size();
        std::vector<UnitFactor> conv(ncol);
        for (size_t c = 0; c != ncol; ++c)
        {
            std::string unit = columns[c].defaultUnit;
            if (!fileCols.empty())
            {
                if (c + 1 > fileCols.size()) throw std::runtime_error("No column info in file header for column " + std::to_string(c + 1));
                unit = fileCols[c].unit;
            }
            if (columns[c].quantity == "specific")
            {
                // an arbitrarily scaled value per unit of wavelength (TextInFile.cpp:232-245): the unit must be one of the
                // per-wavelength styles, and NO conversion factor is applied (waveExponent 0); the per-frequency, neutral
                // and per-energy styles would need the wavelength column and are not supported here
                if (!unitTable().has("wavelengthmonluminosity", unit))
                    throw std::runtime_error("Invalid or unsupported units (" + unit + ") for specific quantity in column "
                                             + std::to_string(c + 1) + " (only per-wavelength units are supported)");
                conv[c] = UnitFactor{1., 1., 0.};
            }
            else if (columns[c].quantity.empty())
            {
                if (!unit.empty() && unit != "1")
                    throw std::runtime_error("Invalid units (" + unit + ") for dimensionless quantity in column " + std::to_string(c + 1));
                conv[c] = UnitFactor{1., 1., 0.};
            }
            else
            {
                if (!unitTable().has(columns[c].quantity, unit))
                    throw std::runtime_error("Invalid units (" + unit + ") for quantity '" + columns[c].quantity + "' in column "
                                             + std::to_string(c + 1));
                conv[c] = unitTable().factorOf(columns[c].quantity, unit);
            }
        }
        // ---- rows (TextInFile::readRow, :286-330): value = factor * value^power
        std::vector<Array> rows;
        std::string line;
        while (std::getline(in, line))
        {
            auto pos = line.find_first_not_of(" \t\r");
            if (pos == std::string::npos || line[pos] == '#') continue;
            Array row(ncol);
            const char* p = line.c_str() + pos;
            for (size_t c = 0; c != ncol; ++c)
            {
                while (*p == ' ' || *p == '\t') ++p;
                if (!*p || *p == '\r') throw std::runtime_error("One or more required value(s) on text line are missing");
                char* end = nullptr;
                double value = strtod(p, &end);
                if (end == p)
                {
                    if (strncasecmp(p, "nan", 3) != 0)
                        throw std::runtime_error(std::string("Input text is not formatted as a floating point number: ") + p);
                    value = std::numeric_limits<double>::quiet_NaN();
                    end = const_cast<char*>(p) + 3;
                }
                p = end;
                if (conv[c].power != 1.) value = pow(value, conv[c].power);
                value *= conv[c].factor;
                row[c] = value;
            }
            rows.push_back(std::move(row));
        }
        return rows;
    }

    // ================================================================ ParticleSnapshot

    void ParticleSnapshot::load(const ParticleImportOptions& o, std::unique_ptr<SmoothingKernel> kernel)
    {
        _kernel = std::move(kernel);
        _holdsNumber = o.holdsNumber;
        // column roles in the reference's order: ParticleMedium.cpp:19-30 then ImportedMedium.cpp:17-18
        std::vector<ColumnSpec> cols = {{"position x", "length", "pc"}, {"position y", "length", "pc"}, {"position z", "length", "pc"},
                                        {"size h", "length", "pc"}};
        const int massIndex = 4;
        if (o.holdsNumber)
            cols.push_back({"number", "", ""});
        else
            cols.push_back({"mass", "mass", "Msun"});
        int metallicityIndex = -1, temperatureIndex = -1;
        if (o.importMetallicity)
        {
            metallicityIndex = static_cast<int>(cols.size());
            cols.push_back({"metallicity", "", ""});
        }
        if (o.importTemperature)
        {
            temperatureIndex = static_cast<int>(cols.size());
            cols.push_back({"temperature", "temperature", "K"});
        }
        // Snapshot::setMassDensityPolicy as called by ImportedMedium (dust: Tmax and metallicity apply; otherwise not)
        const double maxTemperature = (o.isDust && o.importTemperature) ? o.maxTemperature : 0.;
        const bool useMetallicity = o.isDust && o.importMetallicity && metallicityIndex >= 0;
        const bool useTemperatureCutoff = maxTemperature > 0 && temperatureIndex >= 0;

        std::vector<Array> rows = readColumnFile(o.path, cols);
        // ParticleSnapshot::readAndClose (ParticleSnapshot.cpp:79-151)
        _pv.clear();
        _pv.reserve(rows.size());
        _sites.clear();
        _sites.reserve(rows.size());
        double sumImported = 0, sumMetallic = 0, sumEffective = 0;
        for (const Array& prop : rows)
        {
            _sites.push_back(Vec3{prop[0], prop[1], prop[2]});  // (every imported row: ParticleSnapshot::position(m), ParticleSnapshot.cpp:286-289)
            if (useTemperatureCutoff && prop[temperatureIndex] > maxTemperature) continue;
            if (prop[massIndex] == 0.) continue;
            double imported = prop[massIndex];
            double metallic = imported * (useMetallicity ? prop[metallicityIndex] : 1.);
            double effective = metallic * o.massFraction;
            _pv.push_back(Particle{prop[0], prop[1], prop[2], prop[3], effective});
            sumImported += imported;
            sumMetallic += metallic;
            sumEffective += effective;
        }
        if (sumImported < 0 || sumMetallic < 0 || sumEffective < 0)
        {
            _pv.clear();
            sumEffective = 0;
        }
        _mass = sumEffective;
        _search.loadEntities(
            static_cast<int>(_pv.size()),
            [this](int m) {
                const Particle& p = _pv[m];
                return Box(p.x - p.h, p.y - p.h, p.z - p.h, p.x + p.h, p.y + p.h, p.z + p.h);
            },
            [this](int m, const Box& box) { return sphereReachesBox(box, _pv[m].x, _pv[m].y, _pv[m].z, _pv[m].h); });

        // ---- optional: the same evaluation on the MI355X (include/pmc.h pmc_sampler_*)
        if (_api.create && !_pv.empty())
        {
            pmc_particles P{};
            if (_kernel->type() == "CubicSplineSmoothingKernel")
                P.kernel = PMC_KERNEL_CUBIC_SPLINE;
            else if (_kernel->type() == "UniformSmoothingKernel")
                P.kernel = PMC_KERNEL_UNIFORM;
            else
                throw std::runtime_error("device density sampler: " + _kernel->type()
                                         + " evaluates a transcendental function and would not be bit-identical; use the host");
            std::vector<double> table(5 * _pv.size());
            for (size_t m = 0; m != _pv.size(); ++m)
            {
                table[5 * m] = _pv[m].x, table[5 * m + 1] = _pv[m].y, table[5 * m + 2] = _pv[m].z, table[5 * m + 3] = _pv[m].h;
                table[5 * m + 4] = _pv[m].density();
            }
            std::vector<int64_t> start;
            std::vector<int32_t> list;
            _search.flatten(start, list);
            P.num_particles = static_cast<int64_t>(_pv.size());
            P.particle = table.data();
            P.num_blocks = _search.numBlocks();
            P.xgrid = _search.xgrid().data(), P.ygrid = _search.ygrid().data(), P.zgrid = _search.zgrid().data();
            P.block_start = start.data();
            P.block_list = list.data();
            if (_api.create(&P, _api.device, &_sampler) != 0)
                throw std::runtime_error(std::string("device density sampler: ") + (_api.lastError ? _api.lastError() : "create failed"));
        }
    }

    ParticleSnapshot::~ParticleSnapshot()
    {
        if (_sampler && _api.destroy) _api.destroy(_sampler);
    }

    void ParticleSnapshot::densities(const std::vector<Vec3>& positions, std::vector<double>& out) const
    {
        out.resize(positions.size());
        if (positions.empty()) return;
        if (_sampler)
        {
            static_assert(sizeof(Vec3) == 3 * sizeof(double), "Vec3 must be three packed doubles");
            if (_api.density(_sampler, reinterpret_cast<const double*>(positions.data()), static_cast<int64_t>(positions.size()), out.data())
                != 0)
                throw std::runtime_error(std::string("device density sampler: ") + (_api.lastError ? _api.lastError() : "failed"));
            return;
        }
        parallelFor(positions.size(), [&](size_t b, size_t e) {
            for (size_t i = b; i != e; ++i) out[i] = density(positions[i]);
        });
    }

    double ParticleSnapshot::density(Vec3 r) const
    {
        double sum = 0.;
        for (int m : _search.entitiesFor(r))
        {
            const Particle& p = _pv[m];
            const double dx = r.x - p.x, dy = r.y - p.y, dz = r.z - p.z;
            double u = sqrt(dx * dx + dy * dy + dz * dz) / p.h;
            sum += _kernel->density(u) * p.density();
        }
        return sum > 0. ? sum : 0.;
    }
}
