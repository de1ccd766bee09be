// output.cpp -- calibration of the detector arrays and the output files of an instrument:
//   <prefix>_<instr>_sed.dat, _sedstats.dat, _total/_transparent/_primarydirect/_primaryscattered[N].fits, _stats0..4.fits
// restating FluxRecorder::calibrateAndWrite (SKIRT/core/FluxRecorder.cpp:484-846), TextOutFile (TextOutFile.cpp:47-82)
// and FITSInOut::write (FITSInOut.cpp:127-215).  The FITS files are byte-compatible with the reference's CFITSIO
// output (same cards, same order, same value formatting) except for the DATE card.

#include "simulation.hpp"
#include "units.hpp"
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>

namespace skh
{
    namespace
    {
        // ---- FITS header cards, formatted as CFITSIO's ffpky* routines do
        std::string card(const std::string& key, const std::string& valueField, const std::string& comment)
        {
            char buf[256];
            std::string s = key;
            s.resize(8, ' ');
            s += "= ";
            std::string v = valueField;
            if (v.size() < 20) v = std::string(20 - v.size(), ' ') + v;  // numeric values are right-justified to column 30
            s += v + " / " + comment;
            (void)buf;
            s.resize(80, ' ');
            return s.substr(0, 80);
        }
        std::string cardString(const std::string& key, const std::string& value, const std::string& comment)
        {
            std::string q = value;
            if (q.size() < 8) q.resize(8, ' ');  // fixed-format strings hold at least 8 characters
            std::string v = "'" + q + "'";
            if (v.size() < 20) v.resize(20, ' ');  // left-justified, padded to column 30
            std::string s = key;
            s.resize(8, ' ');
            s += "= " + v + " / " + comment;
            s.resize(80, ' ');
            return s.substr(0, 80);
        }
        std::string cardInt(const std::string& key, long value, const std::string& comment)
        {
            return card(key, std::to_string(value), comment);
        }
        std::string cardLogical(const std::string& key, bool value, const std::string& comment)
        {
            return card(key, value ? "T" : "F", comment);
        }
        std::string cardDouble(const std::string& key, double value, const std::string& comment)
        {
            char buf[64];
            snprintf(buf, sizeof(buf), "%.9E", value);  // ffpkyd with 9 decimals
            return card(key, buf, comment);
        }
        std::string cardFixed0(const std::string& key, double value, const std::string& comment)
        {
            char buf[64];
            snprintf(buf, sizeof(buf), "%.0f", value);  // ffpkyg with 0 decimals
            return card(key, buf, comment);
        }
        std::string cardComment(const std::string& text)
        {
            std::string s = "COMMENT   " + text;
            s.resize(80, ' ');
            return s.substr(0, 80);
        }
        void padBlock(std::string& s, char fill)
        {
            size_t rem = s.size() % 2880;
            if (rem) s.append(2880 - rem, fill);
        }
        std::string utcStamp()
        {
            std::time_t t = std::time(nullptr);
            std::tm tm{};
            gmtime_r(&t, &tm);
            char buf[32];
            strftime(buf, sizeof(buf), "%Y-%m-%dT%H:%M:%S", &tm);
            return buf;
        }
    }

    void writeFitsCube(const std::string& filepath, const double* data, const std::string& dataUnits, int nx, int ny,
                       double incx, double incy, double xc, double yc, const std::string& xyUnits, const Array& z,
                       const std::string& zUnits, const FitsObserverInfo* obsInfo)
    {
        int nz = static_cast<int>(z.size());
        size_t nelements = size_t(nx) * size_t(ny) * size_t(nz ? nz : 1);

        std::string header;
        header += cardLogical("SIMPLE", true, "file does conform to FITS standard");
        header += cardInt("BITPIX", -32, "number of bits per data pixel");
        header += cardInt("NAXIS", nz ? 3 : 2, "number of data axes");
        header += cardInt("NAXIS1", nx, "length of data axis 1");
        header += cardInt("NAXIS2", ny, "length of data axis 2");
        if (nz) header += cardInt("NAXIS3", nz, "length of data axis 3");
        header += cardLogical("EXTEND", true, "FITS dataset may contain extensions");
        header += cardComment("FITS (Flexible Image Transport System) format is defined in 'Astronomy");
        header += cardComment("and Astrophysics', volume 376, page 359; bibcode: 2001A&A...376..359H");
        header += cardFixed0("BSCALE", 1., "Array value scale");
        header += cardFixed0("BZERO", 0., "Array value offset");
        header += cardString("DATE", utcStamp(), "Date and time of creation (UTC)");
        header += cardString("ORIGIN", "SKIRT simulation", "Astronomical Observatory, Ghent University");
        header += cardString("BUNIT", dataUnits, "Physical unit of the array values");
        header += cardDouble("CRPIX1", (nx + 1) / 2., "X-axis coordinate system reference pixel");
        header += cardDouble("CRVAL1", xc, "Coordinate value at X-axis reference pixel");
        header += cardDouble("CDELT1", incx, "Coordinate increment along X-axis");
        header += cardString("CUNIT1", xyUnits, "Physical units of the X-axis");
        header += cardString("CTYPE1", " ", "Linear X coordinates");
        header += cardDouble("CRPIX2", (ny + 1) / 2., "Y-axis coordinate system reference pixel");
        header += cardDouble("CRVAL2", yc, "Coordinate value at Y-axis reference pixel");
        header += cardDouble("CDELT2", incy, "Coordinate increment along Y-axis");
        header += cardString("CUNIT2", xyUnits, "Physical units of the Y-axis");
        header += cardString("CTYPE2", " ", "Linear Y coordinates");
        if (nz) header += cardString("CUNIT3", zUnits, "Physical units of the Z-axis");
        if (obsInfo)
        {
            header += cardDouble("CROTA1", obsInfo->inclination, "Inclination angle, in deg");
            header += cardDouble("CROTA2", obsInfo->azimuth, "Azimuth angle, in deg");
            header += cardDouble("CROTA3", obsInfo->roll, "Roll angle, in deg");
            header += cardDouble("REDSHIFT", obsInfo->redshift, "Redshift (if zero, distances are equal)");
            header += cardDouble("DISTLUMI", obsInfo->luminosityDistance, "Luminosity distance");
            header += cardDouble("DISTANGD", obsInfo->angularDiameterDistance, "Angular diameter distance");
            header += cardString("DISTUNIT", obsInfo->distanceUnits, "Units of distances");
        }
        std::string end = "END";
        end.resize(80, ' ');
        header += end;
        padBlock(header, ' ');

        // pixel data: IEEE float32, big endian
        std::string pixels(nelements * 4, '\0');
        for (size_t i = 0; i < nelements; ++i)
        {
            float f = static_cast<float>(data[i]);
            uint32_t u;
            std::memcpy(&u, &f, 4);
            pixels[4 * i + 0] = static_cast<char>(u >> 24);
            pixels[4 * i + 1] = static_cast<char>(u >> 16);
            pixels[4 * i + 2] = static_cast<char>(u >> 8);
            pixels[4 * i + 3] = static_cast<char>(u);
        }
        padBlock(pixels, '\0');

        std::string table;
        if (nz)
        {
            std::string th;
            th += cardString("XTENSION", "TABLE", "ASCII table extension");
            th += cardInt("BITPIX", 8, "8-bit ASCII characters");
            th += cardInt("NAXIS", 2, "2-dimensional ASCII table");
            th += cardInt("NAXIS1", 16, "width of table in characters");
            th += cardInt("NAXIS2", nz, "number of rows in table");
            th += cardInt("PCOUNT", 0, "no group parameters (required keyword)");
            th += cardInt("GCOUNT", 1, "one data group (required keyword)");
            th += cardInt("TFIELDS", 1, "number of fields in each row");
            th += cardString("TTYPE1", "GRID_POINTS", "label for field   1");
            th += cardInt("TBCOL1", 1, "beginning column of field   1");
            th += cardString("TFORM1", "E16.9", "Fortran-77 format of field");
            th += cardString("TUNIT1", zUnits, "physical unit of field");
            th += cardString("EXTNAME", "Z-axis coordinate values", "name of this ASCII table extension");
            th += end;
            padBlock(th, ' ');
            std::string rows;
            for (int i = 0; i < nz; ++i)
            {
                char buf[64];
                snprintf(buf, sizeof(buf), "%16.9E", z[i]);
                rows += buf;
            }
            padBlock(rows, ' ');
            table = th + rows;
        }

        std::remove(filepath.c_str());
        std::ofstream out(filepath, std::ios::binary);
        if (!out) throw std::runtime_error("Could not create FITS file " + filepath);
        out.write(header.data(), header.size());
        out.write(pixels.data(), pixels.size());
        out.write(table.data(), table.size());
        if (!out) throw std::runtime_error("Error writing FITS file " + filepath);
    }

    namespace
    {
        // StringUtils::toString(double) (StringUtils.cpp:386-412)
        std::string smartString(double value)
        {
            char buf[32];
            snprintf(buf, sizeof(buf), "%1.10g", value);
            std::string result(buf);
            auto replaceAll = [&](const std::string& a, const std::string& b) {
                size_t p = 0;
                while ((p = result.find(a, p)) != std::string::npos)
                {
                    result.replace(p, a.size(), b);
                    p += b.size();
                }
            };
            replaceAll("e-0", "e-");
            replaceAll("e+0", "e");
            replaceAll("e+", "e");
            size_t zeroes = 0;
            for (auto it = result.crbegin(); it != result.crend(); ++it, ++zeroes)
                if (*it != '0') break;
            if (zeroes > 3)
            {
                result.erase(result.length() - zeroes);
                result += "e" + std::to_string(zeroes);
            }
            return result;
        }

        class TextColumns
        {
        public:
            explicit TextColumns(const std::string& path) : _out(path)
            {
                if (!_out) throw std::runtime_error("Could not open output file " + path);
            }
            void line(const std::string& s) { _out << s << std::endl; }
            void column(const std::string& quantity, std::string unit = std::string())
            {
                if (unit.empty()) unit = "1";
                line("# column " + std::to_string(++_ncolumns) + ": " + quantity + " (" + unit + ")");
            }
            void row(const std::vector<double>& values)
            {
                std::string s;
                for (size_t i = 0; i < values.size(); ++i)
                {
                    char buf[40];
                    snprintf(buf, sizeof(buf), "%1.9e", values[i]);
                    s += (i ? " " : "");
                    s += buf;
                }
                line(s);
            }

        private:
            std::ofstream _out;
            int _ncolumns{0};
        };
    }

    int64_t Simulation::radiationFieldSize() const
    {
        return _storeRadiationField ? static_cast<int64_t>(_grid->numCells()) * _rfGrid->numBins() : 0;
    }

    // RadiationFieldProbe::probe with a PerCellForm (RadiationFieldProbe.cpp:27-78, PerCellForm.cpp:14-32): the mean
    // intensity J = rf / (4 pi V dlambda) (MediumSystem::meanIntensity, MediumSystem.cpp:1370-1380) of every cell, one
    // column per wavelength bin, in output units
    std::vector<std::string> Simulation::writeRadiationField(const double* rf, const std::string& outdir) const
    {
        std::vector<std::string> files;
        if (!_storeRadiationField) return files;
        std::string base = outdir;
        if (!base.empty() && base.back() != '/') base += '/';
        base += _prefix + "_";
        const WavelengthGrid& wlg = *_rfGrid;
        const int nbins = wlg.numBins();
        // Units::rwavelength(): the output runs over decreasing wavelength unless wavelengths are written as such
        const bool reverse = _units.wavelengthStyle != "Wavelength";
        std::vector<int> order(nbins);
        for (int i = 0; i != nbins; ++i) order[i] = reverse ? nbins - 1 - i : i;
        Array conv(nbins);
        for (int i = 0; i != nbins; ++i) conv[i] = _units.omeanintensity(wlg.wavelength(order[i]), 1.);
        for (const std::string& name : _rfProbeNames)
        {
            std::string path = base + name + "_J.dat";
            std::ofstream out(path);
            if (!out) throw std::runtime_error("Could not open output file " + path);
            out << "# Mean intensity per spatial cell" << std::endl;
            out << "# column 1: spatial cell index (1)" << std::endl;
            for (int i = 0; i != nbins; ++i)
            {
                char buf[40];
                snprintf(buf, sizeof(buf), "%1.6g", _units.owavelength(wlg.wavelength(order[i])));
                out << "# column " << (i + 2) << ": " << _units.smeanintensity() << " at " << _units.swavelength() << " = " << buf
                    << " " << _units.uwavelength() << " (" << _units.umeanintensity() << ")" << std::endl;
            }
            const int numCells = _grid->numCells();
            for (int m = 0; m != numCells; ++m)
            {
                // MediumSystem::meanIntensity
                double factor = 1. / (4. * M_PI * _grid->volume(m));
                std::string line = std::to_string(m);
                for (int i = 0; i != nbins; ++i)
                {
                    int ell = order[i];
                    double J = rf[static_cast<size_t>(m) * nbins + ell] * factor / wlg.effectiveWidth(ell);
                    char buf[40];
                    snprintf(buf, sizeof(buf), " %1.9e", conv[i] * J);
                    line += buf;
                }
                out << line << std::endl;
            }
            files.push_back(path);
        }
        return files;
    }

    std::vector<std::string> Simulation::write(double* frames, const std::string& outdir, bool writeStatistics) const
    {
        std::vector<std::string> files;
        std::string base = outdir;
        if (!base.empty() && base.back() != '/') base += '/';
        base += _prefix + "_";

        for (int i = 0; i < numInstruments(); ++i)
        {
            const InstrumentModel& ins = _instruments[i];
            const pmc_instrument& p = _pmcInstruments[i];
            const pmc_frame_layout& L = _layouts[i];
            const WavelengthGrid& grid = *ins.grid;
            const int numWavelengths = grid.numBins();
            const size_t npix = static_cast<size_t>(L.npix);
            const int ncomp = static_cast<int>(L.num_components);
            const bool totalOnly = !p.record_components;
            const int maxContributionPower = 4;

            // setRestFrameDistance / setObserverFrameRedshift (FluxRecorder.cpp:112-128)
            double luminosityDistance = ins.luminosityDistance, angularDiameterDistance = ins.angularDiameterDistance;
            double pixelSizeX = ins.fieldOfViewX / ins.numPixelsX, pixelSizeY = ins.fieldOfViewY / ins.numPixelsY;
            double fourpid2 = 4. * M_PI * (luminosityDistance * luminosityDistance);
            double omega = 4. * atan(0.5 * pixelSizeX / angularDiameterDistance) * atan(0.5 * pixelSizeY / angularDiameterDistance);

            auto header = [&](const std::string& name) {
                std::string h = "# " + name + " at ";
                h += "inclination " + smartString(_units.out("posangle", ins.inclination)) + " " + _units.unit("posangle");
                h += ", azimuth " + smartString(_units.out("posangle", ins.azimuth)) + " " + _units.unit("posangle");
                if (ins.redshift)
                {
                    h += ", redshift " + smartString(ins.redshift);
                    h += ", luminosity distance " + smartString(_units.out("distance", luminosityDistance)) + " " + _units.unit("distance");
                }
                else
                    h += ", distance " + smartString(_units.out("distance", luminosityDistance)) + " " + _units.unit("distance");
                return h;
            };

            // ---------------- SED
            if (L.sed_offset >= 0)
            {
                double* sed = frames + L.sed_offset;  // [c][ell]
                for (int ell = 0; ell != numWavelengths; ++ell)
                {
                    double factor = 1. / fourpid2 / grid.effectiveWidth(ell) * _units.ofluxdensity(grid.wavelength(ell), 1.);
                    for (int c = 0; c < ncomp; ++c) sed[c * numWavelengths + ell] *= factor;
                }
                // column list (buildCols)
                std::vector<std::string> names;
                std::vector<Array> cols;
                auto comp = [&](int c) { return Array(sed + c * numWavelengths, sed + (c + 1) * numWavelengths); };
                Array zero(numWavelengths, 0.);
                names.push_back("total flux");
                if (totalOnly)
                    cols.push_back(comp(0));
                else
                {
                    Array total(numWavelengths);
                    for (int ell = 0; ell < numWavelengths; ++ell)
                        total[ell] = sed[1 * numWavelengths + ell] + sed[2 * numWavelengths + ell];
                    cols.push_back(total);
                }
                if (ins.recordComponents)
                {
                    names.push_back("transparent flux");
                    cols.push_back(comp(0));
                    names.insert(names.end(), {"direct primary flux", "scattered primary flux", "direct secondary flux",
                                               "scattered secondary flux", "transparent secondary flux"});
                    cols.push_back(totalOnly ? zero : comp(1));
                    cols.push_back(totalOnly ? zero : comp(2));
                    cols.push_back(zero);
                    cols.push_back(zero);
                    cols.push_back(zero);
                    for (int lev = 0; lev != ins.numScatteringLevels; ++lev)
                    {
                        names.push_back(std::to_string(lev + 1) + "-times scattered primary flux");
                        cols.push_back(totalOnly ? zero : comp(3 + lev));
                    }
                }
                std::string path = base + ins.name + "_sed.dat";
                {
                    TextColumns f(path);
                    f.line(header("SED"));
                    f.column("wavelength; " + _units.swavelength(), _units.uwavelength());
                    for (auto& n : names) f.column(n + "; " + _units.sfluxdensity(), _units.ufluxdensity());
                    for (int ell = 0; ell != numWavelengths; ++ell)
                    {
                        std::vector<double> values{_units.owavelength(grid.wavelength(ell))};
                        for (auto& ccol : cols) values.push_back(ccol[ell]);
                        f.row(values);
                    }
                }
                files.push_back(path);
                if (L.wsed_offset >= 0 && writeStatistics)
                {
                    const double* wsed = frames + L.wsed_offset;
                    std::string spath = base + ins.name + "_sedstats.dat";
                    TextColumns f(spath);
                    f.column("wavelength; " + _units.swavelength(), _units.uwavelength());
                    for (int k = 0; k <= maxContributionPower; ++k) f.column("Sum[w_i**" + std::to_string(k) + "]");
                    f.line("# --> w_i is luminosity contribution (in W) from i_th launched photon");
                    for (int ell = 0; ell != numWavelengths; ++ell)
                    {
                        std::vector<double> values{_units.owavelength(grid.wavelength(ell))};
                        for (int k = 0; k <= maxContributionPower; ++k) values.push_back(wsed[k * numWavelengths + ell]);
                        f.row(values);
                    }
                    files.push_back(spath);
                }
            }

            // ---------------- IFU
            if (L.ifu_offset >= 0)
            {
                double* ifu = frames + L.ifu_offset;  // [c][l + ell*npix]
                const size_t len = npix * numWavelengths;
                for (int ell = 0; ell != numWavelengths; ++ell)
                {
                    double factor = 1. / fourpid2 / omega / grid.effectiveWidth(ell)
                                    * _units.osurfacebrightness(grid.wavelength(ell), 1.);
                    size_t begin = ell * npix, end = begin + npix;
                    for (int c = 0; c < ncomp; ++c)
                        for (size_t lell = begin; lell != end; ++lell) ifu[c * len + lell] *= factor;
                }
                Array wavegrid(numWavelengths);
                for (int ell = 0; ell != numWavelengths; ++ell) wavegrid[ell] = _units.owavelength(grid.wavelength(ell));

                double incx = _units.out("angle", 2. * atan(0.5 * pixelSizeX / angularDiameterDistance));
                double incy = _units.out("angle", 2. * atan(0.5 * pixelSizeY / angularDiameterDistance));
                double cx = _units.out("angle", 2. * atan(0.5 * ins.centerX / angularDiameterDistance));
                double cy = _units.out("angle", 2. * atan(0.5 * ins.centerY / angularDiameterDistance));
                std::string unitsxy = _units.unit("angle");

                FitsObserverInfo info;
                info.inclination = ins.inclination * (180. / M_PI);
                info.azimuth = ins.azimuth * (180. / M_PI);
                info.roll = ins.roll * (180. / M_PI);
                info.redshift = ins.redshift;
                info.luminosityDistance = _units.out("distance", luminosityDistance);
                info.angularDiameterDistance = _units.out("distance", angularDiameterDistance);
                info.distanceUnits = _units.unit("distance");

                // file list (buildFiles)
                std::vector<std::pair<std::string, const double*>> outputs;
                Array total;
                if (totalOnly)
                    outputs.emplace_back("total", ifu);
                else
                {
                    total.resize(len);
                    for (size_t q = 0; q < len; ++q) total[q] = ifu[1 * len + q] + ifu[2 * len + q];
                    outputs.emplace_back("total", total.data());
                    outputs.emplace_back("transparent", ifu);
                    outputs.emplace_back("primarydirect", ifu + len);
                    outputs.emplace_back("primaryscattered", ifu + 2 * len);
                    for (int lev = 0; lev != ins.numScatteringLevels; ++lev)
                        outputs.emplace_back("primaryscattered" + std::to_string(lev + 1), ifu + (3 + lev) * len);
                }
                for (auto& o : outputs)
                {
                    std::string path = base + ins.name + "_" + o.first + ".fits";
                    writeFitsCube(path, o.second, _units.usurfacebrightness(), ins.numPixelsX, ins.numPixelsY, incx, incy, cx, cy,
                                  unitsxy, wavegrid, _units.uwavelength(), &info);
                    files.push_back(path);
                }

                if (L.wifu_offset >= 0 && writeStatistics)
                {
                    double* wifu = frames + L.wifu_offset;  // [k][lell]
                    const double WMAX = 1e38;
                    double c = 0.;
                    for (int k = 1; k <= maxContributionPower; ++k)
                    {
                        double mx = wifu[k * len];
                        for (size_t q = 1; q < len; ++q) mx = std::max(mx, wifu[k * len + q]);
                        double cs = pow(WMAX / mx, 1. / k);
                        c = (k == 1) ? cs : std::min(c, cs);
                    }
                    double cn = 1.;
                    for (int k = 0; k <= maxContributionPower; ++k)
                    {
                        for (size_t q = 0; q < len; ++q) wifu[k * len + q] *= cn;
                        std::string path = base + ins.name + "_stats" + std::to_string(k) + ".fits";
                        writeFitsCube(path, wifu + k * len, "", ins.numPixelsX, ins.numPixelsY, incx, incy, cx, cy, unitsxy,
                                      wavegrid, _units.uwavelength(), nullptr);
                        files.push_back(path);
                        cn *= c;
                    }
                }
            }
        }
        return files;
    }
}
