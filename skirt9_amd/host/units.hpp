// units.hpp -- physical constants and the unit conversions a ski file needs.
//
// SKIRT stores everything internally in SI; attribute strings such as "1 pc", "0.55 micron", "3000 m2/kg" carry
// a unit that is converted with value_SI = factor * value^power + offset (reference: SMILE/schema/UnitDef.cpp:64-101
// with the table in SKIRT/core/SkirtUnitDef.cpp:11-764 and constants in SKIRT/utils/Constants.hpp).  Only the
// quantities that the supported ski classes use are tabulated here; an unknown quantity/unit is a fatal error.
#ifndef SKH_UNITS_HPP
#define SKH_UNITS_HPP

#include <cmath>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace skh
{
    namespace constants
    {
        constexpr double c = 2.99792458e8;
        constexpr double h = 6.62606957e-34;
        constexpr double k = 1.3806488e-23;
        constexpr double AU = 1.49597871e11;
        constexpr double pc = 3.08567758e16;
        constexpr double Msun = 1.9891e30;
        constexpr double Lsun = 3.839e26;
        constexpr double Qelectron = 1.602176634e-19;
        constexpr double year = 31557600.;
    }

    struct UnitFactor
    {
        double factor{1.}, power{1.}, offset{0.};
    };

    class UnitTable
    {
    public:
        UnitTable()
        {
            using namespace constants;
            constexpr double arcsec = M_PI / (180. * 3600.);
            constexpr double arcsec2 = arcsec * arcsec;
            for (const char* q : {"length", "distance"})
            {
                add(q, "m", 1.);
                add(q, "cm", 1e-2);
                add(q, "mm", 1e-3);
                add(q, "km", 1e3);
                add(q, "AU", AU);
                add(q, "pc", pc);
                add(q, "kpc", 1e3 * pc);
                add(q, "Mpc", 1e6 * pc);
            }
            add("wavelength", "m", 1.);
            add("wavelength", "cm", 1e-2);
            add("wavelength", "mm", 1e-3);
            add("wavelength", "micron", 1e-6);
            add("wavelength", "nm", 1e-9);
            add("wavelength", "Angstrom", 1e-10);
            add("wavelength", "pm", 1e-12);
            add("wavelength", "Hz", c, -1.);
            add("wavelength", "GHz", 1e-9 * c, -1.);
            add("wavelength", "THz", 1e-12 * c, -1.);
            add("wavelength", "eV", h * c / Qelectron, -1.);
            add("wavelength", "keV", 1e-3 * (h * c) / Qelectron, -1.);
            add("velocity", "m/s", 1.);
            add("velocity", "cm/s", 1e-2);
            add("velocity", "km/s", 1e3);
            add("masscoefficient", "m2/kg", 1.);
            add("masscoefficient", "cm2/g", 0.1);
            add("mass", "kg", 1.);
            add("mass", "g", 1e-3);
            add("mass", "Msun", Msun);
            add("temperature", "K", 1.);
            add("magneticfield", "T", 1.);
            add("magneticfield", "G", 1e-4);
            add("magneticfield", "uG", 1e-10);
            add("magneticfield", "nG", 1e-13);
            add("bolluminosity", "W", 1.);
            add("bolluminosity", "J/s", 1.);
            add("bolluminosity", "erg/s", 1e-7);
            add("bolluminosity", "Lsun", Lsun);
            add("angle", "rad", 1.);
            add("angle", "deg", M_PI / 180.);
            add("angle", "arcsec", arcsec);
            add("posangle", "rad", 1.);
            add("posangle", "deg", M_PI / 180.);
            add("frequencyfluxdensity", "W/m2/Hz", 1.);
            add("frequencyfluxdensity", "Jy", 1e-26);
            add("frequencyfluxdensity", "mJy", 1e-29);
            add("frequencyfluxdensity", "MJy", 1e-20);
            add("frequencysurfacebrightness", "W/m2/Hz/sr", 1.);
            add("frequencysurfacebrightness", "Jy/sr", 1e-26);
            add("frequencysurfacebrightness", "MJy/sr", 1e-20);
            add("frequencysurfacebrightness", "Jy/arcsec2", 1e-26 / arcsec2);
            add("wavelengthfluxdensity", "W/m3", 1.);
            add("wavelengthfluxdensity", "W/m2/m", 1.);
            add("wavelengthfluxdensity", "W/m2/micron", 1e6);
            add("wavelengthsurfacebrightness", "W/m3/sr", 1.);
            add("wavelengthsurfacebrightness", "W/m2/m/sr", 1.);
            add("wavelengthsurfacebrightness", "W/m2/micron/sr", 1e6);
            add("wavelengthsurfacebrightness", "W/m2/micron/arcsec2", 1e6 / arcsec2);
            add("neutralfluxdensity", "W/m2", 1.);
            // specific luminosity per unit of wavelength (a tabulated SED; only the wavelength style is supported on this path)
            add("wavelengthmonluminosity", "W/m", 1.);
            add("wavelengthmonluminosity", "W/micron", 1e6);
            add("wavelengthmonluminosity", "W/Angstrom", 1e10);
            add("wavelengthmonluminosity", "erg/s/cm", 1e-5);
            add("wavelengthmonluminosity", "erg/s/micron", 1e-1);
            add("wavelengthmonluminosity", "erg/s/Angstrom", 1e3);
            add("wavelengthmonluminosity", "Lsun/micron", constants::Lsun * 1e6);
            add("neutralmonluminosity", "W", 1.);
            add("neutralmonluminosity", "erg/s", 1e-7);
            add("neutralmonluminosity", "Lsun", constants::Lsun);
            add("frequencymonluminosity", "W/Hz", 1.);
            add("frequencymonluminosity", "erg/s/Hz", 1e-7);
            add("frequencymonluminosity", "Lsun/Hz", constants::Lsun);
            add("neutralmeanintensity", "W/m2/sr", 1.);
            add("wavelengthmeanintensity", "W/m3/sr", 1.);
            add("wavelengthmeanintensity", "W/m2/micron/sr", 1e6);
            add("frequencymeanintensity", "W/m2/Hz/sr", 1.);
            add("frequencymeanintensity", "MJy/sr", 1e-20);
            add("neutralsurfacebrightness", "W/m2/sr", 1.);
            add("neutralsurfacebrightness", "W/m2/arcsec2", 1. / arcsec2);

            // default units per unit system (SkirtUnitDef.cpp:556-764)
            def("SIUnits", {{"neutralmeanintensity", "W/m2/sr"}, {"wavelengthmeanintensity", "W/m3/sr"},
                            {"frequencymeanintensity", "W/m2/Hz/sr"}, {"length", "m"}, {"distance", "m"}, {"wavelength", "m"}, {"velocity", "m/s"},
                            {"masscoefficient", "m2/kg"}, {"mass", "kg"}, {"temperature", "K"}, {"magneticfield", "T"},
                            {"bolluminosity", "W"}, {"angle", "rad"}, {"posangle", "rad"},
                            {"frequencyfluxdensity", "W/m2/Hz"}, {"frequencysurfacebrightness", "W/m2/Hz/sr"},
                            {"wavelengthfluxdensity", "W/m3"}, {"wavelengthsurfacebrightness", "W/m3/sr"},
                            {"neutralfluxdensity", "W/m2"}, {"neutralsurfacebrightness", "W/m2/sr"}});
            def("StellarUnits", {{"neutralmeanintensity", "W/m2/sr"}, {"wavelengthmeanintensity", "W/m2/micron/sr"},
                                 {"frequencymeanintensity", "W/m2/Hz/sr"}, {"length", "AU"}, {"distance", "pc"}, {"wavelength", "micron"}, {"velocity", "km/s"},
                                 {"masscoefficient", "m2/kg"}, {"mass", "Msun"}, {"temperature", "K"}, {"magneticfield", "uG"},
                                 {"bolluminosity", "Lsun"}, {"angle", "arcsec"}, {"posangle", "deg"},
                                 {"frequencyfluxdensity", "Jy"}, {"frequencysurfacebrightness", "MJy/sr"},
                                 {"wavelengthfluxdensity", "W/m2/micron"},
                                 {"wavelengthsurfacebrightness", "W/m2/micron/arcsec2"},
                                 {"neutralfluxdensity", "W/m2"}, {"neutralsurfacebrightness", "W/m2/arcsec2"}});
            def("ExtragalacticUnits",
                {{"neutralmeanintensity", "W/m2/sr"}, {"wavelengthmeanintensity", "W/m2/micron/sr"},
                 {"frequencymeanintensity", "W/m2/Hz/sr"},
                 {"length", "pc"}, {"distance", "Mpc"}, {"wavelength", "micron"}, {"velocity", "km/s"},
                 {"masscoefficient", "m2/kg"}, {"mass", "Msun"}, {"temperature", "K"}, {"magneticfield", "uG"}, {"bolluminosity", "Lsun"},
                 {"angle", "arcsec"}, {"posangle", "deg"}, {"frequencyfluxdensity", "Jy"},
                 {"frequencysurfacebrightness", "MJy/sr"}, {"wavelengthfluxdensity", "W/m2/micron"},
                 {"wavelengthsurfacebrightness", "W/m2/micron/arcsec2"}, {"neutralfluxdensity", "W/m2"},
                 {"neutralsurfacebrightness", "W/m2/arcsec2"}});
        }

        bool hasSystem(const std::string& system) const { return _systems.count(system) != 0; }

        const std::string& defaultUnit(const std::string& system, const std::string& qty) const
        {
            auto s = _systems.find(system);
            if (s == _systems.end()) throw std::runtime_error("ski: unknown unit system " + system);
            auto u = s->second.find(qty);
            if (u == s->second.end()) throw std::runtime_error("ski: no default unit for quantity " + qty);
            return u->second;
        }
        bool has(const std::string& qty, const std::string& unit) const
        {
            auto q = _quantities.find(qty);
            return q != _quantities.end() && q->second.count(unit) != 0;
        }
        const UnitFactor& factorOf(const std::string& qty, const std::string& unit) const { return find(qty, unit); }
        // UnitDef::in (UnitDef.cpp:64-80)
        double in(const std::string& qty, const std::string& unit, double value) const
        {
            const UnitFactor& f = find(qty, unit);
            if (f.power != 1.) value = pow(value, f.power);
            return f.factor * value + f.offset;
        }
        // UnitDef::out (UnitDef.cpp:84-101)
        double out(const std::string& qty, const std::string& unit, double value) const
        {
            const UnitFactor& f = find(qty, unit);
            value = (value - f.offset) / f.factor;
            if (f.power != 1.) value = pow(value, 1. / f.power);
            return value;
        }

    private:
        std::map<std::string, std::map<std::string, UnitFactor>> _quantities;
        std::map<std::string, std::map<std::string, std::string>> _systems;

        void add(const std::string& qty, const std::string& unit, double factor, double power = 1.)
        {
            _quantities[qty][unit] = UnitFactor{factor, power, 0.};
        }
        void def(const std::string& system, std::initializer_list<std::pair<const char*, const char*>> list)
        {
            for (auto& p : list) _systems[system][p.first] = p.second;
        }
        const UnitFactor& find(const std::string& qty, const std::string& unit) const
        {
            auto q = _quantities.find(qty);
            if (q != _quantities.end())
            {
                auto u = q->second.find(unit);
                if (u != q->second.end()) return u->second;
            }
            throw std::runtime_error("ski: unknown quantity '" + qty + "' and/or unit '" + unit + "'");
        }
    };

    inline const UnitTable& unitTable()
    {
        static const UnitTable table;
        return table;
    }

    // ---- string helpers (whitespace handling as StringUtils::squeeze / split) ----

    inline std::string squeeze(const std::string& s)
    {
        std::string out;
        bool space = false;
        for (char ch : s)
        {
            if (ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r')
                space = true;
            else
            {
                if (space && !out.empty()) out += ' ';
                space = false;
                out += ch;
            }
        }
        return out;
    }

    inline std::vector<std::string> splitOn(const std::string& s, char sep)
    {
        std::vector<std::string> parts;
        size_t b = 0;
        while (true)
        {
            size_t e = s.find(sep, b);
            parts.push_back(s.substr(b, e == std::string::npos ? std::string::npos : e - b));
            if (e == std::string::npos) break;
            b = e + 1;
        }
        return parts;
    }

    // "<number> [unit]" -> SI double (AbstractDoublePropertyHandler::toDouble, AbstractDoublePropertyHandler.cpp:136-167)
    inline double parseQuantity(const std::string& text, const std::string& qty, const std::string& unitSystem)
    {
        std::string value = squeeze(text);
        auto segments = splitOn(value, ' ');
        if (segments.empty() || segments[0].empty() || segments.size() > 2)
            throw std::runtime_error("ski: invalid numeric value '" + text + "'");
        char* end = nullptr;
        double result = std::strtod(segments[0].c_str(), &end);
        if (end != segments[0].c_str() + segments[0].size())
            throw std::runtime_error("ski: invalid numeric value '" + text + "'");
        if (!qty.empty())
        {
            std::string unit = segments.size() == 2 ? segments[1] : unitTable().defaultUnit(unitSystem, qty);
            result = unitTable().in(qty, unit, result);
        }
        else if (segments.size() == 2)
            throw std::runtime_error("ski: unexpected unit in dimensionless value '" + text + "'");
        return result;
    }

    inline std::vector<double> parseQuantityList(const std::string& text, const std::string& qty,
                                                 const std::string& unitSystem)
    {
        std::vector<double> result;
        for (auto& part : splitOn(text, ','))
            if (!squeeze(part).empty()) result.push_back(parseQuantity(part, qty, unitSystem));
        return result;
    }
}

#endif
