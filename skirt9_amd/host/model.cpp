// model.cpp -- implementation of the host model layer (see model.hpp).  Citations are to the SKIRT 9 tree.

#include "model.hpp"
#include "units.hpp"
#include <cstring>
#include <functional>
#include <stdexcept>

namespace skh
{
    // ================================================================ OutputUnits (Units.cpp:76-79,111-160,529-609)

    double OutputUnits::out(const std::string& qty, double value) const
    {
        return unitTable().out(qty, unitTable().defaultUnit(system, qty), value);
    }
    std::string OutputUnits::unit(const std::string& qty) const { return unitTable().defaultUnit(system, qty); }

    namespace
    {
        std::string stylePrefix(const std::string& fluxStyle)
        {
            if (fluxStyle == "Neutral") return "neutral";
            if (fluxStyle == "Wavelength") return "wavelength";
            if (fluxStyle == "Frequency") return "frequency";
            throw std::runtime_error("ski: fluxOutputStyle '" + fluxStyle + "' is not supported on the MI355X path");
        }
    }
    std::string OutputUnits::sfluxdensity() const
    {
        if (fluxStyle == "Neutral") return "lambda*F_lambda";
        if (fluxStyle == "Wavelength") return "F_lambda";
        return "F_nu";
    }
    std::string OutputUnits::ufluxdensity() const { return unit(stylePrefix(fluxStyle) + "fluxdensity"); }
    std::string OutputUnits::usurfacebrightness() const { return unit(stylePrefix(fluxStyle) + "surfacebrightness"); }
    double OutputUnits::ofluxdensity(double lambda, double Flambda) const
    {
        if (fluxStyle == "Neutral") return out("neutralfluxdensity", lambda * Flambda);
        if (fluxStyle == "Wavelength") return out("wavelengthfluxdensity", Flambda);
        return out("frequencyfluxdensity", lambda * lambda * Flambda / constants::c);
    }
    std::string OutputUnits::smeanintensity() const
    {
        if (fluxStyle == "Neutral") return "lambda*J_lambda";
        if (fluxStyle == "Wavelength") return "J_lambda";
        return "J_nu";
    }
    std::string OutputUnits::umeanintensity() const { return unit(stylePrefix(fluxStyle) + "meanintensity"); }
    double OutputUnits::omeanintensity(double lambda, double Jlambda) const
    {
        if (fluxStyle == "Neutral") return out("neutralmeanintensity", lambda * Jlambda);
        if (fluxStyle == "Wavelength") return out("wavelengthmeanintensity", Jlambda);
        return out("frequencymeanintensity", lambda * lambda * Jlambda / constants::c);
    }
    double OutputUnits::osurfacebrightness(double lambda, double flambda) const
    {
        if (fluxStyle == "Neutral") return out("neutralsurfacebrightness", lambda * flambda);
        if (fluxStyle == "Wavelength") return out("wavelengthsurfacebrightness", flambda);
        return out("frequencysurfacebrightness", lambda * lambda * flambda / constants::c);
    }

    // ================================================================ SersicFunction (SersicFunction.cpp:13-101)

    SersicFunction::SersicFunction(double n)
    {
        if (n < 0.5 || n > 10.0) throw std::runtime_error("The Sersic parameter should be between 0.5 and 10");
        double b = 2.0 * n - 1.0 / 3.0 + 4.0 / 405.0 / n + 46.0 / 25515.0 / (n * n) + 131.0 / 1148175.0 / (n * n * n);
        double I0 = pow(b, 2.0 * n) / (M_PI * special::gamma(2.0 * n + 1));
        int Ns = 101;
        _sv.assign(Ns, 0.);
        _Sv.assign(Ns, 0.);
        _Mv.assign(Ns, 0.);
        double logsmin = -6.0;
        double logsmax = 4.0;
        double dlogs = (logsmax - logsmin) / (Ns - 1.0);
        for (int i = 0; i < Ns; i++)
        {
            double logs = logsmin + i * dlogs;
            double s = pow(10.0, logs);
            _sv[i] = s;
            double alpha = b * pow(s, 1.0 / n);
            double sum = 0.0;
            int Nu = 10000;
            double tmax = 100.0;
            double umax = sqrt((tmax + 1.0) * (tmax - 1.0));
            double du = umax / Nu;
            for (int j = 0; j <= Nu; j++)
            {
                double weight = 1.0;
                if (j == 0 || j == Nu) weight = 0.5;
                double u = j * du;
                double u2 = u * u;
                double w;
                if (u > 1e-3)
                    w = (pow(1.0 + u2, 2.0 * n) - 1.0) / u2;
                else
                    w = 2.0 * n + n * (2.0 * n - 1.0) * u2 + 2.0 / 3.0 * n * (2.0 * n - 1.0) * (n - 1.0) * u2 * u2;
                double integrandum = 2.0 * exp(-alpha * (1.0 + u2)) / sqrt(w);
                sum += weight * integrandum;
            }
            _Sv[i] = I0 * pow(b, n) * pow(alpha, 1.0 - n) / M_PI * du * sum;
        }
        for (int i = 1; i < Ns; i++)
        {
            double sum = 0.0;
            for (int j = 0; j <= 32; j++)
            {
                double weight = 1.0;
                if (j == 0 || j == 32) weight = 0.5;
                double ds = (_sv[i] - _sv[i - 1]) / 32.0;
                double s = _sv[i - 1] + j * ds;
                double S = operator()(s);
                sum += weight * S * s * s * ds;
            }
            double dM = 4.0 * M_PI * sum;
            _Mv[i] = _Mv[i - 1] + dM;
        }
        double last = _Mv[Ns - 1];
        for (int i = 0; i < Ns; i++) _Mv[i] /= last;
    }
    double SersicFunction::operator()(double s) const { return nr::clampedValue<nr::interpolateLogLog>(s, _sv, _Sv); }
    double SersicFunction::inverseMass(double M) const { return nr::clampedValue<nr::interpolateLogLog>(M, _Mv, _sv); }

    // ================================================================ geometries

    // UniformBoxGeometry.cpp:12-61
    UniformBoxGeometry::UniformBoxGeometry(const Box& box) : _box(box)
    {
        if (box.xmax - box.xmin <= 0 || box.ymax - box.ymin <= 0 || box.zmax - box.zmin <= 0)
            throw std::runtime_error("The extent of the box should be positive in every direction");
        _rho = 1. / _box.volume();
    }
    double UniformBoxGeometry::density(Vec3 r) const { return _box.contains(r.x, r.y, r.z) ? _rho : 0.; }
    double UniformBoxGeometry::SigmaX() const { return 1. / ((_box.ymax - _box.ymin) * (_box.zmax - _box.zmin)); }
    double UniformBoxGeometry::SigmaY() const { return 1. / ((_box.xmax - _box.xmin) * (_box.zmax - _box.zmin)); }
    double UniformBoxGeometry::SigmaZ() const { return 1. / ((_box.xmax - _box.xmin) * (_box.ymax - _box.ymin)); }

    // ExpDiskGeometry.cpp:13-42,72-90
    ExpDiskGeometry::ExpDiskGeometry(double hR, double hz, double Rmin, double Rmax, double zmax)
        : _hR(hR), _hz(hz), _Rmin(Rmin), _Rmax(Rmax), _zmax(zmax)
    {
        if (_Rmin >= _Rmax && _Rmax > 0)
            throw std::runtime_error("The radius of the central cavity should be smaller than the truncation radius");
        double intphi = 2.0 * M_PI;
        double intz = (_zmax > 0) ? -2.0 * _hz * expm1(-_zmax / _hz) : 2.0 * _hz;
        double tmin = (_Rmin > 0) ? exp(-_Rmin / _hR) * (1.0 + _Rmin / _hR) : 1.0;
        double tmax = (_Rmax > 0) ? exp(-_Rmax / _hR) * (1.0 + _Rmax / _hR) : 0.0;
        double intR = _hR * _hR * (tmin - tmax);
        _rho0 = 1.0 / (intR * intphi * intz);
    }
    double ExpDiskGeometry::density(Vec3 r) const
    {
        double R = sqrt(r.x * r.x + r.y * r.y);  // Position::cylRadius
        double absz = fabs(r.z);
        if (_Rmax > 0.0 && R > _Rmax) return 0.0;
        if (_zmax > 0.0 && absz > _zmax) return 0.0;
        if (R < _Rmin) return 0.0;
        return _rho0 * exp(-R / _hR) * exp(-absz / _hz);
    }
    double ExpDiskGeometry::SigmaR() const
    {
        if (_Rmax > 0.0) return _rho0 * _hR * (exp(-_Rmin / _hR) - exp(-_Rmax / _hR));
        return _rho0 * _hR * exp(-_Rmin / _hR);
    }
    double ExpDiskGeometry::SigmaZ() const
    {
        if (_Rmin > 0.0) return 0.0;
        if (_zmax > 0.0) return -2.0 * _rho0 * _hz * expm1(-_zmax / _hz);
        return 2.0 * _rho0 * _hz;
    }

    namespace
    {
        // SpecialFunctions::LambertW1 (SKIRT/utils/SpecialFunctions.cpp:578-625)
        double lambertW1(double z)
        {
            const double eps = 1.0e-12;
            const double em1 = 0.3678794411714423215955237701614608;
            static const double c[12] = {-1.0,
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
            if (z < -em1 || z > 0.0 || std::isinf(z) || std::isnan(z)) throw std::runtime_error("LambertW1: bad argument");
            if (z == 0.0) return -DBL_MAX;
            double q = z + em1;
            double r = -sqrt(q);
            double t8 = c[8] + r * (c[9] + r * (c[10] + r * c[11]));
            double t5 = c[5] + r * (c[6] + r * (c[7] + r * t8));
            double t1 = c[1] + r * (c[2] + r * (c[3] + r * (c[4] + r * t5)));
            double w0 = c[0] + r * t1;
            if (q < 3.0e-3) return w0;
            double w, e, p, t;
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
                e = exp(w);
                t = w * e - z;
                p = w + 1.0;
                t /= e * p - 0.5 * (p + 1.0) * t / p;
                w -= t;
                if (fabs(t) < eps * (1.0 + fabs(w))) return w;
            }
            throw std::runtime_error("LambertW1: no convergence");
        }
    }
    Vec3 ExpDiskGeometry::generatePosition(Random& random) const
    {
        double R, X;
        do
        {
            X = random.uniform();
            R = _hR * (-1.0 - lambertW1((X - 1.0) / M_E));
        } while ((_Rmax > 0.0 && R >= _Rmax) || R <= _Rmin);
        double phi = 2.0 * M_PI * random.uniform();
        double z;
        do
        {
            X = random.uniform();
            z = (X <= 0.5) ? _hz * log(2.0 * X) : -_hz * log(2.0 * (1.0 - X));
        } while (_zmax > 0.0 && fabs(z) >= _zmax);
        return Vec3{R * cos(phi), R * sin(phi), z};  // Position(R, phi, z, CYLINDRICAL)
    }

    // SersicGeometry.cpp:21-52
    SersicGeometry::SersicGeometry(double reff, double n) : _reff(reff), _n(n)
    {
        _rho0 = 1.0 / (_reff * _reff * _reff);
        _b = 2.0 * _n - 1.0 / 3.0 + 4.0 / 405.0 / _n + 46.0 / 25515.0 / (_n * _n) + 131.0 / 1148175.0 / (_n * _n * _n);
        _function = std::make_unique<SersicFunction>(_n);
    }
    double SersicGeometry::density(Vec3 r) const
    {
        double radius = sqrt(r.x * r.x + r.y * r.y + r.z * r.z);  // Vec::norm
        double s = radius / _reff;
        return _rho0 * (*_function)(s);
    }
    double SersicGeometry::Sigmar() const
    {
        return 1.0 / (_reff * _reff) * pow(_b, 2.0 * _n) / (2.0 * M_PI * special::gamma(2.0 * _n + 1.0));
    }

    Vec3 SersicGeometry::generatePosition(Random& random) const
    {
        double r = _reff * _function->inverseMass(random.uniform());  // SersicGeometry::randomRadius
        Vec3 k = random.direction();
        return Vec3{r * k.x, r * k.y, r * k.z};  // Position(r, bfk)
    }

    // PlummerGeometry.cpp:12-38
    PlummerGeometry::PlummerGeometry(double c) : _c(c) { _rho0 = 0.75 / pow(_c, 3) / M_PI; }
    double PlummerGeometry::density(Vec3 r) const
    {
        double radius = sqrt(r.x * r.x + r.y * r.y + r.z * r.z);
        double s = radius / _c;
        return _rho0 * pow(1.0 + s * s, -2.5);
    }
    double PlummerGeometry::Sigmar() const { return 0.5 / (M_PI * _c * _c); }
    Vec3 PlummerGeometry::generatePosition(Random& random) const
    {
        double t = pow(random.uniform(), 1.0 / 3.0);
        double r = _c * t / sqrt((1.0 - t) * (1.0 + t));
        Vec3 k = random.direction();
        return Vec3{r * k.x, r * k.y, r * k.z};
    }

    // ================================================================ SpheroidalGeometryDecorator

    double SpheroidalGeometry::density(Vec3 bfr) const
    {
        // AxGeometry::density(Position) passes (cylindrical radius, height); the spherical density is evaluated at radius
        // m (a position (m, 0, 0): sqrt(m*m) == m in IEEE arithmetic)
        double R = sqrt(bfr.x * bfr.x + bfr.y * bfr.y), z = bfr.z;
        double m = sqrt(R * R + z * z / (_q * _q));
        return 1.0 / _q * _inner->density(Vec3{m, 0., 0.});
    }
    Vec3 SpheroidalGeometry::generatePosition(Random& random) const
    {
        Vec3 s = _inner->generatePosition(random);
        return Vec3{s.x, s.y, _q * s.z};
    }

    // ================================================================ ShellGeometry, TorusGeometry, RingGeometry

    namespace
    {
        // SpecialFunctions::gln2 (SpecialFunctions.cpp:815-818)
        double gln2(double p, double x1, double x2) { return pow(x2, 1.0 - p) * special::gln(p, x1 / x2); }
    }

    ShellGeometry::ShellGeometry(double rmin, double rmax, double p) : _rmin(rmin), _rmax(rmax), _p(p)
    {
        if (_rmax <= _rmin) throw std::runtime_error("the outer radius of the shell should be larger than the inner radius");
        _smin = special::gln(_p - 2.0, _rmin);
        _sdiff = gln2(_p - 2.0, _rmax, _rmin);
        _tmin = pow(_rmin, 3.0 - _p);
        _tmax = pow(_rmax, 3.0 - _p);
        _A = 0.25 / M_PI / _sdiff;
    }
    double ShellGeometry::density(Vec3 bfr) const
    {
        double r = sqrt(bfr.x * bfr.x + bfr.y * bfr.y + bfr.z * bfr.z);
        if (r < _rmin || r > _rmax) return 0.0;
        return _A * pow(r, -_p);
    }
    double ShellGeometry::Sigmar() const { return _A * gln2(_p, _rmax, _rmin); }
    Vec3 ShellGeometry::generatePosition(Random& random) const
    {
        double X = random.uniform();
        double r;
        if (fabs(_p - 3.0) < 1e-2)
            r = special::gexp(_p - 2.0, _smin + X * _sdiff);
        else
            r = pow((1.0 - X) * _tmin + X * _tmax, 1.0 / (3.0 - _p));
        Vec3 k = random.direction();
        return Vec3{r * k.x, r * k.y, r * k.z};
    }

    TorusGeometry::TorusGeometry(double p, double q, double Delta, double rmin, double rmax, bool rani, double rcut)
        : _p(p), _q(q), _Delta(Delta), _rmin(rmin), _rmax(rmax), _rani(rani), _rcut(rcut)
    {
        _sinDelta = sin(_Delta);
        _smin = special::gln(_p - 2.0, _rmin);
        _sdiff = gln2(_p - 2.0, _rmax, _rmin);
        _tmin = pow(_rmin, 3.0 - _p);
        _tmax = pow(_rmax, 3.0 - _p);
        if (_q > 1e-3)
            _A = _q * 0.25 / M_PI / _sdiff / (1.0 - exp(-_q * _sinDelta));
        else
            _A = 0.25 / M_PI / _sdiff / _sinDelta;
    }
    double TorusGeometry::density(Vec3 bfr) const
    {
        // AxGeometry::density(Position) passes the cylindrical radius and the height (AxGeometry.cpp:11-17)
        double R = sqrt(bfr.x * bfr.x + bfr.y * bfr.y), z = bfr.z;
        double r = sqrt(R * R + z * z);
        double costheta = z / r;
        if (r >= _rmax) return 0.0;
        if (_rani)
        {
            double rminani = _rmin * sqrt(6. / 7. * fabs(costheta) * (2. * fabs(costheta) + 1));
            if (r <= rminani || r < _rcut) return 0.0;
        }
        else
        {
            if (r <= _rmin) return 0.0;
        }
        if (fabs(costheta) >= _sinDelta) return 0.0;
        return _A * pow(r, -_p) * exp(-_q * fabs(costheta));
    }
    double TorusGeometry::SigmaR() const { return _A * gln2(_p, _rmax, _rmin); }
    Vec3 TorusGeometry::generatePosition(Random& random) const
    {
        while (true)
        {
            double X = random.uniform();
            double r;
            if (fabs(_p - 3.0) < 1e-2)
                r = special::gexp(_p - 2.0, _smin + X * _sdiff);
            else
                r = pow((1.0 - X) * _tmin + X * _tmax, 1.0 / (3.0 - _p));
            X = random.uniform();
            double costheta;
            if (_q < 1e-3)
                costheta = (1.0 - 2.0 * X) * _sinDelta;
            else
            {
                double B = 1.0 - exp(-_q * _sinDelta);
                costheta = (X < 0.5) ? -log(1.0 - B * (1.0 - 2.0 * X)) / _q : log(1.0 - B * (2.0 * X - 1.0)) / _q;
            }
            double theta = acos(costheta);
            double phi = 2.0 * M_PI * random.uniform();
            // Position(r, theta, phi, SPHERICAL) (Vec/Position.cpp)
            double sintheta = sin(theta);
            Vec3 pos{r * sintheta * cos(phi), r * sintheta * sin(phi), r * cos(theta)};
            if (density(pos)) return pos;
        }
    }

    RingGeometry::RingGeometry(double R0, double w, double hz) : _R0(R0), _w(w), _hz(hz)
    {
        double t = _R0 / _w / M_SQRT2;
        double intz = 2.0 * _hz;
        double intR = _w * _w * (exp(-t * t) + sqrt(M_PI) * t * (1.0 + erf(t)));
        _A = 1.0 / (2.0 * M_PI * intz * intR);
        int NR = 330;
        nr::linearGrid(_Rv, std::max(0., _R0 - 8 * _w), _R0 + 8 * _w, NR - 1);
        _Xv.resize(NR);
        double sqrtpi = sqrt(M_PI);
        for (int i = 0; i < NR; i++)
        {
            double R = _Rv[i];
            double u = (_R0 - R) / _w / M_SQRT2;
            _Xv[i] = 4.0 * M_PI * _A * _hz * _w * _w * (exp(-t * t) - exp(-u * u) + sqrtpi * t * (erf(t) - erf(u)));
        }
        _Xv[0] = 0.0;
        _Xv[NR - 1] = 1.0;
    }
    double RingGeometry::density(Vec3 bfr) const
    {
        double R = sqrt(bfr.x * bfr.x + bfr.y * bfr.y), z = bfr.z;
        double u = (R - _R0) / (M_SQRT2 * _w);
        return _A * exp(-u * u) * exp(-fabs(z) / _hz);
    }
    double RingGeometry::SigmaR() const
    {
        double t = _R0 / (M_SQRT2 * _w);
        return sqrt(M_PI / 2.0) * _A * _w * (1.0 + erf(t));
    }
    double RingGeometry::SigmaZ() const
    {
        double t = _R0 / (M_SQRT2 * _w);
        return 2.0 * _A * _hz * exp(-t * t);
    }
    Vec3 RingGeometry::generatePosition(Random& random) const
    {
        // Random::cdfLinLin (Random.cpp:190-195) on the tabulated radial distribution, then phi, then z
        double X = random.uniform();
        int i = nr::locateClip(_Xv, X);
        double R = nr::interpolateLinLin(X, _Xv[i], _Xv[i + 1], _Rv[i], _Rv[i + 1]);
        double phi = 2.0 * M_PI * random.uniform();
        X = random.uniform();
        double z = (X <= 0.5) ? _hz * log(2.0 * X) : -_hz * log(2.0 * (1.0 - X));
        return Vec3{R * cos(phi), R * sin(phi), z};
    }

    // ================================================================ DustMix (DustMix.cpp:47-162)

    void DustMix::setup(double rangeMin, double rangeMax, const std::vector<double>& simulationWavelengths)
    {
        if (inLambda.size() != inKappaExt.size() || inLambda.size() != inAlbedo.size()
            || inLambda.size() != inAsymmpar.size())
            throw std::runtime_error("Number of listed properties does not match number of listed wavelengths");
        if (inLambda.empty()) throw std::runtime_error("Dust properties must be tabulated for at least one wavelength");

        // fine grid at integer multiples of 1/1000 dex + all configured wavelengths (DustMix.cpp:57-72)
        std::vector<double> wavelengths;
        const double numWavelengthsPerDex = 1000;
        int minLambdaSerial = std::floor(numWavelengthsPerDex * log10(rangeMin));
        int maxLambdaSerial = std::ceil(numWavelengthsPerDex * log10(rangeMax));
        for (int k = minLambdaSerial; k <= maxLambdaSerial; ++k) wavelengths.push_back(pow(10., k / numWavelengthsPerDex));
        for (double lambda : simulationWavelengths) wavelengths.push_back(lambda);
        std::sort(wavelengths.begin(), wavelengths.end());
        wavelengths.erase(std::unique(wavelengths.begin(), wavelengths.end()), wavelengths.end());

        // radio cutoff beyond 10 cm (DustMix.cpp:74-82)
        const double dm = 0.1;
        bool radioCutoff = wavelengths.back() > dm;
        if (radioCutoff)
        {
            wavelengths.resize(nr::locate(wavelengths, dm) + 1);
            if (wavelengths.empty() || wavelengths.back() != dm) wavelengths.push_back(dm);
            wavelengths.push_back(dm * 1.001);
        }
        lambdaSample = wavelengths;
        int numLambda = static_cast<int>(lambdaSample.size());

        // index grid shifted to the left of the sample points (DustMix.cpp:93-98)
        lambdaBorder.assign(numLambda, 0.);
        lambdaBorder[0] = lambdaSample[0];
        for (int ell = 1; ell != numLambda; ++ell) lambdaBorder[ell] = sqrt(lambdaSample[ell] * lambdaSample[ell - 1]);

        // TabulatedDustMix::getOpticalProperties (TabulatedDustMix.cpp:12-45)
        Array inl = inLambda, ink = inKappaExt, ina = inAlbedo, ing = inAsymmpar;
        if (inl.size() > 1 && inl[0] > inl[inl.size() - 1])
        {
            std::reverse(inl.begin(), inl.end());
            std::reverse(ink.begin(), ink.end());
            std::reverse(ina.begin(), ina.end());
            std::reverse(ing.begin(), ing.end());
        }
        Array insigmaabs(inl.size()), insigmasca(inl.size());
        for (size_t i = 0; i < inl.size(); ++i)
        {
            insigmaabs[i] = mu * ink[i] * (1. - ina[i]);
            insigmasca[i] = mu * ink[i] * ina[i];
        }
        sigmaAbs = nr::clampedResample<nr::interpolateLogLog>(lambdaSample, inl, insigmaabs);
        sigmaSca = nr::clampedResample<nr::interpolateLogLog>(lambdaSample, inl, insigmasca);
        asymmpar = nr::clampedResample<nr::interpolateLogLin>(lambdaSample, inl, ing);

        // clamp g (DustMix.cpp:139-146), derive extinction (:160-162)
        const double gmax = 0.999999;
        for (int ell = 0; ell != numLambda; ++ell)
            if (std::abs(asymmpar[ell]) > gmax) asymmpar[ell] = std::copysign(gmax, asymmpar[ell]);
        sigmaExt.assign(numLambda, 0.);
        for (int ell = 0; ell != numLambda; ++ell) sigmaExt[ell] = sigmaAbs[ell] + sigmaSca[ell];
        if (radioCutoff)
        {
            sigmaAbs[numLambda - 2] = sigmaAbs[numLambda - 1] = 0.;
            sigmaSca[numLambda - 2] = sigmaSca[numLambda - 1] = 0.;
            sigmaExt[numLambda - 2] = sigmaExt[numLambda - 1] = 0.;
        }
    }

    // ================================================================ GeometricMedium

    void GeometricMedium::setup()
    {
        if (normType == "OpticalDepthMaterialNormalization")
        {
            // AxisMaterialNormalization.cpp:11-25, OpticalDepthMaterialNormalization.cpp:13-27
            double geomColumnDensity = normAxis == 'X' ? geometry->SigmaX() : normAxis == 'Y' ? geometry->SigmaY() : geometry->SigmaZ();
            if (geomColumnDensity <= 0.)
                throw std::runtime_error("Can't normalize material for geometry with zero column density along selected axis");
            double section = mix->sectionExt(normWavelength);
            if (section <= 0.) throw std::runtime_error("Can't normalize optical depth for material with zero extinction");
            double reqNumberColumnDensity = normOpticalDepth / section;
            double reqMassColumnDensity = reqNumberColumnDensity * mix->mass();
            number = reqNumberColumnDensity / geomColumnDensity;
            mass = reqMassColumnDensity / geomColumnDensity;
        }
        else if (normType == "MassMaterialNormalization")
        {
            number = normMass / mix->mass();
            mass = normMass;
        }
        else if (normType == "NumberMaterialNormalization")
        {
            number = normNumber;
            mass = normNumber * mix->mass();
        }
        else
            throw std::runtime_error("ski: material normalization '" + normType + "' is not supported on the MI355X path");
    }

    // ================================================================ CartesianSpatialGrid

    void CartesianSpatialGrid::setup()
    {
        // Mesh::mesh of the configured class on [0,1] (LinMesh.cpp:11-16, PowMesh.cpp:11-19, SymPowMesh.cpp:11-19,
        // LogMesh.cpp:11-19, SymLogMesh.cpp:11-42 with the grid builders of NR.hpp:203-320), scaled as
        // CartesianSpatialGrid.cpp:22-24
        auto meshOf = [](const MeshSpec& spec, int n) {
            Array tv;
            if (spec.type == "ListMesh")
            {
                tv.resize(spec.points.size());
                for (size_t i = 0; i < spec.points.size(); ++i) tv[i] = spec.points[i];
                return tv;
            }
            if (spec.type == "PowMesh" && n > 1)
            {
                // NR::buildPowerLawGrid
                if (fabs(spec.ratio - 1.) < 1e-3)
                    nr::linearGrid(tv, 0.0, 1.0, n);
                else
                {
                    tv.resize(n + 1);
                    double range = 1.0 - 0.0;
                    double q = pow(spec.ratio, 1. / (n - 1));
                    double qn = pow(q, n);
                    for (int i = 0; i <= n; ++i) tv[i] = 0.0 + (1. - pow(q, i)) / (1. - qn) * range;
                }
            }
            else if (spec.type == "SymPowMesh" && n > 2)
            {
                // NR::buildSymmetricPowerLawGrid
                if (fabs(spec.ratio - 1.) < 1e-3)
                    nr::linearGrid(tv, 0.0, 1.0, n);
                else
                {
                    tv.resize(n + 1);
                    const double xmin = 0.0, xmax = 1.0;
                    double xc = 0.5 * (xmin + xmax);
                    if (n % 2 == 0)
                    {
                        int M = n / 2;
                        double q = pow(spec.ratio, 1.0 / (M - 1.0));
                        double qM = pow(q, M);
                        tv[M] = xc;
                        for (int i = 1; i <= M; ++i)
                        {
                            double dxi = (1.0 - pow(q, i)) / (1.0 - qM) * 0.5 * (xmax - xmin);
                            tv[M + i] = xc + dxi;
                            tv[M - i] = xc - dxi;
                        }
                    }
                    else
                    {
                        int M = (n + 1) / 2;
                        double q = pow(spec.ratio, 1.0 / (M - 1.0));
                        double qM = pow(q, M);
                        for (int i = 1; i <= M; ++i)
                        {
                            double dxi = (0.5 + 0.5 * q - pow(q, i)) / (0.5 + 0.5 * q - qM) * 0.5 * (xmax - xmin);
                            tv[M - 1 + i] = xc + dxi;
                            tv[M - i] = xc - dxi;
                        }
                    }
                }
            }
            else if (spec.type == "LogMesh" && n > 1)
            {
                // NR::buildZeroLogGrid(tv, centralBinFraction, 1, n)
                tv.assign(n + 1, 0.);
                double logxmin = log(spec.centralBinFraction);
                double dlogx = log(1.0 / spec.centralBinFraction) / (n - 1);
                for (int i = 0; i < n; i++) tv[i + 1] = exp(logxmin + i * dlogx);
            }
            else if (spec.type == "SymLogMesh" && n > 2)
            {
                // the rightmost half as NR::buildLogGrid(tmpv, centralBinFraction, 1, n2), mirrored
                int n2 = (n - 1) / 2;
                Array tmpv(n2 + 1);
                double logxmin = log(spec.centralBinFraction);
                double dlogx = log(1. / spec.centralBinFraction) / n2;
                for (int i = 0; i <= n2; i++) tmpv[i] = exp(logxmin + i * dlogx);
                tv.assign(n + 1, 0.);
                int k = 0;
                tv[k++] = 0.;
                for (int i = n2 - 1; i >= 0; --i) tv[k++] = 0.5 - 0.5 * tmpv[i];
                if (n % 2 == 0) tv[k++] = 0.5;
                for (int i = 0; i <= n2 - 1; ++i) tv[k++] = 0.5 + 0.5 * tmpv[i];
                tv[k++] = 1.;
            }
            else
                nr::linearGrid(tv, 0.0, 1.0, (spec.type == "PowMesh" || spec.type == "LogMesh") ? 1 : n);
            return tv;
        };
        const MeshSpec* specs = meshSpec;
        int axis = 0;
        auto build = [&](Array& v, int n, double lo, double hi) {
            Array tv = meshOf(specs[axis++], n);
            v.resize(n + 1);
            for (int i = 0; i <= n; ++i) v[i] = tv[i] * (hi - lo) + lo;
        };
        build(xv, nx, extent.xmin, extent.xmax);
        build(yv, ny, extent.ymin, extent.ymax);
        build(zv, nz, extent.zmin, extent.zmax);
    }
    Box CartesianSpatialGrid::cellBox(int m) const
    {
        int i = m / (nz * ny);
        int j = (m / nz) % ny;
        int k = m % nz;
        return Box(xv[i], yv[j], zv[k], xv[i + 1], yv[j + 1], zv[k + 1]);
    }
    void CartesianSpatialGrid::fill(pmc_grid& g) const
    {
        g.kind = PMC_GRID_CARTESIAN;
        g.nx = nx;
        g.ny = ny;
        g.nz = nz;
        g.xv = xv.data();
        g.yv = yv.data();
        g.zv = zv.data();
    }

    // ================================================================ OctreeSpatialGrid

    namespace
    {
        const int complementWall[6] = {PMC_WALL_FRONT, PMC_WALL_BACK, PMC_WALL_RIGHT, PMC_WALL_LEFT, PMC_WALL_TOP,
                                       PMC_WALL_BOTTOM};

        inline void makeNeighbors(std::vector<OctreeSpatialGrid::Node>& nodes, int wall1, int node1, int node2)
        {
            nodes[node1].neighbors[wall1].push_back(node2);
            nodes[node2].neighbors[complementWall[wall1]].push_back(node1);
        }
        inline void deleteNeighbor(std::vector<OctreeSpatialGrid::Node>& nodes, int node, int wall, int other)
        {
            auto& list = nodes[node].neighbors[wall];
            for (size_t i = 0; i < list.size(); ++i)
                if (list[i] == other)
                {
                    list.erase(list.begin() + i);
                    break;
                }
        }
        inline double lo(const Box& b, int axis) { return axis == 0 ? b.xmin : axis == 1 ? b.ymin : b.zmin; }
        inline double hi(const Box& b, int axis) { return axis == 0 ? b.xmax : axis == 1 ? b.ymax : b.zmax; }
    }

    // TreeNode::subdivide (TreeNode.cpp:78-83) = OctTreeNode::createChildren (OctTreeNode.cpp:22-33)
    //                                           + OctTreeNode::addNeighbors (OctTreeNode.cpp:45-138)
    void OctreeSpatialGrid::subdivide(int id)
    {
        int first = static_cast<int>(nodes.size());
        {
            const Box b = nodes[id].box;
            Vec3 rc = b.center();
            int level = nodes[id].level + 1;
            for (int l = 0; l < 8; ++l)
            {
                Node child;
                child.box = Box((l & 1) ? rc.x : b.xmin, (l & 2) ? rc.y : b.ymin, (l & 4) ? rc.z : b.zmin,
                                (l & 1) ? b.xmax : rc.x, (l & 2) ? b.ymax : rc.y, (l & 4) ? b.zmax : rc.z);
                child.level = level;
                child.parent = id;
                nodes.push_back(std::move(child));
            }
            nodes[id].firstChild = first;
        }
        // internal neighbours among the siblings, in the reference's call order
        for (int l = 0; l < 8; ++l)
        {
            if (!(l & 1)) makeNeighbors(nodes, PMC_WALL_FRONT, first + l, first + l + 1);
            if (!(l & 2)) makeNeighbors(nodes, PMC_WALL_RIGHT, first + l, first + l + 2);
            if (!(l & 4)) makeNeighbors(nodes, PMC_WALL_TOP, first + l, first + l + 4);
        }
        // hand the parent's outer neighbours to the children that touch them
        double split[3] = {nodes[first].box.xmax, nodes[first].box.ymax, nodes[first].box.zmax};
        for (int wall = 0; wall < 6; ++wall)
        {
            int axis = wall >> 1;    // axis perpendicular to the wall
            int side = wall & 1;     // 0: lower wall, 1: upper wall
            int a1 = axis == 0 ? 1 : 0;               // first transverse axis (x before y before z)
            int a2 = axis == 2 ? 1 : 2;               // second transverse axis
            const std::vector<int> list = nodes[id].neighbors[wall];
            for (int neighbor : list)
            {
                deleteNeighbor(nodes, neighbor, complementWall[wall], id);
                for (int l = 0; l < 8; ++l)
                {
                    if (((l >> axis) & 1) != side) continue;
                    const Box& nb = nodes[neighbor].box;
                    bool ok1 = ((l >> a1) & 1) ? (hi(nb, a1) >= split[a1]) : (lo(nb, a1) <= split[a1]);
                    bool ok2 = ((l >> a2) & 1) ? (hi(nb, a2) >= split[a2]) : (lo(nb, a2) <= split[a2]);
                    if (ok1 && ok2) makeNeighbors(nodes, complementWall[wall], neighbor, first + l);
                }
            }
        }
    }

    // DensityTreePolicy::needsSubdivide + constructTree (DensityTreePolicy.cpp:117-231,245-309) for a single dust
    // medium without MassInBoxInterface, executed by one thread (SerialParallel) so that the random stream is
    // consumed in node order
    void OctreeSpatialGrid::setup(const Medium& medium, int numDensitySamples, Random& random)
    {
        bool hasDustFraction = maxDustFraction > 0;
        bool hasDustOpticalDepth = maxDustOpticalDepth > 0;
        bool hasDustDensityDispersion = maxDustDensityDispersion > 0;
        double dustMass = hasDustFraction ? medium.totalMass() : 0.;
        double dustKappa = 0.;
        if (hasDustOpticalDepth) dustKappa = medium.mix->sectionExt(policyWavelength) / medium.mix->mass();

        nodes.clear();
        Node root;
        root.box = extent;
        nodes.push_back(root);

        // a level is evaluated in batches: the sample positions of a batch are drawn from the random stream node by
        // node (the order in which one reference thread consumes it), the densities are evaluated on all host cores
        const size_t batchNodes = std::max<size_t>(1, (size_t(1) << 22) / std::max(1, numDensitySamples));
        std::vector<Vec3> pos;
        std::vector<double> rhov;
        size_t lbeg = 0, lend = 1;
        while (lend != lbeg)
        {
            size_t numEvalNodes = lend - lbeg;
            std::vector<char> divide(numEvalNodes, 0);
            for (size_t b0 = 0; b0 < numEvalNodes; b0 += batchNodes)
            {
                const size_t b1 = std::min(numEvalNodes, b0 + batchNodes);
                pos.clear();
                for (size_t l = b0; l != b1; ++l)
                {
                    const Node& node = nodes[lbeg + l];
                    if (node.level >= minLevel && node.level < maxLevel)
                        for (int i = 0; i != numDensitySamples; ++i) pos.push_back(random.position(node.box));
                }
                medium.massDensities(pos, rhov);
                size_t at = 0;
                for (size_t l = b0; l != b1; ++l)
                {
                    const Node& node = nodes[lbeg + l];
                    bool need = false;
                    if (node.level < minLevel)
                        need = true;
                    else if (node.level >= maxLevel)
                        need = false;
                    else
                    {
                        double rhomin = DBL_MAX, rhomax = 0., rhosum = 0;
                        for (int i = 0; i != numDensitySamples; ++i)
                        {
                            double rhoi = 0.;
                            rhoi += rhov[at++];
                            rhosum += rhoi;
                            if (rhoi < rhomin) rhomin = rhoi;
                            if (rhoi > rhomax) rhomax = rhoi;
                        }
                        double rho = rhosum / numDensitySamples;
                        double V = node.box.volume();
                        double M = rho * V;
                        if (hasDustFraction && M / dustMass > maxDustFraction) need = true;
                        if (!need && hasDustOpticalDepth && dustKappa * rho * node.box.diagonal() > maxDustOpticalDepth) need = true;
                        if (!need && hasDustDensityDispersion)
                        {
                            double q = rhomax > 0 ? (rhomax - rhomin) / rhomax : 0.;
                            if (q > maxDustDensityDispersion) need = true;
                        }
                    }
                    divide[l] = need;
                }
            }
            for (size_t l = 0; l != numEvalNodes; ++l)
                if (divide[l]) subdivide(static_cast<int>(lbeg + l));
            lbeg = lend;
            lend = nodes.size();
        }
        finish();
    }

    // ================================================================ VoronoiSpatialGrid

    void VoronoiSpatialGrid::setup(Random& random, const Medium& medium)
    {
        std::vector<Vec3> sites;
        if (policy == "DustDensity")
        {
            // VoronoiMeshSpatialGrid.cpp:22-40,73-85 (sampleMedia with ONE dust medium: the uniform deviate that selects
            // the medium is consumed all the same), positions outside the domain are discarded
            sites.resize(numSites);
            for (int m = 0; m != numSites;)
            {
                (void)random.uniform();  // NR::locateClip(Xv, uniform) with Xv = {0, 1}
                Vec3 p = medium.generatePosition(random);
                if (extent.contains(p.x, p.y, p.z)) sites[m++] = p;
            }
        }
        else if (policy == "Uniform")
        {
            // VoronoiMeshSpatialGrid.cpp:49-55: numSites positions from the simulation's random stream
            sites.resize(numSites);
            for (int m = 0; m != numSites; ++m) sites[m] = random.position(extent);
        }
        else
        {
            // VoronoiMeshSnapshot.cpp:408-417
            auto rows = readColumnFile(sitesPath, {{"position x", "length", "pc"}, {"position y", "length", "pc"}, {"position z", "length", "pc"}},
                                       "Voronoi sites");
            for (const Array& row : rows) sites.push_back(Vec3{row[0], row[1], row[2]});
        }
        mesh.build(extent, std::move(sites));
    }

    Vec3 VoronoiSpatialGrid::randomPositionInCell(int m, Random& random) const
    {
        const Box& box = mesh.cellBox(m);
        for (int i = 0; i < 10000; i++)
        {
            Vec3 r = random.position(box);
            if (mesh.isPointClosestTo(r, m)) return r;
        }
        throw std::runtime_error("Can't find random position in cell");
    }

    void VoronoiSpatialGrid::fill(pmc_grid& g) const
    {
        g = pmc_grid{};
        g.kind = PMC_GRID_VORONOI;
        g.xmin = extent.xmin, g.ymin = extent.ymin, g.zmin = extent.zmin;
        g.xmax = extent.xmax, g.ymax = extent.ymax, g.zmax = extent.zmax;
        g.eps = mesh.eps();
        g.num_cells = mesh.numCells();
        g.site = mesh.flatSites().data();
        g.vnbr_start = mesh.nbrStart().data();
        g.vnbr_list = mesh.nbrList().data();
        g.vblock_n = mesh.numBlocks();
        g.vblock_start = mesh.blockStart().data();
        g.vblock_list = mesh.blockList().data();
    }

    void OctreeSpatialGrid::setupFromTopology(const std::vector<char>& topology)
    {
        // first pass: depth-first reconstruction of parent/child relations on temporary ids
        struct Tmp
        {
            bool divided;
            int child[8];
        };
        std::vector<Tmp> tmp;
        size_t pos = 0;
        std::function<int()> read = [&]() -> int {
            if (pos >= topology.size()) throw std::runtime_error("tree topology stream ended prematurely");
            int id = static_cast<int>(tmp.size());
            tmp.push_back(Tmp{topology[pos++] != 0, {0, 0, 0, 0, 0, 0, 0, 0}});
            if (tmp[id].divided)
                for (int l = 0; l < 8; ++l)
                {
                    int c = read();
                    tmp[id].child[l] = c;
                }
            return id;
        };
        read();
        // second pass: breadth-first subdivision in the policy's order
        nodes.clear();
        Node root;
        root.box = extent;
        nodes.push_back(root);
        std::vector<int> tmpOf{0};  // temporary id of each node id
        size_t lbeg = 0, lend = 1;
        while (lend != lbeg)
        {
            for (size_t l = lbeg; l != lend; ++l)
            {
                if (tmp[tmpOf[l]].divided)
                {
                    subdivide(static_cast<int>(l));
                    for (int c = 0; c < 8; ++c) tmpOf.push_back(tmp[tmpOf[l]].child[c]);
                }
            }
            lbeg = lend;
            lend = nodes.size();
        }
        finish();
    }

    // TreeNode::sortNeighbors (TreeNode.cpp:139-207) + TreeSpatialGrid::setupSelfAfter index vectors (:38-49)
    void OctreeSpatialGrid::finish()
    {
        for (size_t id = 0; id < nodes.size(); ++id)
        {
            const Box base = nodes[id].box;
            for (int wall = 0; wall < 6; ++wall)
            {
                int axis = wall >> 1;
                int a1 = axis == 0 ? 1 : 0;
                int a2 = axis == 2 ? 1 : 2;
                auto overlap = [&](int other) {
                    const Box& nb = nodes[other].box;
                    return std::max(std::min(hi(base, a1), hi(nb, a1)) - std::max(lo(base, a1), lo(nb, a1)), 0.)
                           * std::max(std::min(hi(base, a2), hi(nb, a2)) - std::max(lo(base, a2), lo(nb, a2)), 0.);
                };
                auto& list = nodes[id].neighbors[wall];
                std::sort(list.begin(), list.end(), [&](int n1, int n2) { return overlap(n1) > overlap(n2); });
            }
        }
        int numNodes = static_cast<int>(nodes.size());
        cellIndexOfNode.assign(numNodes, -1);
        nodeOfCell.clear();
        for (int l = 0; l != numNodes; ++l)
            if (nodes[l].firstChild < 0)
            {
                cellIndexOfNode[l] = static_cast<int>(nodeOfCell.size());
                nodeOfCell.push_back(l);
            }
        // flatten
        flatBox.resize(6 * size_t(numNodes));
        flatLevel.resize(numNodes);
        flatFirstChild.resize(numNodes);
        flatCell.resize(numNodes);
        flatNbrStart.assign(6 * size_t(numNodes) + 1, 0);
        flatNbrList.clear();
        for (int l = 0; l != numNodes; ++l)
        {
            const Box& b = nodes[l].box;
            double v[6] = {b.xmin, b.ymin, b.zmin, b.xmax, b.ymax, b.zmax};
            std::memcpy(&flatBox[6 * size_t(l)], v, sizeof(v));
            flatLevel[l] = nodes[l].level;
            flatFirstChild[l] = nodes[l].firstChild;
            flatCell[l] = cellIndexOfNode[l];
            for (int wall = 0; wall < 6; ++wall)
            {
                flatNbrStart[6 * size_t(l) + wall] = static_cast<int32_t>(flatNbrList.size());
                for (int nb : nodes[l].neighbors[wall]) flatNbrList.push_back(nb);
            }
        }
        flatNbrStart[6 * size_t(numNodes)] = static_cast<int32_t>(flatNbrList.size());
    }

    void OctreeSpatialGrid::fill(pmc_grid& g) const
    {
        g.kind = PMC_GRID_OCTREE;
        g.num_nodes = static_cast<int32_t>(nodes.size());
        g.node_box = flatBox.data();
        g.node_level = flatLevel.data();
        g.node_first_child = flatFirstChild.data();
        g.node_cell = flatCell.data();
        g.nbr_start = flatNbrStart.data();
        g.nbr_list = flatNbrList.data();
    }

    // ================================================================ WavelengthGrid (DisjointWavelengthGrid.cpp)

    void WavelengthGrid::setWavelengthRange(Array lambda, bool logScale)
    {
        lambdav = std::move(lambda);
        std::sort(lambdav.begin(), lambdav.end());
        size_t n = lambdav.size();
        if (!n) throw std::runtime_error("There must be at least one wavelength in the grid");
        if (lambdav[0] <= 0.0) throw std::runtime_error("All wavelengths should be positive");
        if (std::unique(lambdav.begin(), lambdav.end()) != lambdav.end())
            throw std::runtime_error("There should be no duplicate wavelengths in the grid");
        lambdaleftv.assign(n, 0.);
        lambdarightv.assign(n, 0.);
        borderv.assign(n + 1, 0.);
        if (n == 1)
        {
            lambdaleftv[0] = borderv[0] = lambdav[0] * 0.999;
            lambdarightv[0] = borderv[1] = lambdav[0] * 1.001;
        }
        else if (logScale)
        {
            lambdaleftv[0] = borderv[0] = sqrt(lambdav[0] * lambdav[0] * lambdav[0] / lambdav[1]);
            for (size_t ell = 1; ell != n; ++ell)
                lambdarightv[ell - 1] = lambdaleftv[ell] = borderv[ell] = sqrt(lambdav[ell - 1] * lambdav[ell]);
            lambdarightv[n - 1] = borderv[n] = sqrt(lambdav[n - 1] * lambdav[n - 1] * lambdav[n - 1] / lambdav[n - 2]);
        }
        else
        {
            lambdaleftv[0] = borderv[0] = (3. * lambdav[0] - lambdav[1]) / 2.;
            for (size_t ell = 1; ell != n; ++ell)
                lambdarightv[ell - 1] = lambdaleftv[ell] = borderv[ell] = (lambdav[ell - 1] + lambdav[ell]) / 2.;
            lambdarightv[n - 1] = borderv[n] = (3. * lambdav[n - 1] - lambdav[n - 2]) / 2.;
        }
        if (lambdaleftv[0] <= 0.0) throw std::runtime_error("All wavelength bin borders should be positive");
        dlambdav.resize(n);
        for (size_t ell = 0; ell != n; ++ell) dlambdav[ell] = lambdarightv[ell] - lambdaleftv[ell];
        ellv.assign(n + 2, -1);
        for (size_t ell = 0; ell != n; ++ell) ellv[ell + 1] = static_cast<int32_t>(ell);
    }

    void WavelengthGrid::setWavelengthBins(Array lambda, double relativeHalfWidth, bool constantWidth)
    {
        lambdav = std::move(lambda);
        std::sort(lambdav.begin(), lambdav.end());
        size_t n = lambdav.size();
        if (!n) throw std::runtime_error("There must be at least one wavelength in the grid");
        if (lambdav[0] <= 0) throw std::runtime_error("All wavelengths should be positive");
        lambdaleftv.assign(n, 0.);
        lambdarightv.assign(n, 0.);
        borderv.assign(2 * n, 0.);
        if (!constantWidth)
        {
            for (size_t ell = 0; ell != n; ++ell)
            {
                borderv[2 * ell] = lambdaleftv[ell] = lambdav[ell] * (1. - relativeHalfWidth);
                borderv[2 * ell + 1] = lambdarightv[ell] = lambdav[ell] * (1. + relativeHalfWidth);
            }
        }
        else
        {
            double delta = lambdav[0] * relativeHalfWidth;
            for (size_t ell = 0; ell != n; ++ell)
            {
                borderv[2 * ell] = lambdaleftv[ell] = lambdav[ell] - delta;
                borderv[2 * ell + 1] = lambdarightv[ell] = lambdav[ell] + delta;
            }
        }
        if (!std::is_sorted(borderv.begin(), borderv.end()))
            throw std::runtime_error("Non-adjacent wavelength bins should not overlap");
        dlambdav.resize(n);
        for (size_t ell = 0; ell != n; ++ell) dlambdav[ell] = lambdarightv[ell] - lambdaleftv[ell];
        ellv.assign(2 * n + 1, -1);
        for (size_t ell = 0; ell != n; ++ell) ellv[2 * ell + 1] = static_cast<int32_t>(ell);
    }

    int WavelengthGrid::bin(double lambda) const
    {
        size_t index = std::upper_bound(borderv.begin(), borderv.end(), lambda) - borderv.begin();
        return ellv[index];
    }
}
