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

    // ================================================================ shared pieces of the geometries

    namespace
    {
        double cylindricalRadius(Vec3 r) { return sqrt(r.x * r.x + r.y * r.y); }           // Position::cylRadius
        double sphericalRadius(Vec3 r) { return sqrt(r.x * r.x + r.y * r.y + r.z * r.z); }  // Vec::norm
        Vec3 alongDirection(double radius, Vec3 k) { return Vec3{radius * k.x, radius * k.y, radius * k.z}; }  // Position(r, bfk)
        Vec3 fromCylinder(double R, double phi, double z) { return Vec3{R * cos(phi), R * sin(phi), z}; }

        // composite trapezoid rule over the nodes 0..count with unit spacing: the end nodes weigh one half, the terms are
        // accumulated in node order
        template<class Term> double trapezoidSum(int count, Term term)
        {
            double total = 0.0;
            for (int node = 0; node <= count; ++node)
            {
                const double value = term(node);
                total += (node == 0 || node == count) ? 0.5 * value : value;
            }
            return total;
        }

        // height drawn from a two-sided exponential of scale h (the vertical profile of the disk and of the ring)
        double twoSidedExponential(double h, double X)
        {
            if (X <= 0.5) return h * log(2.0 * X);
            return -h * log(2.0 * (1.0 - X));
        }

        // lower branch W_-1 of the Lambert function on [-1/e, 0] (SpecialFunctions.cpp:578-625): a series in
        // sqrt(z + 1/e) evaluated by Horner's rule, refined by Halley iterations away from the branch point
        double lambertLowerBranch(double z)
        {
            static const double inverseE = 0.3678794411714423215955237701614608;
            static const double series[12] = {-1.0, 2.331643981597124203363536062168, -1.812187885639363490240191647568,
                                              1.936631114492359755363277457668, -2.353551201881614516821543561516,
                                              3.066858901050631912893148922704, -4.175335600258177138854984177460,
                                              5.858023729874774148815053846119, -8.401032217523977370984161688514,
                                              12.250753501314460424, -18.100697012472442755, 27.029044799010561650};
            if (!(z >= -inverseE && z <= 0.0)) throw std::runtime_error("LambertW1: bad argument");
            if (z == 0.0) return -DBL_MAX;
            const double fromBranchPoint = z + inverseE;
            const double root = -sqrt(fromBranchPoint);
            double estimate = series[11];
            for (int term = 10; term >= 0; --term) estimate = series[term] + root * estimate;
            if (fromBranchPoint < 3.0e-3) return estimate;
            double w = estimate;
            if (!(z < -1e-6))
            {
                const double outer = log(-z);
                const double inner = log(-outer);
                w = outer - inner + inner / outer;
            }
            for (int round = 0; round < 10; ++round)
            {
                const double ew = exp(w);
                const double next = w + 1.0;
                double correction = w * ew - z;
                correction /= ew * next - 0.5 * (next + 1.0) * correction / next;
                w -= correction;
                if (fabs(correction) < 1.0e-12 * (1.0 + fabs(w))) return w;
            }
            throw std::runtime_error("LambertW1: no convergence");
        }
    }

    // ================================================================ SersicProfile (SersicFunction.cpp:13-101)

    double SersicProfile::shapeB(double index)
    {
        const double n2 = index * index;
        double b = 2.0 * index - 1.0 / 3.0;
        b = b + 4.0 / 405.0 / index;
        b = b + 46.0 / 25515.0 / n2;
        b = b + 131.0 / 1148175.0 / (n2 * index);
        return b;
    }

    // Abel deprojection of the Sersic surface brightness at radius s (in effective radii): the substitution t^2 = 1 + u^2
    // on the line of sight, 10^4 trapezoid intervals up to t = 100
    double SersicProfile::deproject(double index, double b, double central, double s) const
    {
        const int intervals = 10000;
        const double lastT = 100.0;
        const double lastU = sqrt((lastT + 1.0) * (lastT - 1.0));
        const double spacing = lastU / intervals;
        const double twice = 2.0 * index;
        const double depth = b * pow(s, 1.0 / index);
        const double lineOfSight = trapezoidSum(intervals, [&](int node) {
            const double u = node * spacing;
            const double usq = u * u;
            double jacobian;
            if (u > 1e-3)
                jacobian = (pow(1.0 + usq, twice) - 1.0) / usq;
            else
            {
                const double odd = twice - 1.0;
                jacobian = twice + index * odd * usq + 2.0 / 3.0 * index * odd * (index - 1.0) * usq * usq;
            }
            return 2.0 * exp(-depth * (1.0 + usq)) / sqrt(jacobian);
        });
        return central * pow(b, index) * pow(depth, 1.0 - index) / M_PI * spacing * lineOfSight;
    }

    SersicProfile::SersicProfile(double index)
    {
        if (index < 0.5 || index > 10.0) throw std::runtime_error("The Sersic parameter should be between 0.5 and 10");
        const double b = shapeB(index);
        const double central = pow(b, 2.0 * index) / (M_PI * special::gamma(2.0 * index + 1));
        // 101 radii, ten per decade from 10^-6 to 10^4 effective radii
        const int count = 101;
        const double firstDecade = -6.0, lastDecade = 4.0;
        const double decadeStep = (lastDecade - firstDecade) / (count - 1.0);
        radius_.assign(count, 0.);
        profile_.assign(count, 0.);
        mass_.assign(count, 0.);
        for (int i = 0; i < count; ++i)
        {
            radius_[i] = pow(10.0, firstDecade + i * decadeStep);
            profile_[i] = deproject(index, b, central, radius_[i]);
        }
        // mass within each radius: 32 trapezoid intervals per table interval, on the interpolated profile
        const int pieces = 32;
        for (int i = 1; i < count; ++i)
        {
            const double inner = radius_[i - 1];
            const double piece = (radius_[i] - inner) / 32.0;
            const double shell = trapezoidSum(pieces, [&](int node) {
                const double s = inner + node * piece;
                return value(s) * s * s * piece;
            });
            mass_[i] = mass_[i - 1] + 4.0 * M_PI * shell;
        }
        const double total = mass_[count - 1];
        for (double& m : mass_) m /= total;
    }
    double SersicProfile::value(double s) const { return tab::clampedAt<tab::logLog>(s, radius_, profile_); }
    double SersicProfile::radiusOfMass(double M) const { return tab::clampedAt<tab::logLog>(M, mass_, radius_); }

    // ================================================================ geometries

    // UniformBoxGeometry.cpp:12-61
    UniformBoxGeometry::UniformBoxGeometry(const Box& box) : bounds_(box)
    {
        const bool empty = box.xmax - box.xmin <= 0 || box.ymax - box.ymin <= 0 || box.zmax - box.zmin <= 0;
        if (empty) throw std::runtime_error("The extent of the box should be positive in every direction");
        level_ = 1. / bounds_.volume();
    }
    double UniformBoxGeometry::density(Vec3 r) const { return bounds_.contains(r.x, r.y, r.z) ? level_ : 0.; }
    double UniformBoxGeometry::columnX() const { return 1. / ((bounds_.ymax - bounds_.ymin) * (bounds_.zmax - bounds_.zmin)); }
    double UniformBoxGeometry::columnY() const { return 1. / ((bounds_.xmax - bounds_.xmin) * (bounds_.zmax - bounds_.zmin)); }
    double UniformBoxGeometry::columnZ() const { return 1. / ((bounds_.xmax - bounds_.xmin) * (bounds_.ymax - bounds_.ymin)); }

    // ExpDiskGeometry.cpp:13-90
    ExpDiskGeometry::ExpDiskGeometry(double radialScale, double verticalScale, double innerRadius, double outerRadius, double maxHeight)
        : scaleR_(radialScale), scaleZ_(verticalScale), innerR_(innerRadius), outerR_(outerRadius), maxZ_(maxHeight)
    {
        if (outerR_ > 0 && innerR_ >= outerR_)
            throw std::runtime_error("The radius of the central cavity should be smaller than the truncation radius");
        // the density integrates to one: azimuth x height x radius
        const double azimuth = 2.0 * M_PI;
        const double height = maxZ_ > 0 ? -2.0 * scaleZ_ * expm1(-maxZ_ / scaleZ_) : 2.0 * scaleZ_;
        auto radialPrimitive = [this](double R) { return exp(-R / scaleR_) * (1.0 + R / scaleR_); };
        const double fromInner = innerR_ > 0 ? radialPrimitive(innerR_) : 1.0;
        const double fromOuter = outerR_ > 0 ? radialPrimitive(outerR_) : 0.0;
        const double radius = scaleR_ * scaleR_ * (fromInner - fromOuter);
        central_ = 1.0 / (radius * azimuth * height);
    }
    double ExpDiskGeometry::density(Vec3 r) const
    {
        const double R = cylindricalRadius(r);
        const double height = fabs(r.z);
        const bool outside = (outerR_ > 0.0 && R > outerR_) || (maxZ_ > 0.0 && height > maxZ_) || R < innerR_;
        if (outside) return 0.0;
        return central_ * exp(-R / scaleR_) * exp(-height / scaleZ_);
    }
    double ExpDiskGeometry::radialColumn() const
    {
        const double atInner = exp(-innerR_ / scaleR_);
        if (outerR_ > 0.0) return central_ * scaleR_ * (atInner - exp(-outerR_ / scaleR_));
        return central_ * scaleR_ * atInner;
    }
    double ExpDiskGeometry::columnZ() const
    {
        if (innerR_ > 0.0) return 0.0;  // the axis runs through the cavity
        if (maxZ_ > 0.0) return -2.0 * central_ * scaleZ_ * expm1(-maxZ_ / scaleZ_);
        return 2.0 * central_ * scaleZ_;
    }
    Vec3 ExpDiskGeometry::samplePosition(Random& random) const
    {
        // radius: inversion of 1 - (1 + R/h) exp(-R/h) through the Lambert function, rejection on cavity and truncation
        double R;
        while (true)
        {
            const double X = random.uniform();
            R = scaleR_ * (-1.0 - lambertLowerBranch((X - 1.0) / M_E));
            const bool rejected = (outerR_ > 0.0 && R >= outerR_) || R <= innerR_;
            if (!rejected) break;
        }
        const double phi = 2.0 * M_PI * random.uniform();
        double z;
        do
            z = twoSidedExponential(scaleZ_, random.uniform());
        while (maxZ_ > 0.0 && fabs(z) >= maxZ_);
        return fromCylinder(R, phi, z);
    }

    // SersicGeometry.cpp:21-52
    SersicGeometry::SersicGeometry(double effectiveRadius, double index) : reff_(effectiveRadius), index_(index)
    {
        central_ = 1.0 / (reff_ * reff_ * reff_);
        b_ = SersicProfile::shapeB(index_);
        profile_ = std::make_unique<SersicProfile>(index_);
    }
    double SersicGeometry::density(Vec3 r) const { return central_ * profile_->value(sphericalRadius(r) / reff_); }
    double SersicGeometry::radialColumn() const
    {
        const double twice = 2.0 * index_;
        return 1.0 / (reff_ * reff_) * pow(b_, twice) / (2.0 * M_PI * special::gamma(twice + 1.0));
    }
    Vec3 SersicGeometry::samplePosition(Random& random) const
    {
        const double radius = reff_ * profile_->radiusOfMass(random.uniform());  // SersicGeometry::randomRadius
        return alongDirection(radius, random.direction());
    }

    // PlummerGeometry.cpp:12-38
    PlummerGeometry::PlummerGeometry(double scale) : scale_(scale) { central_ = 0.75 / pow(scale_, 3) / M_PI; }
    double PlummerGeometry::density(Vec3 r) const
    {
        const double s = sphericalRadius(r) / scale_;
        return central_ * pow(1.0 + s * s, -2.5);
    }
    double PlummerGeometry::radialColumn() const { return 0.5 / (M_PI * scale_ * scale_); }
    Vec3 PlummerGeometry::samplePosition(Random& random) const
    {
        const double t = pow(random.uniform(), 1.0 / 3.0);
        const double radius = scale_ * t / sqrt((1.0 - t) * (1.0 + t));
        return alongDirection(radius, random.direction());
    }

    // GaussianGeometry.cpp:11-60
    GaussianGeometry::GaussianGeometry(double dispersion) : sigma_(dispersion)
    {
        central_ = 1.0 / pow(sqrt(2.0 * M_PI) * sigma_, 3);
        const int N = 401;
        const double logtmin = -4.0, logtmax = 4.0;
        const double dlogt = (logtmax - logtmin) / (N - 1.0);
        rv_.assign(N, 0.);
        Xv_.assign(N, 0.);
        for (int i = 1; i < N - 1; i++)
        {
            const double logt = logtmin + i * dlogt;
            const double t = pow(10.0, logt);
            rv_[i] = M_SQRT2 * sigma_ * t;
            Xv_[i] = erf(t) - M_2_SQRTPI * t * exp(-t * t);
        }
        Xv_[N - 1] = 1.0;
    }
    double GaussianGeometry::density(Vec3 r) const
    {
        const double rad = sphericalRadius(r);
        const double r2 = rad * rad;
        const double sigma2 = sigma_ * sigma_;
        return central_ * exp(-0.5 * r2 / sigma2);
    }
    Vec3 GaussianGeometry::samplePosition(Random& random) const
    {
        // Random::cdfLinLin(_rv, _Xv): NR::locateClip + NR::interpolateLinLin
        const double X = random.uniform();
        const int n = static_cast<int>(Xv_.size());
        int i = static_cast<int>(std::upper_bound(Xv_.begin(), Xv_.end(), X) - Xv_.begin()) - 1;
        i = std::max(0, std::min(n - 2, i));
        const double radius = rv_[i] + ((X - Xv_[i]) / (Xv_[i + 1] - Xv_[i])) * (rv_[i + 1] - rv_[i]);  // NR::interpolateLinLin
        return alongDirection(radius, random.direction());
    }

    // ================================================================ SpheroidalGeometryDecorator

    double SpheroidalGeometry::density(Vec3 r) const
    {
        // the spherical density at the spheroidal radius (AxGeometry::density passes cylindrical radius and height; a
        // position (m, 0, 0) has norm m exactly)
        const double R = cylindricalRadius(r);
        const double m = sqrt(R * R + r.z * r.z / (flat_ * flat_));
        return 1.0 / flat_ * sphere_->density(Vec3{m, 0., 0.});
    }
    Vec3 SpheroidalGeometry::samplePosition(Random& random) const
    {
        const Vec3 p = sphere_->samplePosition(random);
        return Vec3{p.x, p.y, flat_ * p.z};
    }

    // ================================================================ ShellGeometry, TorusGeometry, RingGeometry

    void PowerLawRadius::prepare(double rmin, double rmax, double p)
    {
        inner = rmin, outer = rmax, exponent = p;
        // generalised logarithms of exponent p - 2 (SpecialFunctions::gln, gln2): the cumulative radial distribution
        logInner = special::gln(p - 2.0, rmin);
        logSpan = pow(rmin, 1.0 - (p - 2.0)) * special::gln(p - 2.0, rmax / rmin);
        powInner = pow(rmin, 3.0 - p);
        powOuter = pow(rmax, 3.0 - p);
    }
    double PowerLawRadius::sample(double X) const
    {
        if (fabs(exponent - 3.0) < 1e-2) return special::gexp(exponent - 2.0, logInner + X * logSpan);
        return pow((1.0 - X) * powInner + X * powOuter, 1.0 / (3.0 - exponent));
    }
    double PowerLawRadius::column(double amplitude) const
    {
        return amplitude * (pow(inner, 1.0 - exponent) * special::gln(exponent, outer / inner));
    }

    ShellGeometry::ShellGeometry(double rmin, double rmax, double p)
    {
        if (rmax <= rmin) throw std::runtime_error("the outer radius of the shell should be larger than the inner radius");
        radial_.prepare(rmin, rmax, p);
        amplitude_ = 0.25 / M_PI / radial_.logSpan;
    }
    double ShellGeometry::density(Vec3 r) const
    {
        const double radius = sphericalRadius(r);
        if (radius < radial_.inner || radius > radial_.outer) return 0.0;
        return amplitude_ * pow(radius, -radial_.exponent);
    }
    Vec3 ShellGeometry::samplePosition(Random& random) const
    {
        const double radius = radial_.sample(random.uniform());
        return alongDirection(radius, random.direction());
    }

    TorusGeometry::TorusGeometry(double p, double q, double halfOpening, double rmin, double rmax, bool anisotropicInner,
                                 double cutoffRadius)
        : polar_(q), cutoff_(cutoffRadius), anisotropic_(anisotropicInner)
    {
        sinOpening_ = sin(halfOpening);
        radial_.prepare(rmin, rmax, p);
        if (polar_ > 1e-3)
            amplitude_ = polar_ * 0.25 / M_PI / radial_.logSpan / (1.0 - exp(-polar_ * sinOpening_));
        else
            amplitude_ = 0.25 / M_PI / radial_.logSpan / sinOpening_;
    }
    double TorusGeometry::density(Vec3 position) const
    {
        // (AxGeometry::density passes the cylindrical radius and the height, AxGeometry.cpp:11-17)
        const double R = cylindricalRadius(position), z = position.z;
        const double radius = sqrt(R * R + z * z);
        const double slope = fabs(z / radius);  // |cos theta|
        if (radius >= radial_.outer) return 0.0;
        if (anisotropic_)
        {
            // inner wall that recedes towards the equator, and the sublimation cutoff
            const double wall = radial_.inner * sqrt(6. / 7. * slope * (2. * slope + 1));
            if (radius <= wall || radius < cutoff_) return 0.0;
        }
        else if (radius <= radial_.inner)
            return 0.0;
        if (slope >= sinOpening_) return 0.0;
        return amplitude_ * pow(radius, -radial_.exponent) * exp(-polar_ * slope);
    }
    Vec3 TorusGeometry::samplePosition(Random& random) const
    {
        // radius, polar angle and azimuth from their marginal distributions; rejection against the anisotropic wall
        while (true)
        {
            const double radius = radial_.sample(random.uniform());
            const double X = random.uniform();
            double cosine;
            if (polar_ < 1e-3)
                cosine = (1.0 - 2.0 * X) * sinOpening_;
            else
            {
                const double span = 1.0 - exp(-polar_ * sinOpening_);
                if (X < 0.5)
                    cosine = -log(1.0 - span * (1.0 - 2.0 * X)) / polar_;
                else
                    cosine = log(1.0 - span * (2.0 * X - 1.0)) / polar_;
            }
            const double polarAngle = acos(cosine);
            const double phi = 2.0 * M_PI * random.uniform();
            const double sine = sin(polarAngle);  // Position(r, theta, phi, SPHERICAL)
            const Vec3 candidate{radius * sine * cos(phi), radius * sine * sin(phi), radius * cos(polarAngle)};
            if (density(candidate)) return candidate;
        }
    }

    RingGeometry::RingGeometry(double ringRadius, double width, double verticalScale)
        : centre_(ringRadius), width_(width), scaleZ_(verticalScale)
    {
        const double rootPi = sqrt(M_PI);
        const double t = centre_ / width_ / M_SQRT2;
        const double gaussAtAxis = exp(-t * t);
        const double errAtAxis = erf(t);
        const double height = 2.0 * scaleZ_;
        const double radial = width_ * width_ * (gaussAtAxis + rootPi * t * (1.0 + errAtAxis));
        amplitude_ = 1.0 / (2.0 * M_PI * height * radial);
        // cumulative radial distribution on 330 radii within eight widths of the ring
        const int count = 330;
        tab::linearGrid(tableR_, std::max(0., centre_ - 8 * width_), centre_ + 8 * width_, count - 1);
        tableCdf_.assign(count, 0.);
        const double scale = 4.0 * M_PI * amplitude_ * scaleZ_ * width_ * width_;
        for (int i = 0; i < count; ++i)
        {
            const double u = (centre_ - tableR_[i]) / width_ / M_SQRT2;
            tableCdf_[i] = scale * (gaussAtAxis - exp(-u * u) + rootPi * t * (errAtAxis - erf(u)));
        }
        tableCdf_.front() = 0.0;
        tableCdf_.back() = 1.0;
    }
    double RingGeometry::density(Vec3 r) const
    {
        const double u = (cylindricalRadius(r) - centre_) / (M_SQRT2 * width_);
        return amplitude_ * exp(-u * u) * exp(-fabs(r.z) / scaleZ_);
    }
    double RingGeometry::radialColumn() const
    {
        const double t = centre_ / (M_SQRT2 * width_);
        return sqrt(M_PI / 2.0) * amplitude_ * width_ * (1.0 + erf(t));
    }
    double RingGeometry::columnZ() const
    {
        const double t = centre_ / (M_SQRT2 * width_);
        return 2.0 * amplitude_ * scaleZ_ * exp(-t * t);
    }
    Vec3 RingGeometry::samplePosition(Random& random) const
    {
        // Random::cdfLinLin (Random.cpp:190-195) on the tabulated radial distribution, then azimuth, then height
        const double X = random.uniform();
        const int i = tab::bracketClipped(tableCdf_, X);
        const double R = tab::linLin(X, tableCdf_[i], tableCdf_[i + 1], tableR_[i], tableR_[i + 1]);
        const double phi = 2.0 * M_PI * random.uniform();
        const double z = twoSidedExponential(scaleZ_, random.uniform());
        return fromCylinder(R, phi, z);
    }

    // ================================================================ DustMix (DustMix.cpp:47-162)

    void DustMix::setup(double rangeMin, double rangeMax, const std::vector<double>& simulationWavelengths)
    {
        if (inLambda.size() != inKappaExt.size() || inLambda.size() != inAlbedo.size()
            || inLambda.size() != inAsymmpar.size())
            throw std::runtime_error("Number of listed properties does not match number of listed wavelengths");
        if (inLambda.empty()) throw std::runtime_error("Dust properties must be tabulated for at least one wavelength");

        // sample wavelengths: every integer multiple of 1/1000 dex inside the range, plus all configured wavelengths
        // (DustMix.cpp:57-72)
        const double perDecade = 1000;
        const int firstTick = std::floor(perDecade * log10(rangeMin));
        const int lastTick = std::ceil(perDecade * log10(rangeMax));
        std::vector<double> samples;
        samples.reserve(lastTick - firstTick + 1 + simulationWavelengths.size());
        for (int tick = firstTick; tick <= lastTick; ++tick) samples.push_back(pow(10., tick / perDecade));
        samples.insert(samples.end(), simulationWavelengths.begin(), simulationWavelengths.end());
        std::sort(samples.begin(), samples.end());
        samples.erase(std::unique(samples.begin(), samples.end()), samples.end());

        // beyond 10 cm the dust is transparent: the table ends with 10 cm and one point just above it (DustMix.cpp:74-82)
        const double tenCm = 0.1;
        const bool beyondRadio = samples.back() > tenCm;
        if (beyondRadio)
        {
            samples.resize(tab::bracket(samples, tenCm) + 1);
            if (samples.empty() || samples.back() != tenCm) samples.push_back(tenCm);
            samples.push_back(tenCm * 1.001);
        }
        lambdaSample = samples;
        const int numLambda = static_cast<int>(lambdaSample.size());

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
        sigmaAbs = tab::resampleClamped<tab::logLog>(lambdaSample, inl, insigmaabs);
        sigmaSca = tab::resampleClamped<tab::logLog>(lambdaSample, inl, insigmasca);
        asymmpar = tab::resampleClamped<tab::logLin>(lambdaSample, inl, ing);

        // clamp g (DustMix.cpp:139-146), derive extinction (:160-162)
        const double gmax = 0.999999;
        for (int ell = 0; ell < numLambda; ++ell)
            if (std::abs(asymmpar[ell]) > gmax) asymmpar[ell] = std::copysign(gmax, asymmpar[ell]);
        sigmaExt.assign(numLambda, 0.);
        for (int ell = 0; ell < numLambda; ++ell) sigmaExt[ell] = sigmaAbs[ell] + sigmaSca[ell];
        if (beyondRadio)
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
            double geomColumnDensity = normAxis == 'X' ? geometry->columnX() : normAxis == 'Y' ? geometry->columnY() : geometry->columnZ();
            if (geomColumnDensity <= 0.)
                throw std::runtime_error("Can't normalize material for geometry with zero column density along selected axis");
            double section = mix->sectionExt(normWavelength);
            if (section <= 0.) throw std::runtime_error("Can't normalize optical depth for material with zero extinction");
            const double numberColumn = normOpticalDepth / section;
            const double massColumn = numberColumn * mix->mass();
            number = numberColumn / geomColumnDensity;
            mass = massColumn / geomColumnDensity;
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

    // ================================================================ meshes on the unit interval

    // Border points of the reference's mesh classes (LinMesh.cpp:11-16, PowMesh.cpp:11-19, SymPowMesh.cpp:11-19,
    // LogMesh.cpp:11-19, SymLogMesh.cpp:11-42 with the grid builders of NR.hpp:203-320), n bins on [0, 1]
    namespace mesh
    {
        // bin widths in geometric progression: the last bin is `ratio` times the first
        Array geometric(int n, double ratio)
        {
            Array points(n + 1);
            const double first = 0.0, span = 1.0 - 0.0;
            const double step = pow(ratio, 1. / (n - 1));
            const double whole = pow(step, n);
            for (int k = 0; k <= n; ++k) points[k] = first + (1. - pow(step, k)) / (1. - whole) * span;
            return points;
        }
        // the same progression mirrored about the centre: outermost bins `ratio` times the innermost
        Array symmetricGeometric(int n, double ratio)
        {
            Array points(n + 1);
            const double centre = 0.5 * (0.0 + 1.0);
            const double halfSpan = 0.5;
            const int half = n % 2 == 0 ? n / 2 : (n + 1) / 2;
            const double step = pow(ratio, 1.0 / (half - 1.0));
            const double whole = pow(step, half);
            if (n % 2 == 0)
            {
                points[half] = centre;
                for (int k = 1; k <= half; ++k)
                {
                    const double offset = (1.0 - pow(step, k)) / (1.0 - whole) * halfSpan;
                    points[half + k] = centre + offset;
                    points[half - k] = centre - offset;
                }
            }
            else
            {
                // the central bin straddles the centre
                const double pivot = 0.5 + 0.5 * step;
                for (int k = 1; k <= half; ++k)
                {
                    const double offset = (pivot - pow(step, k)) / (pivot - whole) * halfSpan;
                    points[half - 1 + k] = centre + offset;
                    points[half - k] = centre - offset;
                }
            }
            return points;
        }
        // first bin [0, fraction], the others logarithmic up to one
        Array logarithmicFromZero(int n, double fraction)
        {
            Array points(n + 1, 0.);
            const double origin = log(fraction);
            const double spacing = log(1.0 / fraction) / (n - 1);
            for (int k = 0; k < n; ++k) points[k + 1] = exp(origin + k * spacing);
            return points;
        }
        // logarithmic bins away from the centre on both sides; central bin(s) of width `fraction` of a half
        Array symmetricLogarithmic(int n, double fraction)
        {
            const int perSide = (n - 1) / 2;
            Array side;
            tab::logGrid(side, fraction, 1., perSide);
            Array points;
            points.reserve(n + 1);
            points.push_back(0.);
            for (int k = perSide - 1; k >= 0; --k) points.push_back(0.5 - 0.5 * side[k]);
            if (n % 2 == 0) points.push_back(0.5);
            for (int k = 0; k < perSide; ++k) points.push_back(0.5 + 0.5 * side[k]);
            points.push_back(1.);
            return points;
        }
    }

    // ================================================================ CartesianSpatialGrid

    void CartesianSpatialGrid::setup()
    {
        // Mesh::mesh of the configured class on [0,1] (LinMesh.cpp:11-16, PowMesh.cpp:11-19, SymPowMesh.cpp:11-19,
        // LogMesh.cpp:11-19, SymLogMesh.cpp:11-42 with the grid builders of NR.hpp:203-320), scaled as
        // CartesianSpatialGrid.cpp:22-24
        auto meshOf = [](const MeshSpec& spec, int n) {
            const bool uniformRatio = fabs(spec.ratio - 1.) < 1e-3;
            Array points;
            if (spec.type == "ListMesh")
                points.assign(spec.points.begin(), spec.points.end());
            else if (spec.type == "PowMesh" && n > 1 && !uniformRatio)
                points = mesh::geometric(n, spec.ratio);
            else if (spec.type == "SymPowMesh" && n > 2 && !uniformRatio)
                points = mesh::symmetricGeometric(n, spec.ratio);
            else if (spec.type == "LogMesh" && n > 1)
                points = mesh::logarithmicFromZero(n, spec.centralBinFraction);
            else if (spec.type == "SymLogMesh" && n > 2)
                points = mesh::symmetricLogarithmic(n, spec.centralBinFraction);
            else
            {
                // uniform bins; a one-bin power-law or logarithmic mesh is the unit interval
                const bool single = (spec.type == "PowMesh" || spec.type == "LogMesh") && n <= 1;
                tab::linearGrid(points, 0.0, 1.0, single ? 1 : n);
            }
            return points;
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
        if (hasDustOpticalDepth) dustKappa = medium.dustKappa(policyWavelength);

        nodes.clear();
        Node root;
        root.box = extent;
        nodes.push_back(root);

        // a level is evaluated in batches: the sample positions of a batch are drawn from the random stream node by
        // node (the order in which one reference thread consumes it), the densities are evaluated on all host cores
        const size_t batchNodes = std::max<size_t>(1, (size_t(1) << 22) / std::max(1, numDensitySamples));
        std::vector<Vec3> pos;
        std::vector<double> sampleDensities;
        size_t levelFirst = 0, levelEnd = 1;
        while (levelEnd != levelFirst)
        {
            size_t levelCount = levelEnd - levelFirst;
            std::vector<char> divide(levelCount, 0);
            for (size_t b0 = 0; b0 < levelCount; b0 += batchNodes)
            {
                const size_t b1 = std::min(levelCount, b0 + batchNodes);
                pos.clear();
                for (size_t l = b0; l != b1; ++l)
                {
                    const Node& node = nodes[levelFirst + l];
                    if (node.level >= minLevel && node.level < maxLevel)
                        for (int i = 0; i != numDensitySamples; ++i) pos.push_back(random.position(node.box));
                }
                medium.massDensities(pos, sampleDensities);
                size_t at = 0;
                for (size_t l = b0; l != b1; ++l)
                {
                    const Node& node = nodes[levelFirst + l];
                    bool need = false;
                    if (node.level < minLevel)
                        need = true;
                    else if (node.level >= maxLevel)
                        need = false;
                    else
                    {
                        double lowest = DBL_MAX, highest = 0., sampleSum = 0;
                        for (int i = 0; i != numDensitySamples; ++i)
                        {
                            double sampled = 0.;
                            sampled += sampleDensities[at++];
                            sampleSum += sampled;
                            if (sampled < lowest) lowest = sampled;
                            if (sampled > highest) highest = sampled;
                        }
                        double rho = sampleSum / numDensitySamples;
                        double V = node.box.volume();
                        double M = rho * V;
                        if (hasDustFraction && M / dustMass > maxDustFraction) need = true;
                        if (!need && hasDustOpticalDepth && dustKappa * rho * node.box.diagonal() > maxDustOpticalDepth) need = true;
                        if (!need && hasDustDensityDispersion)
                        {
                            const double contrast = highest > 0 ? (highest - lowest) / highest : 0.;
                            if (contrast > maxDustDensityDispersion) need = true;
                        }
                    }
                    divide[l] = need;
                }
            }
            for (size_t l = 0; l < levelCount; ++l)
                if (divide[l]) subdivide(static_cast<int>(levelFirst + l));
            levelFirst = levelEnd;
            levelEnd = nodes.size();
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
            for (int m = 0; m < numSites;)
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
        else if (policy == "CentralPeak")
        {
            // VoronoiMeshSpatialGrid.cpp:67-82: the first site stays at the origin, the others follow a 1/r distribution down to
            // 1/1000 of the domain's largest radius, in isotropic directions; positions outside the domain are drawn again
            const int a = 1000;
            const double rscale = std::sqrt(extent.xmax * extent.xmax + extent.ymax * extent.ymax + extent.zmax * extent.zmax);
            sites.assign(numSites, Vec3{0., 0., 0.});
            for (int m = 1; m < numSites;)
            {
                const double r = rscale * std::pow(1. / a, random.uniform());
                const Vec3 k = random.direction();
                const Vec3 p{r * k.x, r * k.y, r * k.z};
                if (extent.contains(p.x, p.y, p.z)) sites[m++] = p;
            }
        }
        else if (policy == "ImportedSites")
        {
            // VoronoiMeshSpatialGrid.cpp:127-132: the positions of the entities of the first imported medium component
            // (MediumSystem::interface<SiteListInterface>; ImportedMedium.cpp:268-278), in file order
            const ParticleMedium* imported = dynamic_cast<const ParticleMedium*>(&medium);
            if (!imported)
                if (const CompositeMedium* all = dynamic_cast<const CompositeMedium*>(&medium))
                    for (const Medium* part : all->parts)
                        if (!imported) imported = dynamic_cast<const ParticleMedium*>(part);
            if (!imported) throw std::runtime_error("VoronoiMeshSpatialGrid policy ImportedSites: no medium component offers a list of sites (an imported medium)");
            sites = imported->snapshot.sitePositions();
        }
        else
        {
            // VoronoiMeshSnapshot.cpp:408-417
            auto rows = readColumnFile(sitesPath, {{"position x", "length", "pc"}, {"position y", "length", "pc"}, {"position z", "length", "pc"}},
                                       "Voronoi sites");
            for (const Array& row : rows) sites.push_back(Vec3{row[0], row[1], row[2]});
        }
        mesh.build(extent, std::move(sites), relaxSites);
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
        size_t levelFirst = 0, levelEnd = 1;
        while (levelEnd != levelFirst)
        {
            for (size_t l = levelFirst; l < levelEnd; ++l)
            {
                if (tmp[tmpOf[l]].divided)
                {
                    subdivide(static_cast<int>(l));
                    for (int c = 0; c < 8; ++c) tmpOf.push_back(tmp[tmpOf[l]].child[c]);
                }
            }
            levelFirst = levelEnd;
            levelEnd = nodes.size();
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
        for (int l = 0; l < numNodes; ++l)
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
        for (int l = 0; l < numNodes; ++l)
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
            for (size_t ell = 1; ell < n; ++ell)
                lambdarightv[ell - 1] = lambdaleftv[ell] = borderv[ell] = sqrt(lambdav[ell - 1] * lambdav[ell]);
            lambdarightv[n - 1] = borderv[n] = sqrt(lambdav[n - 1] * lambdav[n - 1] * lambdav[n - 1] / lambdav[n - 2]);
        }
        else
        {
            lambdaleftv[0] = borderv[0] = (3. * lambdav[0] - lambdav[1]) / 2.;
            for (size_t ell = 1; ell < n; ++ell)
                lambdarightv[ell - 1] = lambdaleftv[ell] = borderv[ell] = (lambdav[ell - 1] + lambdav[ell]) / 2.;
            lambdarightv[n - 1] = borderv[n] = (3. * lambdav[n - 1] - lambdav[n - 2]) / 2.;
        }
        if (lambdaleftv[0] <= 0.0) throw std::runtime_error("All wavelength bin borders should be positive");
        dlambdav.resize(n);
        for (size_t ell = 0; ell < n; ++ell) dlambdav[ell] = lambdarightv[ell] - lambdaleftv[ell];
        ellv.assign(n + 2, -1);
        for (size_t ell = 0; ell < n; ++ell) ellv[ell + 1] = static_cast<int32_t>(ell);
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
            for (size_t ell = 0; ell < n; ++ell)
            {
                borderv[2 * ell] = lambdaleftv[ell] = lambdav[ell] * (1. - relativeHalfWidth);
                borderv[2 * ell + 1] = lambdarightv[ell] = lambdav[ell] * (1. + relativeHalfWidth);
            }
        }
        else
        {
            double delta = lambdav[0] * relativeHalfWidth;
            for (size_t ell = 0; ell < n; ++ell)
            {
                borderv[2 * ell] = lambdaleftv[ell] = lambdav[ell] - delta;
                borderv[2 * ell + 1] = lambdarightv[ell] = lambdav[ell] + delta;
            }
        }
        if (!std::is_sorted(borderv.begin(), borderv.end()))
            throw std::runtime_error("Non-adjacent wavelength bins should not overlap");
        dlambdav.resize(n);
        for (size_t ell = 0; ell < n; ++ell) dlambdav[ell] = lambdarightv[ell] - lambdaleftv[ell];
        ellv.assign(2 * n + 1, -1);
        for (size_t ell = 0; ell < n; ++ell) ellv[2 * ell + 1] = static_cast<int32_t>(ell);
    }

    int WavelengthGrid::bin(double lambda) const
    {
        size_t index = std::upper_bound(borderv.begin(), borderv.end(), lambda) - borderv.begin();
        return ellv[index];
    }
}
