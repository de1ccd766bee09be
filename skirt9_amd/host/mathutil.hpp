// mathutil.hpp -- numerical helpers of the host model layer.
//
// Restated so that tables built at setup time come out bit-identical to the reference's (same formulas, same
// operation order, IEEE double, glibc libm):
//   tab::     bracketing, interpolation, clamped resampling, cumulative         SKIRT/utils/NR.hpp:130-190,203-209,
//             distributions                                                     300-306,328-362,373-436,446-475; NR.cpp:25-54
//   special:: lngamma, gamma, gln, gexp                                         SKIRT/utils/SpecialFunctions.cpp:11-30,798-836
//   Random: mt19937_64 + uniform_real_distribution(nextafter(0,1), 1)          SKIRT/core/Random.cpp:18-54,70-73
#ifndef SKH_MATHUTIL_HPP
#define SKH_MATHUTIL_HPP

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <random>
#include <thread>
#include <vector>

namespace skh
{
    using Array = std::vector<double>;

    struct Vec3
    {
        double x{0}, y{0}, z{0};
    };

    struct Box
    {
        double xmin{0}, ymin{0}, zmin{0}, xmax{0}, ymax{0}, zmax{0};
        Box() {}
        Box(double x0, double y0, double z0, double x1, double y1, double z1)
            : xmin(x0), ymin(y0), zmin(z0), xmax(x1), ymax(y1), zmax(z1)
        {}
        bool contains(double x, double y, double z) const
        {
            return x >= xmin && x <= xmax && y >= ymin && y <= ymax && z >= zmin && z <= zmax;
        }
        double volume() const { return (xmax - xmin) * (ymax - ymin) * (zmax - zmin); }
        double diagonal() const
        {
            return sqrt((xmax - xmin) * (xmax - xmin) + (ymax - ymin) * (ymax - ymin) + (zmax - zmin) * (zmax - zmin));
        }
        Vec3 center() const { return Vec3{0.5 * (xmin + xmax), 0.5 * (ymin + ymax), 0.5 * (zmin + zmax)}; }
        Vec3 fracPos(double xf, double yf, double zf) const
        {
            return Vec3{xmin + xf * (xmax - xmin), ymin + yf * (ymax - ymin), zmin + zf * (zmax - zmin)};
        }
    };

    // ---- ordered tables: bracketing, interpolation, resampling.  Semantics of the reference's table helpers
    // (SKIRT/utils/NR.hpp:130-190,203-209,300-306,328-362,373-436,446-475), which the setup-time tables depend on.
    namespace tab
    {
        // largest index i in [-1, limit) with table[i] <= x (table ascending); -1 if x lies below the first entry
        inline int lastNotAbove(const Array& table, double x, int limit)
        {
            int below = -1, above = limit;
            while (above - below > 1)
            {
                const int middle = (above + below) >> 1;
                (x < table[middle] ? above : below) = middle;
            }
            return below;
        }
        // interval [table[i], table[i+1]) that holds x; the last interval is closed; -1 below, size-1 above the table
        inline int bracket(const Array& table, double x)
        {
            const int count = static_cast<int>(table.size());
            return x == table[count - 1] ? count - 2 : lastNotAbove(table, x, count);
        }
        // the same, with values outside the table assigned to the first / last interval
        inline int bracketClipped(const Array& table, double x)
        {
            const int count = static_cast<int>(table.size());
            return x < table[0] ? 0 : lastNotAbove(table, x, count - 1);
        }
        // the same, but -1 for a value above the table
        inline int bracketOrMiss(const Array& table, double x)
        {
            const int count = static_cast<int>(table.size());
            return x > table[count - 1] ? -1 : lastNotAbove(table, x, count - 1);
        }
        // count + 1 equidistant points from first to last; returns the spacing
        inline double linearGrid(Array& points, double first, double last, int count)
        {
            const double spacing = (last - first) / count;
            points.resize(count + 1);
            for (int k = 0; k <= count; ++k) points[k] = first + k * spacing;
            return spacing;
        }
        // count + 1 points from first to last, equidistant in the logarithm
        inline void logGrid(Array& points, double first, double last, int count)
        {
            const double origin = log(first);
            const double spacing = log(last / first) / count;
            points.resize(count + 1);
            for (int k = 0; k <= count; ++k) points[k] = exp(origin + k * spacing);
        }
        // interpolation between (x1, f1) and (x2, f2): linear in both, logarithmic in x, logarithmic in both
        inline double linLin(double x, double x1, double x2, double f1, double f2)
        {
            const double fraction = (x - x1) / (x2 - x1);
            return f1 + fraction * (f2 - f1);
        }
        inline double logLin(double x, double x1, double x2, double f1, double f2)
        {
            if (!(x1 > 0 && x2 > 0)) return 0;
            const double fraction = log(x / x1) / log(x2 / x1);
            return f1 + fraction * (f2 - f1);
        }
        inline double logLog(double x, double x1, double x2, double f1, double f2)
        {
            if (f1 > 0 && f2 > 0)
            {
                const double fraction = log(x / x1) / log(x2 / x1);
                return f1 * exp(fraction * (log(f2 / f1)));
            }
            // a vanishing end point: exact at the end points, zero in between
            return x == x1 ? f1 : x == x2 ? f2 : 0;
        }
        // value at x of the table (xs, ys) interpolated with Rule, held constant beyond its ends
        template<double Rule(double, double, double, double, double)>
        inline double clampedAt(double x, const Array& xs, const Array& ys)
        {
            const int last = static_cast<int>(xs.size()) - 1;
            const int i = bracket(xs, x);
            if (i < 0) return ys[0];
            if (i >= last) return ys[last];
            return Rule(x, xs[i], xs[i + 1], ys[i], ys[i + 1]);
        }
        template<double Rule(double, double, double, double, double)>
        inline Array resampleClamped(const Array& targets, const Array& xs, const Array& ys)
        {
            Array values(targets.size());
            for (size_t k = 0; k != targets.size(); ++k) values[k] = clampedAt<Rule>(targets[k], xs, ys);
            return values;
        }
    }

    // ---- special functions (SKIRT/utils/SpecialFunctions.cpp:11-30,798-836)
    namespace special
    {
        // ln Gamma(a) by the six-term Lanczos series
        inline double lngamma(double a)
        {
            static const double lanczos[6] = {76.18009172947146,     -86.50532032941677,   24.01409824083091, -1.231739572450155,
                                              0.1208650973866179e-2, -0.5395239384953e-5};
            double shifted = a + 5.5;
            shifted -= (a + 0.5) * log(shifted);
            double series = 1.000000000190015;
            double denominator = a;
            for (double coefficient : lanczos) series += coefficient / ++denominator;
            const double rootTwoPi = 2.5066282746310005;
            return -shifted + log(rootTwoPi * series / a);
        }
        inline double gamma(double a) { return exp(lngamma(a)); }
        // generalised logarithm (x^(1-p) - 1)/(1-p), continuous through p = 1
        inline double gln(double p, double x)
        {
            const double q = 1.0 - p;
            if (q == 0.0) return log(x);
            if (fabs(q) >= 1e-3) return (pow(x, q) - 1.0) / q;
            const double logx = log(x);
            const double t = q * logx;
            return logx * (1.0 + 0.5 * t + 1.0 / 6.0 * t * t + 1.0 / 24.0 * t * t * t);
        }
        // its inverse, the generalised exponential (1 + (1-p) x)^(1/(1-p))
        inline double gexp(double p, double x)
        {
            const double q = 1.0 - p;
            if (q == 0.0) return exp(x);
            if (fabs(q) >= 1e-3) return pow(1.0 + q * x, 1.0 / q);
            const double xsq = x * x;
            const double first = 0.5 * xsq * q;
            const double second = 1.0 / 24.0 * x * xsq * (8.0 + 3.0 * x) * q * q;
            const double third = 1.0 / 48.0 * xsq * xsq * (12.0 + 8.0 * x + xsq) * q * q * q;
            return exp(x) * (1.0 - first + second - third);
        }
    }

    namespace tab
    {
        // cumulative distribution of the tabulated density (xs, density) with linear or power-law segments (NR.cpp:25-54):
        // fills the normalised cumulative table, normalises the density in place, returns the norm
        inline double cumulative(bool powerLaw, const Array& xs, Array& density, Array& cumul)
        {
            const size_t segments = xs.size() - 1;
            cumul.assign(segments + 1, 0.);
            for (size_t k = 0; k != segments; ++k)
            {
                const double left = density[k], right = density[k + 1];
                double piece = 0.;
                if (!powerLaw)
                {
                    const double mean = 0.5 * (left + right);
                    piece = mean * (xs[k + 1] - xs[k]);
                }
                else if (left > 0 && right > 0)
                {
                    const double ratio = xs[k + 1] / xs[k];
                    const double slope = log(right / left) / log(ratio);
                    piece = left * xs[k] * special::gln(-slope, ratio);
                }
                cumul[k + 1] = cumul[k] + piece;
            }
            const double norm = cumul[segments];
            if (norm > 0.)
            {
                for (double& value : density) value /= norm;
                for (double& value : cumul) value /= norm;
            }
            return norm;
        }
    }

    // Runs body(begin, end) over [0, n) on the host cores (SKH_THREADS overrides the count).  Used for the density
    // evaluations of the setup phase only: the sample POSITIONS are drawn sequentially from the one random stream, in the
    // reference's order, and every evaluation is independent, so the results do not depend on the thread count.
    inline void parallelFor(size_t n, const std::function<void(size_t, size_t)>& body)
    {
        size_t threads = std::thread::hardware_concurrency();
        if (const char* env = std::getenv("SKH_THREADS")) threads = static_cast<size_t>(std::max(1, atoi(env)));
        threads = std::max<size_t>(1, std::min<size_t>(threads, n / 256));
        if (threads <= 1)
        {
            body(0, n);
            return;
        }
        std::vector<std::thread> pool;
        const size_t chunk = (n + threads - 1) / threads;
        for (size_t t = 0; t != threads; ++t)
        {
            const size_t b = t * chunk, e = std::min(n, b + chunk);
            if (b < e) pool.emplace_back(body, b, e);
        }
        for (auto& th : pool) th.join();
    }

    // The parent-thread generator of the reference: seeded from <Random seed="..."/> (Random.cpp:40-46), one
    // 64-bit draw per uniform deviate in ]0,1[.  Setup-time sampling (tree construction, cell densities) and the
    // single-thread photon loop consume ONE stream in program order; the draw counter lets a caller continue it.
    class Random
    {
    public:
        explicit Random(int seed = 0) { setSeed(seed); }
        void setSeed(int seed)
        {
            // the eight words the reference offsets by the seed (Random.cpp:40-46)
            static const unsigned int words[8] = {979364188u, 871244425u, 1693909487u, 1290454318u,
                                                  210509498u, 542237529u, 3429911442u, 3321294726u};
            unsigned int keyed[8];
            for (int k = 0; k != 8; ++k) keyed[k] = words[k] + seed;
            std::seed_seq sequence(keyed, keyed + 8);
            engine_.seed(sequence);
            seed_ = seed;
            drawn_ = 0;
        }
        double uniform()
        {
            ++drawn_;
            return unit_(engine_);
        }
        Vec3 position(const Box& box)
        {
            const double fx = uniform();
            const double fy = uniform();
            const double fz = uniform();
            return box.fracPos(fx, fy, fz);
        }
        // isotropic direction: polar angle from the first deviate, azimuth from the second (Random.cpp:121-126), as a
        // unit vector with the poles snapped (Direction.cpp:11-38)
        Vec3 direction()
        {
            const double polar = acos(2.0 * uniform() - 1.0);
            const double azimuth = 2.0 * M_PI * uniform();
            const double snap = 1e-8;
            if (polar <= snap) return Vec3{0, 0, 1};
            if (polar >= M_PI - snap) return Vec3{0, 0, -1};
            const double sine = sin(polar);
            return Vec3{sine * cos(azimuth), sine * sin(azimuth), cos(polar)};
        }
        int seed() const { return seed_; }
        unsigned long long draws() const { return drawn_; }

    private:
        std::mt19937_64 engine_;
        std::uniform_real_distribution<double> unit_{std::nextafter(0., 1.), 1.};
        int seed_{0};
        unsigned long long drawn_{0};
    };
}

#endif
