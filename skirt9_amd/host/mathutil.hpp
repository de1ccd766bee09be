// mathutil.hpp -- numerical helpers of the host model layer.
//
// Restated so that tables built at setup time come out bit-identical to the reference's (same formulas, same
// operation order, IEEE double, glibc libm):
//   locate / locateClip / locateFail, interpolation, clamped resampling, cdf   SKIRT/utils/NR.hpp:130-190,203-209,
//                                                                             300-306,328-362,373-436,446-475
//   cdf2 (log-log / lin-lin cumulative)                                        SKIRT/utils/NR.cpp:25-54
//   lngamma, gamma, gln, gexp                                                  SKIRT/utils/SpecialFunctions.cpp:11-30,798-836
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

    namespace nr
    {
        inline int locateBasic(const Array& xv, double x, int n)
        {
            int jl = -1;
            int ju = n;
            while (ju - jl > 1)
            {
                int jm = (ju + jl) >> 1;
                if (x < xv[jm])
                    ju = jm;
                else
                    jl = jm;
            }
            return jl;
        }
        inline int locate(const Array& xv, double x)
        {
            int n = static_cast<int>(xv.size());
            if (x == xv[n - 1]) return n - 2;
            return locateBasic(xv, x, n);
        }
        inline int locateClip(const Array& xv, double x)
        {
            int n = static_cast<int>(xv.size());
            if (x < xv[0]) return 0;
            return locateBasic(xv, x, n - 1);
        }
        inline int locateFail(const Array& xv, double x)
        {
            int n = static_cast<int>(xv.size());
            if (x > xv[n - 1]) return -1;
            return locateBasic(xv, x, n - 1);
        }
        inline double linearGrid(Array& xv, double xmin, double xmax, int n)
        {
            xv.resize(n + 1);
            double dx = (xmax - xmin) / n;
            for (int i = 0; i <= n; i++) xv[i] = xmin + i * dx;
            return dx;
        }
        inline void logGrid(Array& xv, double xmin, double xmax, int n)
        {
            xv.resize(n + 1);
            double logxmin = log(xmin);
            double dlogx = log(xmax / xmin) / n;
            for (int i = 0; i <= n; i++) xv[i] = exp(logxmin + i * dlogx);
        }
        inline double interpolateLinLin(double x, double x1, double x2, double f1, double f2)
        {
            return f1 + ((x - x1) / (x2 - x1)) * (f2 - f1);
        }
        inline double interpolateLogLin(double x, double x1, double x2, double f1, double f2)
        {
            if (x1 <= 0 || x2 <= 0) return 0;
            return f1 + log(x / x1) / log(x2 / x1) * (f2 - f1);
        }
        inline double interpolateLogLog(double x, double x1, double x2, double f1, double f2)
        {
            if (f1 <= 0 || f2 <= 0)
            {
                if (x == x1) return f1;
                if (x == x2) return f2;
                return 0;
            }
            return f1 * exp(log(x / x1) / log(x2 / x1) * (log(f2 / f1)));
        }
        template<double F(double, double, double, double, double)>
        inline double clampedValue(double x, const Array& xv, const Array& yv)
        {
            int n = static_cast<int>(xv.size());
            int i = locate(xv, x);
            if (i < 0) return yv[0];
            if (i >= n - 1) return yv[n - 1];
            return F(x, xv[i], xv[i + 1], yv[i], yv[i + 1]);
        }
        template<double F(double, double, double, double, double)>
        inline Array clampedResample(const Array& xresv, const Array& xoriv, const Array& yoriv)
        {
            Array yresv(xresv.size());
            for (size_t l = 0; l < xresv.size(); l++) yresv[l] = clampedValue<F>(xresv[l], xoriv, yoriv);
            return yresv;
        }
    }

    namespace special
    {
        inline double lngamma(double a)
        {
            static const double cof[6] = {76.18009172947146,  -86.50532032941677,    24.01409824083091,
                                          -1.231739572450155, 0.1208650973866179e-2, -0.5395239384953e-5};
            double xx, y, tmp, ser;
            y = xx = a;
            tmp = xx + 5.5;
            tmp -= (xx + 0.5) * log(tmp);
            ser = 1.000000000190015;
            for (int j = 0; j < 6; j++) ser += cof[j] / ++y;
            return -tmp + log(2.5066282746310005 * ser / xx);
        }
        inline double gamma(double a) { return exp(lngamma(a)); }
        inline double gln(double p, double x)
        {
            const double q = 1.0 - p;
            if (q == 0.0) return log(x);
            if (fabs(q) < 1e-3)
            {
                double lnx = log(x);
                double s = q * lnx;
                return lnx * (1.0 + 0.5 * s + 1.0 / 6.0 * s * s + 1.0 / 24.0 * s * s * s);
            }
            return (pow(x, q) - 1.0) / q;
        }
        inline double gexp(double p, double x)
        {
            const double q = 1.0 - p;
            if (q == 0.0) return exp(x);
            if (fabs(q) < 1e-3)
            {
                double x2 = x * x;
                return exp(x)
                       * (1.0 - 0.5 * x2 * q + 1.0 / 24.0 * x * x2 * (8.0 + 3.0 * x) * q * q
                          - 1.0 / 48.0 * x2 * x2 * (12.0 + 8.0 * x + x2) * q * q * q);
            }
            return pow(1.0 + q * x, 1.0 / q);
        }
    }

    namespace nr
    {
        // NR::cdf2 (NR.cpp:25-54): builds the normalised cumulative distribution, normalises pv in place
        inline double cdf2(bool loglog, const Array& xv, Array& pv, Array& Pv)
        {
            size_t n = xv.size() - 1;
            Pv.assign(n + 1, 0.);
            for (size_t i = 0; i != n; ++i)
            {
                double area = 0.;
                if (!loglog)
                    area = 0.5 * (pv[i] + pv[i + 1]) * (xv[i + 1] - xv[i]);
                else if (pv[i] > 0 && pv[i + 1] > 0)
                {
                    double alpha = log(pv[i + 1] / pv[i]) / log(xv[i + 1] / xv[i]);
                    area = pv[i] * xv[i] * special::gln(-alpha, xv[i + 1] / xv[i]);
                }
                Pv[i + 1] = Pv[i] + area;
            }
            double norm = Pv[n];
            if (norm > 0.)
            {
                for (auto& p : pv) p /= norm;
                for (auto& P : Pv) P /= norm;
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
            std::seed_seq seedseq{979364188u + seed, 871244425u + seed, 1693909487u + seed, 1290454318u + seed,
                                  210509498u + seed, 542237529u + seed, 3429911442u + seed, 3321294726u + seed};
            _generator.seed(seedseq);
            _seed = seed;
            _draws = 0;
        }
        double uniform()
        {
            ++_draws;
            return _distribution(_generator);
        }
        Vec3 position(const Box& box)
        {
            double x = uniform();
            double y = uniform();
            double z = uniform();
            return box.fracPos(x, y, z);
        }
        // Random::direction() (Random.cpp:121-126) with Direction(theta, phi) (Direction.cpp:11-38)
        Vec3 direction()
        {
            double theta = acos(2.0 * uniform() - 1.0);
            double phi = 2.0 * M_PI * uniform();
            const double eps = 1e-8;
            if (theta <= eps) return Vec3{0, 0, 1};
            if (theta >= M_PI - eps) return Vec3{0, 0, -1};
            double sintheta = sin(theta);
            return Vec3{sintheta * cos(phi), sintheta * sin(phi), cos(theta)};
        }
        int seed() const { return _seed; }
        unsigned long long draws() const { return _draws; }

    private:
        std::mt19937_64 _generator;
        std::uniform_real_distribution<double> _distribution{std::nextafter(0., 1.), 1.};
        int _seed{0};
        unsigned long long _draws{0};
    };
}

#endif
