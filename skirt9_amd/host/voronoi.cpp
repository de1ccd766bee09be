// voronoi.cpp -- see voronoi.hpp
#include "voronoi.hpp"
#include <algorithm>
#include <cmath>
#include <map>
#include <stdexcept>

namespace skh
{
    namespace
    {
        inline Vec3 sub(Vec3 a, Vec3 b) { return Vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
        inline double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
        inline Vec3 cross(Vec3 a, Vec3 b) { return Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
        // (r - site).norm2() as Cell::squaredDistanceTo (VoronoiMeshSnapshot.cpp:158, Vec.hpp:67)
        inline double squaredDistance(Vec3 r, Vec3 s)
        {
            double x = r.x - s.x, y = r.y - s.y, z = r.z - s.z;
            return x * x + y * y + z * z;
        }

        // a convex polyhedron as faces with vertex loops; every face remembers the plane that made it
        struct Polyhedron
        {
            struct Face
            {
                int id;  // neighbour site, or -1..-6 for a domain wall
                std::vector<int> loop;
            };
            std::vector<Vec3> v;
            std::vector<Face> faces;

            void initBox(const Box& b)
            {
                v = {{b.xmin, b.ymin, b.zmin}, {b.xmax, b.ymin, b.zmin}, {b.xmax, b.ymax, b.zmin}, {b.xmin, b.ymax, b.zmin},
                     {b.xmin, b.ymin, b.zmax}, {b.xmax, b.ymin, b.zmax}, {b.xmax, b.ymax, b.zmax}, {b.xmin, b.ymax, b.zmax}};
                faces = {{-1, {0, 3, 7, 4}}, {-2, {1, 2, 6, 5}}, {-3, {0, 1, 5, 4}},
                         {-4, {3, 2, 6, 7}}, {-5, {0, 1, 2, 3}}, {-6, {4, 5, 6, 7}}};
            }

            // keeps the part with n.x <= c; returns false if the plane does not cut the polyhedron
            bool clip(Vec3 n, double c, int id, double tol)
            {
                const size_t nv = v.size();
                std::vector<double> d(nv);
                bool anyOut = false;
                for (const Face& f : faces)
                    for (int i : f.loop)
                    {
                        d[i] = dot(n, v[i]) - c;
                        if (d[i] > tol) anyOut = true;
                    }
                if (!anyOut) return false;
                std::map<std::pair<int, int>, int> cut;  // crossing edge -> vertex on the plane
                auto onPlane = [&](int a, int b) {       // a inside, b outside
                    if (d[a] >= -tol) return a;          // a lies on the plane already
                    auto key = std::make_pair(std::min(a, b), std::max(a, b));
                    auto it = cut.find(key);
                    if (it != cut.end()) return it->second;
                    double t = d[a] / (d[a] - d[b]);
                    Vec3 p{v[a].x + t * (v[b].x - v[a].x), v[a].y + t * (v[b].y - v[a].y), v[a].z + t * (v[b].z - v[a].z)};
                    v.push_back(p);
                    int idx = static_cast<int>(v.size()) - 1;
                    cut[key] = idx;
                    return idx;
                };
                std::vector<Face> kept;
                std::vector<std::pair<int, int>> segments;  // the new face's edges
                for (const Face& f : faces)
                {
                    Face g;
                    g.id = f.id;
                    int first = -1, second = -1;
                    const size_t m = f.loop.size();
                    for (size_t e = 0; e != m; ++e)
                    {
                        int a = f.loop[e], b = f.loop[(e + 1) % m];
                        bool ina = d[a] <= tol, inb = d[b] <= tol;
                        if (ina && (g.loop.empty() || g.loop.back() != a)) g.loop.push_back(a);
                        if (ina != inb)
                        {
                            int p = ina ? onPlane(a, b) : onPlane(b, a);
                            if (g.loop.empty() || g.loop.back() != p) g.loop.push_back(p);
                            (first < 0 ? first : second) = p;
                        }
                    }
                    while (g.loop.size() > 1 && g.loop.front() == g.loop.back()) g.loop.pop_back();
                    if (first >= 0 && second >= 0 && first != second) segments.emplace_back(first, second);
                    if (g.loop.size() >= 3) kept.push_back(std::move(g));
                }
                // chain the segments into the loop of the new face
                if (segments.size() >= 3)
                {
                    Face cap;
                    cap.id = id;
                    std::vector<char> used(segments.size(), 0);
                    cap.loop.push_back(segments[0].first);
                    int cur = segments[0].second;
                    used[0] = 1;
                    for (size_t step = 1; step < segments.size(); ++step)
                    {
                        cap.loop.push_back(cur);
                        bool found = false;
                        for (size_t s = 0; s != segments.size() && !found; ++s)
                            if (!used[s])
                            {
                                if (segments[s].first == cur)
                                {
                                    cur = segments[s].second;
                                    used[s] = 1;
                                    found = true;
                                }
                                else if (segments[s].second == cur)
                                {
                                    cur = segments[s].first;
                                    used[s] = 1;
                                    found = true;
                                }
                            }
                        if (!found) break;
                    }
                    if (cap.loop.size() >= 3) kept.push_back(std::move(cap));
                }
                faces.swap(kept);
                return true;
            }

            double maxRadius2(Vec3 c) const
            {
                double r2 = 0.;
                for (const Face& f : faces)
                    for (int i : f.loop) r2 = std::max(r2, squaredDistance(v[i], c));
                return r2;
            }
        };
    }

    void VoronoiMesh::build(const Box& extent, std::vector<Vec3> sites, bool relax)
    {
        _extent = extent;
        _eps = 1e-12 * extent.diagonal();  // VoronoiMeshSnapshot::setExtent (:393-397)
        // ---- VoronoiMeshSnapshot::buildMesh (:494-526): drop sites outside the domain, sort by x, drop near-duplicates
        sites.erase(std::remove_if(sites.begin(), sites.end(), [&](Vec3 p) { return !extent.contains(p.x, p.y, p.z); }), sites.end());
        std::sort(sites.begin(), sites.end(), [](Vec3 a, Vec3 b) { return a.x < b.x; });
        {
            const int n = static_cast<int>(sites.size());
            std::vector<char> drop(n, 0);
            for (int m = 0; m != n; ++m)
                for (int j = m + 1; j != n && sites[j].x - sites[m].x < _eps; ++j)
                    if (squaredDistance(sites[j], sites[m]) < _eps * _eps)
                    {
                        drop[m] = 1;
                        break;
                    }
            std::vector<Vec3> kept;
            for (int m = 0; m != n; ++m)
                if (!drop[m]) kept.push_back(sites[m]);
            sites.swap(kept);
        }
        _sites = sites;
        const int N = numCells();
        if (N <= 0) throw std::runtime_error("Voronoi grid without sites inside the domain");
        // ---- VoronoiMeshSnapshot::buildMesh with relaxSites (:550-601): ONE relaxation step -- every site moves to the centroid of its cell in
        // the tessellation of the sites as given (order, and thus the cell indices, stay as they are); the final tessellation follows below.
        // (The reference takes the centroid from Voro++, voronoicell_base::centroid; this one sums the same tetrahedra over the faces of its own
        // polyhedron: the relaxed sites agree with the reference's to rounding, not bit for bit.)
        if (relax)
        {
            std::vector<Vec3> moved(_sites);
            build(extent, std::move(sites), false);  // (the tessellation of the sites as given; fills _centroids)
            for (int m = 0; m != N; ++m) moved[m] = Vec3{moved[m].x + _centroids[m].x, moved[m].y + _centroids[m].y, moved[m].z + _centroids[m].z};
            _sites = moved;
        }
        _centroids.assign(N, Vec3{0., 0., 0.});

        // ---- bucket grid over the sites for the neighbour candidates
        const int nbk = std::max(1, static_cast<int>(std::cbrt(N / 2.0)));
        const double wx = (extent.xmax - extent.xmin) / nbk, wy = (extent.ymax - extent.ymin) / nbk, wz = (extent.zmax - extent.zmin) / nbk;
        const double wmin = std::min(wx, std::min(wy, wz));
        auto bucketOf = [&](Vec3 p, int& i, int& j, int& k) {
            i = std::max(0, std::min(nbk - 1, static_cast<int>((p.x - extent.xmin) / wx)));
            j = std::max(0, std::min(nbk - 1, static_cast<int>((p.y - extent.ymin) / wy)));
            k = std::max(0, std::min(nbk - 1, static_cast<int>((p.z - extent.zmin) / wz)));
        };
        std::vector<std::vector<int>> buckets(static_cast<size_t>(nbk) * nbk * nbk);
        for (int m = 0; m != N; ++m)
        {
            int i, j, k;
            bucketOf(_sites[m], i, j, k);
            buckets[(static_cast<size_t>(i) * nbk + j) * nbk + k].push_back(m);
        }

        // ---- the cells
        _boxes.assign(N, Box());
        _volumes.assign(N, 0.);
        std::vector<std::vector<int32_t>> nbrs(N);
        const double tol = 1e-11 * extent.diagonal();
        parallelFor(N, [&](size_t mb, size_t me) {
            Polyhedron P;
            std::vector<std::pair<double, int>> shell;
            for (size_t mm = mb; mm != me; ++mm)
            {
                const int m = static_cast<int>(mm);
                const Vec3 pr = _sites[m];
                P.initBox(extent);
                int bi, bj, bk;
                bucketOf(pr, bi, bj, bk);
                double rmax2 = P.maxRadius2(pr);
                for (int s = 0; s <= nbk; ++s)
                {
                    // every site in shell s is at least (s - 1) bucket widths away
                    if (s >= 1 && (s - 1) * wmin > 2. * std::sqrt(rmax2)) break;
                    shell.clear();
                    for (int i = std::max(0, bi - s); i <= std::min(nbk - 1, bi + s); ++i)
                        for (int j = std::max(0, bj - s); j <= std::min(nbk - 1, bj + s); ++j)
                            for (int k = std::max(0, bk - s); k <= std::min(nbk - 1, bk + s); ++k)
                            {
                                if (std::max(std::abs(i - bi), std::max(std::abs(j - bj), std::abs(k - bk))) != s) continue;
                                for (int q : buckets[(static_cast<size_t>(i) * nbk + j) * nbk + k])
                                    if (q != m) shell.emplace_back(squaredDistance(_sites[q], pr), q);
                            }
                    std::sort(shell.begin(), shell.end());
                    for (const auto& cand : shell)
                    {
                        if (cand.first > 4. * rmax2) break;  // farther than twice the largest vertex radius: cannot cut
                        const Vec3 pi = _sites[cand.second];
                        const Vec3 n = sub(pi, pr);
                        const double len = std::sqrt(cand.first);
                        const Vec3 nu{n.x / len, n.y / len, n.z / len};
                        const Vec3 mid{0.5 * (pi.x + pr.x), 0.5 * (pi.y + pr.y), 0.5 * (pi.z + pr.z)};
                        if (P.clip(nu, dot(nu, mid), cand.second, tol)) rmax2 = P.maxRadius2(pr);
                    }
                }
                // neighbours, bounding box, volume, centroid (relative to the site: the sum over the tetrahedra site - face fan)
                Box bb(DBL_MAX, DBL_MAX, DBL_MAX, -DBL_MAX, -DBL_MAX, -DBL_MAX);
                double vol = 0.;
                Vec3 moment{0., 0., 0.};
                for (const auto& f : P.faces)
                {
                    nbrs[m].push_back(f.id);
                    const Vec3 a = sub(P.v[f.loop[0]], pr);
                    for (size_t e = 0; e != f.loop.size(); ++e)
                    {
                        const Vec3 p = P.v[f.loop[e]];
                        bb.xmin = std::min(bb.xmin, p.x), bb.xmax = std::max(bb.xmax, p.x);
                        bb.ymin = std::min(bb.ymin, p.y), bb.ymax = std::max(bb.ymax, p.y);
                        bb.zmin = std::min(bb.zmin, p.z), bb.zmax = std::max(bb.zmax, p.z);
                        if (e >= 1 && e + 1 < f.loop.size())
                        {
                            const Vec3 b = sub(P.v[f.loop[e]], pr), c = sub(P.v[f.loop[e + 1]], pr);
                            const double tet = std::abs(dot(a, cross(b, c))) / 6.;
                            vol += tet;
                            moment.x += tet * (a.x + b.x + c.x), moment.y += tet * (a.y + b.y + c.y), moment.z += tet * (a.z + b.z + c.z);
                        }
                    }
                }
                _boxes[m] = bb;
                _volumes[m] = vol;
                if (vol > 0.) _centroids[m] = Vec3{0.25 * moment.x / vol, 0.25 * moment.y / vol, 0.25 * moment.z / vol};
            }
        });

        // ---- flattened tables
        _flatSites.resize(3 * static_cast<size_t>(N));
        _nbrStart.assign(N + 1, 0);
        _nbrList.clear();
        for (int m = 0; m != N; ++m)
        {
            _flatSites[3 * m] = _sites[m].x, _flatSites[3 * m + 1] = _sites[m].y, _flatSites[3 * m + 2] = _sites[m].z;
            _nbrStart[m] = static_cast<int32_t>(_nbrList.size());
            _nbrList.insert(_nbrList.end(), nbrs[m].begin(), nbrs[m].end());
        }
        _nbrStart[N] = static_cast<int32_t>(_nbrList.size());

        // ---- search blocks (VoronoiMeshSnapshot::buildSearchPerBlock, :765-787): per block the cells whose bounding box
        //      overlaps it
        _nb = std::max(3, std::min(250, static_cast<int>(std::cbrt(N))));
        auto blockIndices = [&](Vec3 r, int& i, int& j, int& k) {  // Box::cellIndices (Box.hpp:171-176)
            i = std::max(0, std::min(_nb - 1, static_cast<int>(_nb * (r.x - extent.xmin) / (extent.xmax - extent.xmin))));
            j = std::max(0, std::min(_nb - 1, static_cast<int>(_nb * (r.y - extent.ymin) / (extent.ymax - extent.ymin))));
            k = std::max(0, std::min(_nb - 1, static_cast<int>(_nb * (r.z - extent.zmin) / (extent.zmax - extent.zmin))));
        };
        std::vector<std::vector<int32_t>> lists(static_cast<size_t>(_nb) * _nb * _nb);
        for (int m = 0; m != N; ++m)
        {
            int i1, j1, k1, i2, j2, k2;
            blockIndices(Vec3{_boxes[m].xmin - _eps, _boxes[m].ymin - _eps, _boxes[m].zmin - _eps}, i1, j1, k1);
            blockIndices(Vec3{_boxes[m].xmax + _eps, _boxes[m].ymax + _eps, _boxes[m].zmax + _eps}, i2, j2, k2);
            for (int i = i1; i <= i2; i++)
                for (int j = j1; j <= j2; j++)
                    for (int k = k1; k <= k2; k++) lists[(static_cast<size_t>(i) * _nb + j) * _nb + k].push_back(m);
        }
        _blockStart.assign(lists.size() + 1, 0);
        _blockList.clear();
        for (size_t b = 0; b != lists.size(); ++b)
        {
            _blockStart[b] = static_cast<int32_t>(_blockList.size());
            _blockList.insert(_blockList.end(), lists[b].begin(), lists[b].end());
        }
        _blockStart[lists.size()] = static_cast<int32_t>(_blockList.size());
    }

    int VoronoiMesh::cellIndex(Vec3 r) const
    {
        if (!_extent.contains(r.x, r.y, r.z)) return -1;
        int i = std::max(0, std::min(_nb - 1, static_cast<int>(_nb * (r.x - _extent.xmin) / (_extent.xmax - _extent.xmin))));
        int j = std::max(0, std::min(_nb - 1, static_cast<int>(_nb * (r.y - _extent.ymin) / (_extent.ymax - _extent.ymin))));
        int k = std::max(0, std::min(_nb - 1, static_cast<int>(_nb * (r.z - _extent.zmin) / (_extent.zmax - _extent.zmin))));
        const size_t b = (static_cast<size_t>(i) * _nb + j) * _nb + k;
        int m = -1;
        double mdist = DBL_MAX;
        for (int32_t q = _blockStart[b]; q != _blockStart[b + 1]; ++q)
        {
            const int id = _blockList[q];
            double idist = squaredDistance(r, _sites[id]);
            if (idist < mdist)
            {
                m = id;
                mdist = idist;
            }
        }
        return m;
    }

    bool VoronoiMesh::isPointClosestTo(Vec3 r, int m) const
    {
        const double target = squaredDistance(r, _sites[m]);
        for (int32_t q = _nbrStart[m]; q != _nbrStart[m + 1]; ++q)
        {
            const int id = _nbrList[q];
            if (id >= 0 && squaredDistance(r, _sites[id]) < target) return false;
        }
        return true;
    }
}
