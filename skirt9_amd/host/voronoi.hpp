// voronoi.hpp -- Voronoi tessellation of a box for VoronoiMeshSpatialGrid (BASELINE configs[4]).
//
// The reference builds its cells with the vendored Voro++ library (SKIRT/core/VoronoiMeshSnapshot.cpp:491-733).  What
// the photon loop consumes of a cell is (VoronoiMeshSnapshot.cpp:1058-1188, 1006-1040, 976-989):
//   * the site positions, in the reference's order: sites outside the domain and near-duplicates removed, then sorted by
//     x coordinate (:494-526) -- the cell index m is the position in that order;
//   * the neighbour list of every cell: the sites whose bisecting plane carries a face of the cell, and the domain walls
//     the cell touches as negative ids (-1 xmin, -2 xmax, -3 ymin, -4 ymax, -5 zmin, -6 zmax, the Voro++ convention);
//   * the bounding box of the cell (rejection sampling of random positions in the cell) and its volume;
//   * a block grid with, per block, the cells whose bounding box overlaps the block (nearest-site search).
// This file computes the same quantities with its own algorithm: every cell is the domain box clipped by the bisecting
// planes of the surrounding sites in order of increasing distance, until no remaining site can cut it (distance > twice
// the largest vertex radius).  Neighbour SETS agree with Voro++ up to faces of negligible area; list ORDER differs (it
// only decides exact ties between exit distances).  Bounding boxes and volumes agree to rounding, not bit for bit -- so
// cell densities sampled from random positions in a cell follow the reference statistically, not bitwise.
#ifndef SKH_VORONOI_HPP
#define SKH_VORONOI_HPP

#include "mathutil.hpp"
#include <cstdint>
#include <vector>

namespace skh
{
    class VoronoiMesh
    {
    public:
        // sites as configured (any order); applies the reference's filtering and ordering, then builds the cells
        // relax: one relaxation step first (VoronoiMeshSpatialGrid::relaxSites): every site moves to the centroid of its cell
        void build(const Box& extent, std::vector<Vec3> sites, bool relax = false);

        int numCells() const { return static_cast<int>(_sites.size()); }
        const Box& extent() const { return _extent; }
        double eps() const { return _eps; }
        Vec3 site(int m) const { return _sites[m]; }
        const Box& cellBox(int m) const { return _boxes[m]; }
        double volume(int m) const { return _volumes[m]; }
        // VoronoiMeshSnapshot::cellIndex (:1006-1040): the cell whose site is nearest, -1 outside the domain
        int cellIndex(Vec3 r) const;
        // VoronoiMeshSnapshot::isPointClosestTo
        bool isPointClosestTo(Vec3 r, int m) const;

        // flattened tables (the pmc_grid Voronoi members)
        const std::vector<double>& flatSites() const { return _flatSites; }         // 3 * numCells
        const std::vector<int32_t>& nbrStart() const { return _nbrStart; }          // numCells + 1
        const std::vector<int32_t>& nbrList() const { return _nbrList; }            // site index, or -1..-6 for a wall
        int numBlocks() const { return _nb; }
        const std::vector<int32_t>& blockStart() const { return _blockStart; }      // nb^3 + 1
        const std::vector<int32_t>& blockList() const { return _blockList; }

    private:
        Box _extent;
        double _eps{0};
        std::vector<Vec3> _sites;
        std::vector<Box> _boxes;
        std::vector<double> _volumes;
        std::vector<Vec3> _centroids;  // of every cell, relative to its site
        std::vector<double> _flatSites;
        std::vector<int32_t> _nbrStart, _nbrList;
        int _nb{0};
        std::vector<int32_t> _blockStart, _blockList;
    };
}

#endif
