// pmc_tables.hip -- pmc_create's table builders: the reference-shaped grid tables of pmc_scene become the device layout of pmc_device.h.
//
// Octree: verify that every node box is consistent with one per-axis dyadic coordinate table (true for any tree built by recursive midpoint
// subdivision, OctTreeNode.cpp:22-33), build that table from the reference's own doubles, and replace the per-wall neighbour lists by links
// (per wall: the neighbour leaf covering it, or the internal node below which finer neighbours are found).  The reference lists themselves
// are uploaded too, re-indexed by cell, for the exact fallback path.
// Voronoi: cell records, neighbour entries, cone masks, and the tables of runs that the walk kernels read (per direction cone, per observer).
#include "pmc_context.h"

namespace
{
    // ---- octree flattening -------------------------------------------------------------------------
    struct TreeBuild
    {
        int lmax{0};
        int tabn{0};
        std::vector<double> table;       // [3][tabn]
        std::vector<LeafRec> leaves;     // by cell index m
        std::vector<CellRec> cells;      // by device cell index: the walk step's hot record
        std::vector<NodeRec> internals;  // by internal index
        std::vector<int32_t> nbrStart, nbrList;
        uint32_t rootLink{0};
        int coarseLevel{0};
        std::vector<uint32_t> coarse;    // [2^Lc]^3 (z, y, x): link of the covering node at level <= Lc
        // device numbering of the cells: dev = perm[m], depth-first order of the tree (the eight leaves of a node whose
        // children are all leaves are consecutive, in child order); cellExt[dev] = m (or -1 for padding); cellSlots =
        // entries per device table
        std::vector<int32_t> perm, cellExt;
        int cellSlots{0};
    };

    int buildTree(const pmc_grid& g, const double* density, TreeBuild& T)
    {
        const int numNodes = g.num_nodes;
        if (numNodes < 1) return fail(PMC_ERR_INVALID, "octree without nodes");
        int maxLevel = 0;
        for (int id = 0; id < numNodes; ++id) maxLevel = std::max(maxLevel, g.node_level[id]);
        if (maxLevel > PMC_MAX_LEVEL)
            return fail(PMC_ERR_UNSUPPORTED, "octree deeper than " + std::to_string(PMC_MAX_LEVEL) + " levels");
        T.lmax = maxLevel;
        T.tabn = (1 << maxLevel) + 1;
        const double unset = std::nan("");
        T.table.assign(3 * size_t(T.tabn), unset);

        // fine lower-corner indices of every node, from the topology alone
        std::vector<int32_t> fx(numNodes), fy(numNodes), fz(numNodes);
        std::vector<int32_t> internalIndex(numNodes, -1);
        int numInternal = 0;
        fx[0] = fy[0] = fz[0] = 0;
        if (g.node_level[0] != 0) return fail(PMC_ERR_INVALID, "octree root is not at level 0");
        for (int id = 0; id < numNodes; ++id)
        {
            int first = g.node_first_child[id];
            if (first < 0) continue;
            internalIndex[id] = numInternal++;
            if (first + 8 > numNodes) return fail(PMC_ERR_INVALID, "octree child index out of range");
            int half = 1 << (maxLevel - g.node_level[id] - 1);
            for (int l = 0; l < 8; ++l)
            {
                int c = first + l;
                if (g.node_level[c] != g.node_level[id] + 1) return fail(PMC_ERR_INVALID, "octree child level mismatch");
                fx[c] = fx[id] + ((l & 1) ? half : 0);
                fy[c] = fy[id] + ((l & 2) ? half : 0);
                fz[c] = fz[id] + ((l & 4) ? half : 0);
            }
        }
        // coordinate table from the reference's own box doubles, with consistency check
        auto put = [&](int axis, int index, double value) -> bool {
            double& slot = T.table[size_t(axis) * T.tabn + index];
            if (std::isnan(slot))
            {
                slot = value;
                return true;
            }
            return slot == value;
        };
        for (int id = 0; id < numNodes; ++id)
        {
            const double* b = g.node_box + 6 * size_t(id);
            int size = 1 << (maxLevel - g.node_level[id]);
            bool ok = put(0, fx[id], b[0]) && put(0, fx[id] + size, b[3]) && put(1, fy[id], b[1]) && put(1, fy[id] + size, b[4])
                      && put(2, fz[id], b[2]) && put(2, fz[id] + size, b[5]);
            if (!ok)
                return fail(PMC_ERR_UNSUPPORTED,
                            "octree node boxes are not consistent with a dyadic coordinate table (node " + std::to_string(id) + ")");
        }
        // entries that are no node's wall (inside coarse leaves) get the dyadic midpoints: no decision ever depends on
        // them, but the table becomes strictly monotonic, which the index search of topDown (pmc_walk.inc) relies on
        for (int axis = 0; axis < 3; ++axis)
            for (int size = 1 << maxLevel; size >= 2; size >>= 1)
                for (int lo = 0; lo + size <= (1 << maxLevel); lo += size)
                {
                    double& mid = T.table[size_t(axis) * T.tabn + lo + size / 2];
                    if (std::isnan(mid))
                        mid = (T.table[size_t(axis) * T.tabn + lo] + T.table[size_t(axis) * T.tabn + lo + size]) / 2.;
                }
        // device numbering of the cells: depth-first order of the tree, children in child order (tuning aids:
        // PMC_CELL_ORDER=ref keeps the caller's numbering -- breadth-first by level in SKIRT; PMC_CELL_SHUFFLE=g scatters
        // groups of 2^g cells of that numbering over the table)
        {
            const int n = g.num_cells;
            T.perm.assign(n, -1);
            const char* order = pmcTune("PMC_CELL_ORDER");
            int gb = -1;
            if (const char* env = pmcTune("PMC_CELL_SHUFFLE")) gb = atoi(env);
            if (gb >= 0 && gb <= 16)
            {
                const int groupSize = 1 << gb;
                const int numGroups = (n + groupSize - 1) / groupSize;
                std::vector<int32_t> where(numGroups);
                for (int i = 0; i < numGroups; ++i) where[i] = i;
                uint64_t state = 0x9E3779B97F4A7C15ull;  // fixed: the numbering is a pure function of the scene
                for (int i = numGroups - 1; i > 0; --i)
                {
                    state = state * 6364136223846793005ull + 1442695040888963407ull;
                    const int j = int((state >> 33) % uint64_t(i + 1));
                    std::swap(where[i], where[j]);
                }
                T.cellSlots = numGroups * groupSize;
                for (int m = 0; m < n; ++m) T.perm[m] = where[m >> gb] * groupSize + (m & (groupSize - 1));
            }
            else if (order && !strcmp(order, "ref"))
            {
                T.cellSlots = n;
                for (int m = 0; m < n; ++m) T.perm[m] = m;
            }
            else
            {
                T.cellSlots = n;
                int next = 0;
                std::vector<int> stack{0};
                while (!stack.empty())
                {
                    const int id = stack.back();
                    stack.pop_back();
                    const int first = g.node_first_child[id];
                    if (first < 0)
                    {
                        const int m = g.node_cell[id];
                        if (m < 0 || m >= n || T.perm[m] >= 0) return fail(PMC_ERR_INVALID, "octree leaf without a valid cell index");
                        T.perm[m] = next++;
                    }
                    else
                        for (int l = 7; l >= 0; --l) stack.push_back(first + l);
                }
                if (next != n) return fail(PMC_ERR_INVALID, "octree leaves and cells do not match");
            }
            T.cellExt.assign(T.cellSlots, -1);
            for (int m = 0; m < n; ++m) T.cellExt[T.perm[m]] = m;
        }
        // box code (pmc_walk.inc decodeBoxWords): the indices of the three lower walls in the coordinate table of their axis + size exponent
        // (e >= 15: the field holds 15 and the low bits of the x index, zero in such a box, hold e - 15)
        auto code = [&](int id) -> uint64_t {
            const uint64_t e = uint64_t(maxLevel - g.node_level[id]);
            uint64_t ix = uint64_t(fx[id]);
            const uint64_t iy = uint64_t(fy[id]), iz = uint64_t(fz[id]);
            if (e >= 15) ix |= e - 15;
            return ix | (iy << 20) | (iz << 40) | (std::min<uint64_t>(e, 15) << 60);
        };
        // link word (pmc_device.h): size exponent | index << PMC_LINK_EXP_BITS | node flag
        auto linkOf = [&](int id) -> uint32_t {
            if (id < 0) return PMC_LINK_NONE;
            const uint32_t e = uint32_t(maxLevel - g.node_level[id]);
            return g.node_first_child[id] < 0 ? (e | (uint32_t(T.perm[g.node_cell[id]]) << PMC_LINK_EXP_BITS))
                                              : (e | (uint32_t(internalIndex[id]) << PMC_LINK_EXP_BITS) | PMC_LINK_NODE);
        };
        // the link of a cell record through a wall: as linkOf, but an internal node whose children are all leaves with
        // consecutive device indices in child order becomes an octet link (the walk picks the child without a load)
        auto wallLinkOf = [&](int id) -> uint32_t {
            if (id < 0 || g.node_first_child[id] < 0) return linkOf(id);
            const int first = g.node_first_child[id];
            for (int l = 0; l < 8; ++l)
                if (g.node_first_child[first + l] >= 0) return linkOf(id);
            const int base = T.perm[g.node_cell[first]];
            for (int l = 1; l < 8; ++l)
                if (T.perm[g.node_cell[first + l]] != base + l) return linkOf(id);
            return uint32_t(maxLevel - g.node_level[id]) | (uint32_t(base) << PMC_LINK_EXP_BITS) | PMC_LINK_OCTET;
        };
        T.rootLink = linkOf(0);
        // top-down search table (pmc_walk.inc topDown): per cell of the regular grid of level Lc the node of level Lc
        // that covers it, or the coarser leaf
        {
            const int lc = std::min(maxLevel, 6);
            T.coarseLevel = lc;
            const int nc = 1 << lc;
            T.coarse.resize(size_t(nc) * nc * nc);
            for (int cz = 0; cz < nc; ++cz)
                for (int cy = 0; cy < nc; ++cy)
                    for (int cx = 0; cx < nc; ++cx)
                    {
                        const int px = cx << (maxLevel - lc), py = cy << (maxLevel - lc), pz = cz << (maxLevel - lc);
                        int node = 0;
                        while (g.node_first_child[node] >= 0 && g.node_level[node] < lc)
                        {
                            const int half = 1 << (maxLevel - g.node_level[node] - 1);
                            const int l = ((px - fx[node]) >= half ? 1 : 0) + ((py - fy[node]) >= half ? 2 : 0)
                                          + ((pz - fz[node]) >= half ? 4 : 0);
                            node = g.node_first_child[node] + l;
                        }
                        T.coarse[(size_t(cz) * nc + cy) * nc + cx] = linkOf(node);
                    }
        }

        // the node at level <= level(id) that covers the region just across `wall` of node id (-1: outside the grid)
        auto covering = [&](int id, int wall) -> int {
            int axis = wall >> 1, side = wall & 1;
            int size = 1 << (maxLevel - g.node_level[id]);
            int px = fx[id], py = fy[id], pz = fz[id];  // a fine cell index inside the neighbour region
            int* pa = axis == 0 ? &px : axis == 1 ? &py : &pz;
            *pa += side ? size : -1;
            int full = 1 << maxLevel;
            if (*pa < 0 || *pa >= full) return -1;
            int node = 0;
            while (g.node_first_child[node] >= 0 && g.node_level[node] < g.node_level[id])
            {
                int half = 1 << (maxLevel - g.node_level[node] - 1);
                int l = ((px - fx[node]) >= half ? 1 : 0) + ((py - fy[node]) >= half ? 2 : 0) + ((pz - fz[node]) >= half ? 4 : 0);
                node = g.node_first_child[node] + l;
            }
            return node;
        };

        const int numCells = g.num_cells;
        const int cellSlots = T.cellSlots;
        T.leaves.assign(cellSlots, LeafRec{});
        T.cells.assign(size_t(cellSlots), CellRec{});
        T.internals.assign(numInternal, NodeRec{});
        T.nbrStart.assign(6 * size_t(cellSlots) + 1, 0);
        T.nbrList.clear();
        std::vector<int32_t> nodeOfCell(numCells, -1);
        for (int id = 0; id < numNodes; ++id)
        {
            int m = g.node_cell[id];
            if (g.node_first_child[id] < 0)
            {
                if (m < 0 || m >= numCells) return fail(PMC_ERR_INVALID, "octree leaf without a valid cell index");
                nodeOfCell[m] = id;
            }
            else
            {
                NodeRec& rec = T.internals[internalIndex[id]];
                rec.code = code(id);
                for (int l = 0; l < 8; ++l) rec.child[l] = linkOf(g.node_first_child[id] + l);
            }
        }
        for (int m = 0; m < numCells; ++m)
            if (nodeOfCell[m] < 0) return fail(PMC_ERR_INVALID, "cell without a leaf node");
        for (int dev = 0; dev < cellSlots; ++dev)
        {
            const int m = T.cellExt[dev];
            for (int wall = 0; wall < 6; ++wall) T.nbrStart[6 * size_t(dev) + wall] = (int32_t)T.nbrList.size();
            if (m < 0) continue;  // padding of the last group
            const int id = nodeOfCell[m];
            LeafRec& rec = T.leaves[dev];
            rec.code = code(id);
            rec.density = density[m];
            CellRec& hot = T.cells[dev];
            hot.density = density[m];
            for (int wall = 0; wall < 6; ++wall)
            {
                // the leaf across the wall (same size or coarser), or the same-size internal node (finer neighbours: the walk
                // picks the child by the index bits of its position), or "outside"
                hot.link[wall] = wallLinkOf(covering(id, wall));
                // the reference's neighbour list of this leaf, in device numbering
                T.nbrStart[6 * size_t(dev) + wall] = (int32_t)T.nbrList.size();
                for (int qq = g.nbr_start[6 * size_t(id) + wall]; qq < g.nbr_start[6 * size_t(id) + wall + 1]; ++qq)
                {
                    int nb = g.nbr_list[qq];
                    if (g.node_first_child[nb] >= 0)
                        return fail(PMC_ERR_INVALID, "neighbour list of a leaf contains a non-leaf node");
                    T.nbrList.push_back(T.perm[g.node_cell[nb]]);
                }
            }
        }
        T.nbrStart[6 * size_t(cellSlots)] = (int32_t)T.nbrList.size();
        return PMC_OK;
    }
}

int pmcUploadOctreeGrid(pmc_ctx* ctx, const pmc_scene* scene, const pmc_medium& med, std::vector<int32_t>& devToCell)
{
    DevScene& D = ctx->dev;
    const pmc_grid& g = scene->grid;
    int rc = PMC_OK;
        TreeBuild T;
        if ((rc = buildTree(g, med.number_density, T))) return rc;
        D.lmax = T.lmax;
        D.root_link = T.rootLink;
        if (size_t(T.cellSlots) > PMC_LINK_MAX_INDEX || T.internals.size() > PMC_LINK_MAX_INDEX)
            return (fail(PMC_ERR_UNSUPPORTED, "octree with 2^25 cells or nodes or more (25-bit link index)"));
        D.tab_stride_bytes = 8u * uint32_t(T.tabn);
        D.fine_scale[0] = double(1 << T.lmax) / (g.xmax - g.xmin);
        D.fine_scale[1] = double(1 << T.lmax) / (g.ymax - g.ymin);
        D.fine_scale[2] = double(1 << T.lmax) / (g.zmax - g.zmin);
        if ((rc = ctx->upload(T.table.data(), T.table.size(), &D.coord_tab))) return rc;
        if ((rc = ctx->upload(T.leaves.data(), T.leaves.size(), &D.leaves))) return rc;
        if ((rc = ctx->upload(T.cells.data(), T.cells.size(), &D.cell_tab))) return rc;
        if ((rc = ctx->upload(T.internals.data(), T.internals.size(), &D.nodes))) return rc;
        D.coarse_level = T.coarseLevel;
        if ((rc = ctx->upload(T.coarse.data(), T.coarse.size(), &D.coarse_tab))) return rc;
        if ((rc = ctx->upload(T.nbrStart.data(), T.nbrStart.size(), &D.nbr_start))) return rc;
        if ((rc = ctx->upload(T.nbrList.data(), T.nbrList.size(), &D.nbr_list))) return rc;
        if ((rc = ctx->upload(T.cellExt.data(), T.cellExt.size(), &D.cell_ext))) return rc;
        devToCell = T.cellExt;
        D.cell_slots = T.cellSlots;
        // (levels 13-20: 0.2 ... 25 MB: not in LDS; the walk reads the six walls of a step from global memory)
        D.tab_in_lds = T.lmax <= 12 ? 1 : 0;
        D.lds_grid_len = D.tab_in_lds ? 3 * T.tabn : 0;
    return rc;
}

int pmcUploadVoronoiGrid(pmc_ctx* ctx, const pmc_scene* scene, const pmc_medium& med)
{
    DevScene& D = ctx->dev;
    const pmc_grid& g = scene->grid;
    int rc = PMC_OK;
        if (g.num_cells < 1 || !g.site || !g.vnbr_start || !g.vnbr_list || g.vblock_n < 1 || !g.vblock_start || !g.vblock_list)
            return (fail(PMC_ERR_INVALID, "Voronoi grid tables are missing"));
        std::vector<double> rec(4 * size_t(g.num_cells));
        for (int m = 0; m < g.num_cells; ++m)
        {
            rec[4 * size_t(m)] = g.site[3 * size_t(m)], rec[4 * size_t(m) + 1] = g.site[3 * size_t(m) + 1];
            rec[4 * size_t(m) + 2] = g.site[3 * size_t(m) + 2], rec[4 * size_t(m) + 3] = med.number_density[m];
        }
        const size_t nb3 = size_t(g.vblock_n) * g.vblock_n * g.vblock_n;
        for (int m = 0; m < g.num_cells; ++m)
            for (int q = g.vnbr_start[m]; q < g.vnbr_start[m + 1]; ++q)
                if (g.vnbr_list[q] < -6 || g.vnbr_list[q] >= g.num_cells)
                    return (fail(PMC_ERR_INVALID, "Voronoi neighbour list holds an invalid index"));
        if ((rc = ctx->upload(rec.data(), rec.size(), &D.vsite))) return rc;
        if ((rc = ctx->upload(g.vnbr_start, size_t(g.num_cells) + 1, &D.vnbr_start))) return rc;
        if ((rc = ctx->upload(g.vnbr_list, size_t(g.vnbr_start[g.num_cells]), &D.vnbr_list))) return rc;
        {
            const size_t np = size_t(g.vnbr_start[g.num_cells]);
            std::vector<double> pair(4 * np, 0.);
            for (size_t q = 0; q < np; ++q)
            {
                const int mi = g.vnbr_list[q];
                if (mi >= 0)
                {
                    pair[4 * q] = g.site[3 * size_t(mi)], pair[4 * q + 1] = g.site[3 * size_t(mi) + 1];
                    pair[4 * q + 2] = g.site[3 * size_t(mi) + 2];
                }
                const long long bits = mi;
                std::memcpy(&pair[4 * q + 3], &bits, sizeof(double));
            }
            if ((rc = ctx->upload(pair.data(), pair.size(), &D.vpair))) return rc;
        }
        {
            // the header record of a cell (DevScene::vhead)
            std::vector<double> head(8 * size_t(g.num_cells), 0.);
            for (int m = 0; m < g.num_cells; ++m)
            {
                for (int a = 0; a < 3; ++a) head[8 * size_t(m) + a] = g.site[3 * size_t(m) + a];
                head[8 * size_t(m) + 3] = med.number_density[m];
                const int32_t bounds[2] = {g.vnbr_start[m], g.vnbr_start[m + 1]};
                std::memcpy(&head[8 * size_t(m) + 4], bounds, sizeof(double));
            }
            if ((rc = ctx->upload(head.data(), head.size(), &D.vhead))) return rc;
        }
        D.vcull = nullptr;
        for (int i = 0; i < 16; ++i) D.vobs_of_inst[i] = -1;
        if (!pmcTune("PMC_VORO_NO_CULL"))
        {
            // neighbours that no direction of a cone can leave the cell through (DevScene::vcull).  A cone = the directions with
            // one sign pattern and one order of |k_x|, |k_y|, |k_z|: the non-negative combinations of three extreme rays, so
            // n . k <= 0 on the cone <=> n . e <= 0 for the three rays; the margin (1e-9 |n| |e|) is far above the rounding of
            // the kernel's n . k.
            static const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
            // every cone is divided once more at the midpoints of its edges (PMC_VORO_CONES = 192: four sub-cones, the rays
            // e1, e1 + e2, e1 + e3 | e2, e1 + e2, e2 + e3 | e3, e1 + e3, e2 + e3 | e1 + e2, e1 + e3, e2 + e3)
            static const int sub[4][3][3] = {{{1, 0, 0}, {1, 1, 0}, {1, 0, 1}}, {{0, 1, 0}, {1, 1, 0}, {0, 1, 1}}, {{0, 0, 1}, {1, 0, 1}, {0, 1, 1}},
                                             {{1, 1, 0}, {1, 0, 1}, {0, 1, 1}}};
            // 32-bit masks, 768 bytes per cell: a cell of the 10^5- and 10^6-site grids of the BASELINE scene has 15.2 / 15.4
            // neighbours on average, 99 % of the cells at most 24, 35 at most (neighbours beyond the 32nd are always read)
            const int ncell = g.num_cells;
            std::vector<uint32_t> cull(size_t(ncell) * PMC_VORO_CONES, 0u);
            auto cullCells = [&](int mFirst, int mLast) {
            for (int m = mFirst; m < mLast; ++m)
                for (int q = g.vnbr_start[m]; q < g.vnbr_start[m + 1] && q - g.vnbr_start[m] < 32; ++q)
                {
                    const int j = q - g.vnbr_start[m];
                    const int mi = g.vnbr_list[q];
                    double nv[3] = {0., 0., 0.};
                    double norm = 0.;
                    if (mi >= 0)
                    {
                        for (int a = 0; a < 3; ++a) nv[a] = g.site[3 * size_t(mi) + a] - g.site[3 * size_t(m) + a];
                        norm = std::sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
                    }
                    for (int sgn = 0; sgn < 8; ++sgn)
                        for (int p = 0; p < 6; ++p)
                        {
                            // the extreme rays of the cone: e1 along the largest component, e2 = e1 + the second, e3 = e2 + the third
                            double e[3][3] = {{0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}};
                            for (int r = 0; r < 3; ++r)
                                for (int t = r; t < 3; ++t) e[t][perm[p][r]] = ((sgn >> perm[p][r]) & 1) ? -1. : 1.;
                            for (int c = 0; c < PMC_VORO_CONES / 48; ++c)
                            {
                                bool skip;
                                if (mi >= 0)
                                {
                                    skip = norm > 0.;
                                    for (int r = 0; r < 3 && skip; ++r)
                                    {
                                        double ray[3];
                                        for (int a = 0; a < 3; ++a)
                                            ray[a] = PMC_VORO_CONES == 48 ? e[r][a] : sub[c][r][0] * e[0][a] + sub[c][r][1] * e[1][a] + sub[c][r][2] * e[2][a];
                                        const double dot = nv[0] * ray[0] + nv[1] * ray[1] + nv[2] * ray[2];
                                        if (!(dot <= -1e-9 * norm * 4.)) skip = false;
                                    }
                                }
                                else
                                {
                                    // walls -1 .. -6: x min, x max, y min, y max, z min, z max (reached only by k_a < 0 / k_a > 0)
                                    const int wall = -mi - 1;
                                    if (wall > 5) continue;
                                    const bool negative = ((sgn >> (wall >> 1)) & 1) != 0;
                                    skip = (wall & 1) ? negative : !negative;
                                }
                                // (cone-major: the masks of one cone are consecutive -- the walks towards an observer all use one cone)
                                if (skip) cull[size_t((sgn * 6 + p) * (PMC_VORO_CONES / 48) + c) * size_t(ncell) + size_t(m)] |= 1u << j;
                            }
                        }
                }
            };
            {
                // (cells are independent: all host cores)
                const int workers = std::max(1, std::min<int>(64, (int)std::thread::hardware_concurrency()));
                std::vector<std::thread> pool;
                for (int t = 0; t < workers; ++t)
                    pool.emplace_back(cullCells, int(int64_t(ncell) * t / workers), int(int64_t(ncell) * (t + 1) / workers));
                for (auto& t : pool) t.join();
            }
            if ((rc = ctx->upload(cull.data(), cull.size(), &D.vcull))) return rc;
            // ---- a table of RUNS (DevScene::vobs_run, vgen_run): per cell ONE run of 64-byte units -- header {site, density, number of entries}, then its
            // entries (first[m] .. first[m + 1] of `entries`: {site x, y, z, neighbour index}) in groups of PMC_VORO_RUN_LANES, a group as {x, y} of each
            // entry followed by {z, tag} of each: the lanes that share a walk read a group with two coalesced loads.  An entry's tag carries the unit
            // at which its neighbour's run starts next to the neighbour's index
            const auto uploadRuns = [&](const std::vector<double>& entries, const std::vector<int32_t>& first, const double** runsOut, const uint32_t** startOut) -> int {
                constexpr size_t LANES = PMC_VORO_RUN_LANES, GROUP_UNITS = LANES / 2;
                // (cells with more entries than a link can name -- 30 -- say so in their header; PMC_VORO_LINK_COUNT_MAX lowers the limit: a test of that path)
                uint32_t linkCountMax = PMC_VORO_RUN_COUNT_UNKNOWN - 1u;
                if (const char* v = pmcTune("PMC_VORO_LINK_COUNT_MAX")) linkCountMax = std::min<uint32_t>(linkCountMax, (uint32_t)std::max(0, atoi(v)));
                std::vector<uint32_t> start(size_t(ncell) + 1);
                size_t units = 0;
                for (int m = 0; m < ncell; ++m)
                {
                    // (the link to the run: its first unit and, up to 30, the number of its entries -- pmc_device.h PMC_VORO_RUN_UNIT_BITS)
                    const uint32_t entriesOf = uint32_t(first[m + 1] - first[m]);
                    start[m] = uint32_t(units & PMC_VORO_RUN_UNIT_MASK) | ((entriesOf > linkCountMax ? PMC_VORO_RUN_COUNT_UNKNOWN : entriesOf) << PMC_VORO_RUN_UNIT_BITS);
                    units += 1 + GROUP_UNITS * ((size_t(entriesOf) + LANES - 1) / LANES);
                }
                if (units + PMC_VORO_RUN_PAD >= (size_t(1) << PMC_VORO_RUN_UNIT_BITS))
                    return fail(PMC_ERR_UNSUPPORTED, "Voronoi table of runs beyond 2^27 units of 64 bytes");
                std::vector<double> orun(8 * (units + PMC_VORO_RUN_PAD), 0.);  // (padding: a walk may request a group that the run does not have)
                const unsigned long long noEntry = (unsigned long long)(uint32_t)(-7);
                for (int m = 0; m < ncell; ++m)
                {
                    double* head = &orun[8 * size_t(start[m] & PMC_VORO_RUN_UNIT_MASK)];
                    for (int a = 0; a < 3; ++a) head[a] = g.site[3 * size_t(m) + a];
                    head[3] = med.number_density[m];
                    const int32_t count[2] = {first[m + 1] - first[m], 0};
                    std::memcpy(&head[4], count, sizeof(double));
                    const size_t groups = (size_t(count[0]) + LANES - 1) / LANES;
                    for (size_t e = 0; e < groups * LANES; ++e)
                    {
                        double* group = head + 8 + 4 * LANES * (e / LANES);
                        double* xy = group + 2 * (e % LANES);
                        double* zt = group + 2 * LANES + 2 * (e % LANES);
                        unsigned long long tag = noEntry;
                        if (e < size_t(count[0]))
                        {
                            const double* src = &entries[4 * (size_t(first[m]) + e)];
                            xy[0] = src[0], xy[1] = src[1], zt[0] = src[2];
                            long long bits;
                            std::memcpy(&bits, &src[3], sizeof(double));
                            const int mi = int(bits);
                            tag = (unsigned long long)(uint32_t)mi | (mi >= 0 ? (unsigned long long)start[mi] << 32 : 0ull);
                        }
                        std::memcpy(&zt[1], &tag, sizeof(double));
                    }
                }
                int rcu;
                if ((rcu = ctx->upload(orun.data(), orun.size(), runsOut))) return rcu;
                return ctx->upload(start.data(), size_t(ncell), startOut);
            };
            // ---- all neighbours of a cell as a run (DevScene::vgen_run): what a PROPAGATION walk in voroPropKernel reads -- no mask, one run of
            // memory (4.75 lines per visit instead of header + mask + scattered entries: 6.2); left out where device memory is short
            if (!pmcTune("PMC_VORO_NO_PROP_KERNEL"))
            {
                size_t freeBytes = 0, totalBytes = 0;
                const size_t need = 32 * size_t(g.vnbr_start[ncell]) + 96 * size_t(ncell);
                if (hipMemGetInfo(&freeBytes, &totalBytes) != hipSuccess || need <= freeBytes / 4)
                {
                    std::vector<double> all(4 * size_t(g.vnbr_start[ncell]), 0.);
                    std::vector<int32_t> firstAll(size_t(ncell) + 1);
                    for (int m = 0; m <= ncell; ++m) firstAll[m] = g.vnbr_start[m];
                    for (int q = 0; q < g.vnbr_start[ncell]; ++q)
                    {
                        const int mi = g.vnbr_list[q];
                        double* e = &all[4 * size_t(q)];
                        if (mi >= 0) e[0] = g.site[3 * size_t(mi)], e[1] = g.site[3 * size_t(mi) + 1], e[2] = g.site[3 * size_t(mi) + 2];
                        const long long bits = mi;
                        std::memcpy(&e[3], &bits, sizeof(double));
                    }
                    if ((rc = uploadRuns(all, firstAll, &D.vgen_run, &D.vgen_start))) return rc;
                    // ... and per main cone the entries its sub-cones' masks keep (a neighbour beyond the 32nd is always kept)
                    if (PMC_VORO_CONES == 192 && !pmcTune("PMC_VORO_NO_CONE_TABLES") && (hipMemGetInfo(&freeBytes, &totalBytes) != hipSuccess || 48 * need <= freeBytes / 4))
                    {
                        const int workers = std::max(1, std::min<int>(16, (int)std::thread::hardware_concurrency()));
                        for (int c0 = 0; c0 < 48; c0 += workers)
                        {
                            const int nc = std::min(workers, 48 - c0);
                            std::vector<std::vector<double>> kept(nc);
                            std::vector<std::vector<int32_t>> firstKept(nc);
                            std::vector<std::thread> pool;
                            for (int w = 0; w < nc; ++w)
                                pool.emplace_back([&, w]() {
                                    const int c = c0 + w;
                                    std::vector<double>& e = kept[w];
                                    std::vector<int32_t>& f = firstKept[w];
                                    f.resize(size_t(ncell) + 1);
                                    e.reserve(all.size() * 3 / 4);
                                    for (int m = 0; m < ncell; ++m)
                                    {
                                        f[m] = int32_t(e.size() / 4);
                                        uint32_t mask = 0xFFFFFFFFu;  // culled by every sub-cone
                                        for (int sub = 0; sub < 4; ++sub) mask &= cull[size_t(4 * c + sub) * size_t(ncell) + size_t(m)];
                                        for (int q = g.vnbr_start[m]; q < g.vnbr_start[m + 1]; ++q)
                                        {
                                            const int j = q - g.vnbr_start[m];
                                            if (j < 32 && ((mask >> j) & 1u)) continue;
                                            e.insert(e.end(), &all[4 * size_t(q)], &all[4 * size_t(q)] + 4);
                                        }
                                    }
                                    f[ncell] = int32_t(e.size() / 4);
                                });
                            for (auto& t : pool) t.join();
                            for (int w = 0; w < nc; ++w)
                                if ((rc = uploadRuns(kept[w], firstKept[w], &D.vcone_run[c0 + w], &D.vcone_start[c0 + w]))) return rc;
                        }
                    }
                }
            }
            // ---- per observer: the kept neighbour entries of its cone, packed (DevScene::vobs_*).  All peel-off walks towards an observer
            // have ONE direction, hence one cone and one mask per cell
            int observers = 0;
            for (int i = 0; i < scene->num_instruments; ++i) observers += scene->instruments[i].same_observer_as_preceding ? 0 : 1;
            if (observers <= 4 && scene->num_instruments <= 16 && !pmcTune("PMC_VORO_NO_OBSERVER_LISTS"))
            {
                int k = -1;
                for (int i = 0; i < scene->num_instruments; ++i)
                {
                    const pmc_instrument& ins = scene->instruments[i];
                    if (ins.same_observer_as_preceding)
                    {
                        D.vobs_of_inst[i] = (int8_t)k;
                        continue;
                    }
                    // (memory: at most 32 B per neighbour entry + 64 B per cell and observer; the tables are an acceleration only, so an
                    //  observer whose tables would take more than a quarter of the free device memory goes without, as do the later ones:
                    //  their walks use the cone masks above)
                    {
                        size_t freeBytes = 0, totalBytes = 0;
                        const size_t need = 32 * size_t(g.vnbr_start[ncell]) + 64 * size_t(ncell);
                        if (hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess && need > freeBytes / 4)
                        {
                            fprintf(stderr, "libpmc: per-observer Voronoi tables left out from observer %d on (%.1f GiB each, %.1f GiB free): peel-off walks use the cone masks\n",
                                    k + 1, double(need) / double(1 << 30), double(freeBytes) / double(1 << 30));
                            break;
                        }
                    }
                    ++k;
                    D.vobs_of_inst[i] = (int8_t)k;
                    // the cone of the observer's direction: as voroCone (pmc_walk.inc)
                    const double kx = ins.kobs[0], ky = ins.kobs[1], kz = ins.kobs[2];
                    const int sgn = (kx < 0. ? 1 : 0) | (ky < 0. ? 2 : 0) | (kz < 0. ? 4 : 0);
                    const double ax = std::fabs(kx), ay = std::fabs(ky), az = std::fabs(kz);
                    int pp;
                    if (ax >= ay)
                        pp = ay >= az ? 0 : ax >= az ? 1 : 4;
                    else
                        pp = ax >= az ? 2 : ay >= az ? 3 : 5;
                    int cone = sgn * 6 + pp;
                    if (PMC_VORO_CONES != 48)
                    {
                        const double x = std::fmax(ax, std::fmax(ay, az)), z = std::fmin(ax, std::fmin(ay, az)), y = (ax + ay + az) - x - z;
                        const double a = x - y, b = y - z, c = z;
                        const int subc = a >= b + c ? 0 : b >= a + c ? 1 : c >= a + b ? 2 : 3;
                        cone = cone * 4 + subc;
                    }
                    const bool exactCull = !pmcTune("PMC_VORO_CONE_CULL_ONLY");
                    // pass 1: the kept entries in list order, {site x, y, z, neighbour index} and where every cell's entries start
                    std::vector<double> opair;
                    std::vector<int32_t> first(size_t(ncell) + 1);
                    opair.reserve(4 * size_t(g.vnbr_start[ncell]) * 2 / 3);
                    for (int m = 0; m < ncell; ++m)
                    {
                        const uint32_t mask = cull[size_t(cone) * size_t(ncell) + size_t(m)];
                        first[m] = int32_t(opair.size() / 4);
                        for (int q = g.vnbr_start[m]; q < g.vnbr_start[m + 1]; ++q)
                        {
                            const int j = q - g.vnbr_start[m];
                            if (j < 32 && ((mask >> j) & 1u)) continue;
                            const int mi = g.vnbr_list[q];
                            double e[4] = {0., 0., 0., 0.};
                            if (mi >= 0)
                            {
                                e[0] = g.site[3 * size_t(mi)], e[1] = g.site[3 * size_t(mi) + 1], e[2] = g.site[3 * size_t(mi) + 2];
                                // ONE direction per observer: the candidate test of the walk itself (voroCandidate: n . k > 0, the same
                                // doubles in the same order, no contraction) decides here which sites can ever be the exit -- half of them
                                if (exactCull)
                                {
                                    const double nx = e[0] - g.site[3 * size_t(m)], ny = e[1] - g.site[3 * size_t(m) + 1], nz = e[2] - g.site[3 * size_t(m) + 2];
                                    const double ndotk = nx * kx + ny * ky + nz * kz;
                                    if (!(ndotk > 0)) continue;
                                }
                            }
                            const long long bits = mi;
                            std::memcpy(&e[3], &bits, sizeof(double));
                            opair.insert(opair.end(), e, e + 4);
                        }
                    }
                    first[ncell] = int32_t(opair.size() / 4);
                    if ((rc = uploadRuns(opair, first, &D.vobs_run[k], &D.vobs_start[k]))) return rc;
                }
            }
        }
        D.vblock_n = g.vblock_n;
        if ((rc = ctx->upload(g.vblock_start, nb3 + 1, &D.vblock_start))) return rc;
        if ((rc = ctx->upload(g.vblock_list, size_t(g.vblock_start[nb3]), &D.vblock_list))) return rc;
        D.lds_grid_len = 0;
        D.lmax = 0;
    return rc;
}
