#!/usr/bin/env python3
"""Synthetic Voronoi sites for BASELINE configs[4] (VoronoiMeshSpatialGrid, policy File): 80 % of the sites follow the
double-exponential dust disk of the config-2 scene (scale length 4000 pc, scale height 250 pc), 20 % are uniform in
the domain box (+-20 x +-20 x +-4 kpc), so that the cells are small where the dust is and the whole box is covered.

  tools/make_sites.py --n 100000 --seed 1 out.txt
Deterministic for a given (n, seed, numpy version); the tests regenerate the file instead of committing it."""
import argparse

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    n_disk = int(0.8 * args.n)
    pts = []
    while sum(len(p) for p in pts) < n_disk:
        R = rng.gamma(2.0, 4000.0, size=n_disk)
        z = rng.laplace(0.0, 250.0, size=n_disk)
        phi = rng.uniform(0.0, 2 * np.pi, size=n_disk)
        x, y = R * np.cos(phi), R * np.sin(phi)
        ok = (np.abs(x) < 19990) & (np.abs(y) < 19990) & (np.abs(z) < 3990)
        pts.append(np.column_stack([x, y, z])[ok])
    disk = np.concatenate(pts)[:n_disk]
    n_uni = args.n - n_disk
    uni = np.column_stack([rng.uniform(-19990, 19990, n_uni), rng.uniform(-19990, 19990, n_uni), rng.uniform(-3990, 3990, n_uni)])
    sites = np.concatenate([disk, uni])
    with open(args.out, "w") as fh:
        fh.write("# synthetic Voronoi sites (tools/make_sites.py --n %d --seed %d)\n" % (args.n, args.seed))
        fh.write("# Column 1: position x (pc)\n# Column 2: position y (pc)\n# Column 3: position z (pc)\n")
        np.savetxt(fh, sites, fmt="%.9g")


if __name__ == "__main__":
    main()
