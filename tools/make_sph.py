#!/usr/bin/env python3
"""Synthetic smoothed-particle dust disk for BASELINE configs[3] (SPH-imported medium): N equal-mass particles drawn
from a double-exponential disk (scale length hR, scale height hz), smoothing length = eta * (m / rho)^(1/3) from the
analytic density at the particle (clamped), written as a SKIRT column text file with the unit header TextInFile reads.

  tools/make_sph.py --n 1000000 --seed 1 out.txt          (config 4: 10^6 particles, about 70 MB of text)
  tools/make_sph.py --n 3000 --seed 1 tests/ski/cfg4small_sph.txt   (the committed reduced fixture)
Deterministic for a given (n, seed, numpy version): the tests regenerate the large file instead of committing it."""
import argparse

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--hr", type=float, default=4000.0, help="scale length (pc)")
    ap.add_argument("--hz", type=float, default=250.0, help="scale height (pc)")
    ap.add_argument("--mass", type=float, default=1.6e7, help="total dust mass (Msun): tau_z(0.55 micron) ~ 1 at kappa 3000 m2/kg")
    ap.add_argument("--eta", type=float, default=2.0)
    ap.add_argument("--rmax", type=float, default=18000.0, help="truncation radius (pc): particles stay inside the grid")
    ap.add_argument("--zmax", type=float, default=3000.0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    n = args.n
    R = np.empty(0)
    while R.size < n:
        r = rng.gamma(2.0, args.hr, size=n)
        R = np.concatenate([R, r[r < args.rmax]])
    R = R[:n]
    z = np.empty(0)
    while z.size < n:
        t = rng.laplace(0.0, args.hz, size=n)
        z = np.concatenate([z, t[np.abs(t) < args.zmax]])
    z = z[:n]
    phi = rng.uniform(0.0, 2 * np.pi, size=n)
    x, y = R * np.cos(phi), R * np.sin(phi)
    m = args.mass / n
    rho = args.mass / (4 * np.pi * args.hr ** 2 * args.hz) * np.exp(-R / args.hr - np.abs(z) / args.hz)  # Msun/pc3
    h = args.eta * (m / rho) ** (1.0 / 3.0)
    h = np.clip(h, 10.0, 1500.0)
    with open(args.out, "w") as fh:
        fh.write("# synthetic SPH dust disk (tools/make_sph.py --n %d --seed %d)\n" % (n, args.seed))
        fh.write("# Column 1: position x (pc)\n# Column 2: position y (pc)\n# Column 3: position z (pc)\n")
        fh.write("# Column 4: size h (pc)\n# Column 5: mass (Msun)\n")
        data = np.column_stack([x, y, z, h, np.full(n, m)])
        np.savetxt(fh, data, fmt="%.9g")


if __name__ == "__main__":
    main()
