#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory from the UNMODIFIED reference (oracle/_ref, built by
`make -f oracle/Makefile.ref`).  Runs only where /root/reference exists; the fixtures themselves are data
(FITS cubes, SED tables, ray segment lists, per-cell densities) and are committed.

  python tests/golden/make_golden.py

Fixtures:
  cfg1_*       config 1 (tests/ski/cfg1.ski): 10^6 packets, seed 0, one thread -> FITS/SED/statistics files
  cfg2small_*  reduced config 2 (tests/ski/cfg2small.ski): 2x10^4 packets -> FITS/SED/statistics files
  cfg2deep_*   the config-2 scene with a cusped dust geometry (Plummer, 60 pc; tests/ski/cfg2deep.ski): a 2318-cell octree that
               reaches level 12 (the 21-bit index variants of the octree walk kernels) -> files, rays, cells
  cfg3small_*  reduced config 3 (tests/ski/cfg3small.ski): panchromatic, four instruments, 2x10^4 packets -> files
  cfg3z_*      the reduced config-3 scene at model redshift 0.5 (FlatUniverseCosmology) with SIX instruments, three of them in
               the observer frame (distance 0: wavelength bins at lambda (1+z), cosmological distances) -> files
  cfg1nf_*     non-forced scattering variant of config 1 (tests/ski/cfg1nf.ski): 2x10^5 packets -> files
  cfg2nf_*     non-forced scattering on the octree of reduced config 2 (tests/ski/cfg2nf.ski) -> files
  cfg1sed_*    config 1 with two SEDInstruments next to the FullInstrument (tests/ski/cfg1sed.ski): 10^5 packets -> files
  cfg3sed_*    reduced config 3 with a FileSED source spectrum (tests/ski/cfg3sed.ski + cfg3sed_sed.txt) -> files
  cfg3norm_*   the same with a SpecificLuminosityNormalization (per unit of frequency): SED files only
  cfg1file_*   config 1 with MeanFileDustMix (tests/ski/cfg1file.ski + cfg1file_dust.txt): 10^5 packets -> files
  cfg1rf_*, cfg3rf_*   config 1 and reduced config 3 with storeRadiationField and a RadiationFieldProbe (PerCellForm):
               the probe file <name>_rf_J.dat (gzip) and the SED files
  cfg2ea_*, cfg1nfea_*, cfg1rfea_*   the photon cycle with explicitAbsorption="true" (MonteCarloSimulation.cpp:568-569, 729-733, 751-766;
               MediumSystem.cpp:905-975, 1075-1110): reduced config 2 (forced scattering), the non-forced variant of config 1, and config 1 with
               the radiation field stored (probe file, gzip)
  cfg1con_*, cfg1netzer_*, cfg1laser_*, cfg2agn_*   point sources with an axisymmetric angular distribution (AxAngularDistribution.cpp:27-41,
      Conical / Netzer / LaserAngularDistribution.cpp; emission peel-off bias PhotonPacket.cpp:78); cfg2agn: Netzer source in a dust torus (octree)
  cfg2mm_*, cfg2mmea_*, cfg1mmnf_*, cfg3mm_*, cfg1mmrf_*   several medium components with constant cross sections (MediumSystem.cpp:874-887,
               678-730, 796-817): reduced config 2 with a second dust component (Plummer sphere, other mix, mass normalisation), the same
               with explicit absorption, the non-forced variant of config 1 with THREE components, reduced config 3 (panchromatic) with two,
               and config 1 with three components and the radiation field stored
  cfg1nomed_*  a simulation WITHOUT a medium system (simulationMode OligoNoMedium; MonteCarloSimulation.cpp:557: emission peel-off only,
               FluxRecorder records the total flux only): the point source of config 1, 10^5 packets -> files
  cfg5small_*  reduced config 5 (tests/ski/cfg5small.ski): Voronoi grid with 1500 random sites, panchromatic, four
               instruments -> files, rays, cells (the host layer's tessellation is its own: the traversal is compared bit
               for bit, cell volumes to rounding, sampled densities and output files statistically)
  cfg5_rays*   config 5 at full size (tests/ski/cfg5.ski, 10^5 Voronoi sites from tools/make_sites.py): 208 fixed rays and
               the reference's (m, ds) sequences
  cfg2_rays*, cfg2_cells.json   BASELINE configs[1] at FULL size (tests/ski/cfg2.ski: the 953 688-cell octree of the headline
               benchmark): 312 fixed rays (random, mid-plane, axis-parallel, and rays aimed at cell corners and edges) with the
               reference's (m, ds) sequences (gzip), the cell count and SHA-256 digests of the reference's per-cell volumes and
               number densities (bit patterns, cell order)
  cfg2_i0_sed.dat, cfg2_i0_sedstats.dat, cfg2_full_rebinned.npz   the photon loop of the same full-size scene: 10^5 packets, seed 0, one
               thread; the two SED files as written, and every FITS frame (flux components, statistics w^0 .. w^4) summed over
               8 x 8 blocks of the 512^2 pixels in double precision (the ten 1 MB FITS files themselves are not committed)
  cfg5dd_cells.npz   the same grid with its sites drawn from the dust density (policy DustDensity): volumes, densities
  cfg4deepest_*   an 18-level octree around a cusp of nested smoothed particles (tests/ski/cfg4deepest.ski): 88 rays, half of them through the
                  deepest levels, and the cell table
  cfg5relax_*   the reduced config 5 with relaxSites="true" (tests/ski/cfg5relax.ski): 48 rays and the cell table
  cfg5peak_*, cfg5imp_*   Voronoi site policies CentralPeak and ImportedSites (tests/ski/cfg5peak.ski, cfg5imp.ski): 40 rays and the cell table
  cfg4small_*  reduced config 4 (tests/ski/cfg4small.ski): dust imported from 3000 smoothed particles
               (tests/ski/cfg4small_sph.txt, made by tools/make_sph.py), 2x10^4 packets -> files, rays, cells
  *_rays.txt / *_rays_ref.txt   fixed rays and the reference's (m, ds) sequences (C99 hex floats)
  *_cells.npz  per-cell volume and number density as the reference computed them (bit patterns), and the dust
               cross sections at 0.55 micron
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "release", "SKIRT", "main", "skirt_ref")


def rays(scale, n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        r = (rng.random(3) - 0.5) * 2 * scale * (1.3 if i % 4 == 0 else 0.9)
        k = rng.normal(size=3)
        k /= np.linalg.norm(k)
        out.append((r, k))
    s2 = 1 / np.sqrt(2.0)
    out += [(np.zeros(3), np.array([1.0, 0, 0])), (np.zeros(3), np.array([0, -1.0, 0])), (np.zeros(3), np.array([0, 0, 1.0])),
            (np.zeros(3), np.array([s2, s2, 0])), (np.array([scale * 0.25, scale * 0.125, 0.0]), np.array([0, 0, -1.0])),
            (np.array([-scale * 2, 0.0, 0.0]), np.array([1.0, 0, 0])), (np.array([-scale * 2, 1.0, 1.0]), np.array([-1.0, 0, 0])),
            (np.array([1e15, 2e15, -3e15]), np.array([0.3, 0.5, 0.81]) / np.linalg.norm([0.3, 0.5, 0.81]))]
    return out


def rays_towards(centre, scale, n, seed):
    """rays that pass a point at impact parameters from `scale` down to scale / 2^(n/2): through the deepest levels of a tree refined around it"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        k = rng.normal(size=3)
        k /= np.linalg.norm(k)
        b = (rng.random(3) - 0.5) * 2 * scale * 2.0 ** (-0.5 * i)
        out.append((np.array(centre) + b - k * 30 * scale, k))
    return out


def rays_config2():
    """fixed rays for the full-size octree of tests/ski/cfg2.ski (box +-20 kpc x +-20 kpc x +-4 kpc): random rays, rays
    nearly in the mid-plane (hundreds of cells of the finest levels), the special rays of rays(), and rays through the
    corners and edges of cells (ties between exit walls, steps that overshoot a second wall)"""
    pc = 3.08567758e16
    scale = 4000 * pc
    rng = np.random.default_rng(7)
    out = []
    for i in range(280):
        r = (rng.random(3) - 0.5) * 2 * scale * np.array([5.0, 5.0, 1.0]) * (1.2 if i % 5 == 0 else 0.95)
        k = rng.normal(size=3)
        if i % 3 == 0:
            k[2] *= 0.02
        k /= np.linalg.norm(k)
        out.append((r, k))
    out += rays(scale, 0, 1)
    # dyadic points of the box are node corners: (i/2^a) of the extent; rays through them along diagonals
    ext = np.array([20000 * pc, 20000 * pc, 4000 * pc])
    s2, s3 = 1 / np.sqrt(2.0), 1 / np.sqrt(3.0)
    for frac, k in (((0.0, 0.0, 0.0), (s3, s3, s3)), ((0.25, 0.125, 0.0), (s2, -s2, 0.0)), ((0.03125, -0.0625, 0.015625), (s3, -s3, s3)),
                    ((-0.5, 0.25, 0.0), (0.0, s2, s2)), ((0.0078125, 0.0078125, 0.0), (1.0, 0.0, 0.0)),
                    ((0.001953125, -0.00390625, 0.0009765625), (-s3, s3, s3)), ((0.0, 0.0, 0.001953125), (s2, s2, 0.0)),
                    ((0.5, 0.5, 0.5), (-s3, -s3, -s3)), ((-1.5, -1.5, -1.5), (s3, s3, s3)), ((0.0625, 0.0, 0.0), (0.0, 0.0, 1.0)),
                    ((0.0, 0.03125, 0.0), (0.6, 0.0, 0.8)), ((0.015625, 0.015625, 0.015625), (0.0, -0.6, 0.8)),
                    ((0.125, 0.125, 0.0), (-s2, -s2, 0.0)), ((0.0, 0.0, 0.0), (s2, 0.0, s2)), ((0.0009765625, 0.0, 0.0), (0.0, 1.0, 0.0)),
                    ((-0.25, -0.25, -0.25), (s3, s3, s3)), ((0.0, 0.0, 0.0), (-s3, s3, -s3)), ((0.75, -0.75, 0.0), (-s2, s2, 0.0)),
                    ((0.00390625, 0.00390625, 0.00390625), (s3, s3, -s3)), ((0.0, -0.001953125, 0.0), (s2, s2, 0.0)),
                    ((0.25, 0.0, 0.0625), (-0.8, 0.0, -0.6)), ((0.0, 0.0, -0.03125), (0.0, s2, s2)),
                    ((0.0625, 0.0625, 0.0), (s2, s2, 0.0)), ((0.0, 0.0, 0.0), (0.0, s2, -s2))):
        out.append((np.array(frac) * ext, np.array(k) / np.linalg.norm(k)))
    return out


def read_fits(path):
    """primary image of a FITS file written by FITSInOut::write: float32, big endian"""
    raw = open(path, "rb").read()
    cards = {}
    pos = 0
    while True:
        card = raw[pos:pos + 80].decode("ascii")
        pos += 80
        if card.startswith("END"):
            break
        if "=" in card[:10]:
            cards[card[:8].strip()] = card[10:].split("/")[0].strip()
    pos = (pos + 2879) // 2880 * 2880
    shape = [int(cards[f"NAXIS{i}"]) for i in range(int(cards["NAXIS"]), 0, -1)]
    count = int(np.prod(shape))
    return np.frombuffer(raw[pos:pos + 4 * count], dtype=">f4").astype(np.float64).reshape(shape)


def rebin(a, f=8):
    a = a.reshape(a.shape[-2], a.shape[-1])
    ny, nx = a.shape
    return a.reshape(ny // f, f, nx // f, f).sum(axis=(1, 3))


def main():
    if not os.path.exists(REF):
        sys.exit("build the reference first: make -f oracle/Makefile.ref -j8")
    for name, scale in (("cfg1", 3.08567758e16), ("cfg1mesh", 3.08567758e16), ("cfg1mesh2", 3.08567758e16), ("cfg2small", 4000 * 3.08567758e16), ("cfg2deep", 300 * 3.08567758e16), ("cfg2deeper", 100 * 3.08567758e16), ("cfg3small", None), ("cfg3z", None), ("cfg1nf", None), ("cfg2nf", None),
                        ("cfg4small", 4000 * 3.08567758e16), ("cfg1file", None), ("cfg1sed", None), ("cfg3sed", None), ("cfg3norm", "sed"), ("cfg3disk", "sed"), ("cfg3multi", "sed"), ("cfg3ten", "sed"), ("cfg3flat", "sed"), ("cfg3off", "sed"), ("cfg3plum", "sed"), ("cfg1rf", "rf"), ("cfg3rf", "rf"), ("cfg2ea", None), ("cfg1nfea", None), ("cfg1rfea", "rf"), ("cfg2mm", None), ("cfg2mmea", None), ("cfg1mmnf", None), ("cfg3mm", None), ("cfg1mmrf", "rf"), ("cfg1con", None), ("cfg1netzer", None), ("cfg1laser", None), ("cfg2agn", None), ("cfg1nomed", None),
                        ("cfg5small", 4000 * 3.08567758e16), ("cfg5dd", "cells"), ("cfg5peak", "cellrays"), ("cfg5imp", "cellrays"), ("cfg5relax", "cellrays"), ("cfg4deepest", "cellrays"), ("cfg2shell", "cells"), ("cfg2torus", "cells"), ("cfg2ring", "cells"), ("cfg2gauss", "cells"), ("cfg5ddgauss", "cells"), ("cfg1list", "cells")):
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        ski = os.path.join(ROOT, "tests", "ski", name + ".ski")
        with tempfile.TemporaryDirectory() as tmp:
            # input files named in the ski file are read from the working directory (FilePaths::input)
            for f in os.listdir(os.path.join(ROOT, "tests", "ski")):
                if f.endswith(".txt"):
                    shutil.copy(os.path.join(ROOT, "tests", "ski", f), tmp)
            subprocess.check_call([REF, "run", ski, "-t", "1", "-o", tmp], cwd=tmp, stdout=subprocess.DEVNULL)
            if scale == "cells":
                # only the per-cell table (site order through the cell centres, volumes, sampled densities)
                cells = os.path.join(tmp, "cells.txt")
                subprocess.check_call([REF, "cells", ski, cells, "-w", "0.55e-6", "-o", tmp], cwd=tmp, stdout=subprocess.DEVNULL)
                vol, dens = [], []
                for line in open(cells):
                    t = line.split()
                    if t[0] in ("cells", "mix"):
                        continue
                    vol.append(float.fromhex(t[4]))
                    dens.append(float.fromhex(t[5]))
                np.savez_compressed(os.path.join(HERE, name + "_cells.npz"), volume=np.array(vol), density=np.array(dens))
                continue
            if scale == "sed":
                for f in sorted(os.listdir(tmp)):
                    if f.endswith("_sed.dat"):
                        shutil.copy(os.path.join(tmp, f), os.path.join(HERE, f))
                continue
            if scale == "rf":
                # radiation field variants of scenes above: keep the probe file (gzip) and the SEDs only
                import gzip
                for f in sorted(os.listdir(tmp)):
                    if f.endswith("_J.dat"):
                        with open(os.path.join(tmp, f), "rb") as src, gzip.GzipFile(os.path.join(HERE, f + ".gz"), "wb", mtime=0) as dst:
                            dst.write(src.read())
                    elif f.endswith("_sed.dat"):
                        shutil.copy(os.path.join(tmp, f), os.path.join(HERE, f))
                continue
            if scale == "cellrays":
                scale = 4000 * 3.08567758e16  # (further Voronoi site policies: the ray dump and the cell table only, no output files)
            else:
                for f in sorted(os.listdir(tmp)):
                    if f.endswith(".fits") or f.endswith(".dat"):
                        shutil.copy(os.path.join(tmp, f), os.path.join(HERE, f))
            if scale is None:
                continue  # grids of the same kinds as above: no separate ray / cell fixtures
            # rays: directions are written as hex floats so that both sides parse identical doubles
            rayfile = os.path.join(HERE, name + "_rays.txt")
            with open(rayfile, "w") as fh:
                some = rays(scale, 40, 1)
                if name == "cfg4deepest":
                    # (an 18-level octree around one point: rays that pass it at 100 pc ... 1e-4 pc)
                    some += rays_towards([1234.5 * 3.08567758e16, -567.25 * 3.08567758e16, 89.125 * 3.08567758e16], 100 * 3.08567758e16, 40, 3)
                for r, k in some:
                    fh.write(" ".join(float(v).hex() for v in list(r) + list(k)) + "\n")
            subprocess.check_call([REF, "rays", ski, rayfile, os.path.join(HERE, name + "_rays_ref.txt"), "-o", tmp], cwd=tmp,
                                  stdout=subprocess.DEVNULL)
            cells = os.path.join(tmp, "cells.txt")
            subprocess.check_call([REF, "cells", ski, cells, "-w", "0.55e-6", "-o", tmp], cwd=tmp, stdout=subprocess.DEVNULL)
            vol, dens, mix = [], [], None
            for line in open(cells):
                t = line.split()
                if t[0] == "cells":
                    continue
                if t[0] == "mix":
                    mix = [float.fromhex(v) for v in t[1:]]
                    continue
                vol.append(float.fromhex(t[4]))
                dens.append(float.fromhex(t[5]))
            np.savez_compressed(os.path.join(HERE, name + "_cells.npz"), volume=np.array(vol), density=np.array(dens),
                                mix=np.array(mix))
    # config 5 at full size: only the traversal is pinned (10^5 Voronoi cells; the site file is regenerated, not committed)
    if len(sys.argv) == 1 or "cfg5" in sys.argv[1:]:
        with tempfile.TemporaryDirectory() as tmp:
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_sites.py"), "--n", "100000", "--seed", "1",
                                   os.path.join(tmp, "cfg5_sites.txt")])
            rayfile = os.path.join(HERE, "cfg5_rays.txt")
            with open(rayfile, "w") as fh:
                for r, k in rays(4000 * 3.08567758e16, 200, 5):
                    fh.write(" ".join(float(v).hex() for v in list(r) + list(k)) + "\n")
            subprocess.check_call([REF, "rays", os.path.join(ROOT, "tests", "ski", "cfg5.ski"), rayfile,
                                   os.path.join(HERE, "cfg5_rays_ref.txt"), "-o", tmp], cwd=tmp, stdout=subprocess.DEVNULL)
    # config 2 at full size: the traversal and the cell table of the benchmark's own octree
    if len(sys.argv) == 1 or "cfg2" in sys.argv[1:]:
        import gzip
        import hashlib
        import json
        ski = os.path.join(ROOT, "tests", "ski", "cfg2.ski")
        with tempfile.TemporaryDirectory() as tmp:
            # the photon loop at full size (27 s): SED files + the FITS frames rebinned 8 x 8
            subprocess.check_call([REF, "run", ski, "-t", "1", "-o", tmp], cwd=tmp, stdout=subprocess.DEVNULL)
            for f in ("cfg2_i0_sed.dat", "cfg2_i0_sedstats.dat"):
                shutil.copy(os.path.join(tmp, f), os.path.join(HERE, f))
            frames = {}
            for name in ("total", "transparent", "primarydirect", "primaryscattered", "stats0", "stats1", "stats2", "stats3", "stats4"):
                frames[name] = rebin(read_fits(os.path.join(tmp, f"cfg2_i0_{name}.fits")))
            np.savez_compressed(os.path.join(HERE, "cfg2_full_rebinned.npz"), **frames)
            rayfile = os.path.join(HERE, "cfg2_rays.txt")
            with open(rayfile, "w") as fh:
                for r, k in rays_config2():
                    fh.write(" ".join(float(v).hex() for v in list(r) + list(k)) + "\n")
            ref = os.path.join(tmp, "cfg2_rays_ref.txt")
            subprocess.check_call([REF, "rays", ski, rayfile, ref, "-o", tmp], cwd=tmp, stdout=subprocess.DEVNULL)
            with open(ref, "rb") as src, gzip.GzipFile(os.path.join(HERE, "cfg2_rays_ref.txt.gz"), "wb", mtime=0) as dst:
                dst.write(src.read())
            cells = os.path.join(tmp, "cells.txt")
            subprocess.check_call([REF, "cells", ski, cells, "-w", "0.55e-6", "-o", tmp], cwd=tmp, stdout=subprocess.DEVNULL)
            vol, dens = [], []
            for line in open(cells):
                t = line.split()
                if t[0] in ("cells", "mix"):
                    continue
                vol.append(float.fromhex(t[4]))
                dens.append(float.fromhex(t[5]))
            vol, dens = np.array(vol), np.array(dens)
            json.dump({"num_cells": int(len(dens)), "volume_sha256": hashlib.sha256(vol.tobytes()).hexdigest(),
                       "density_sha256": hashlib.sha256(dens.tobytes()).hexdigest(),
                       "density_sum": float(dens.sum()).hex(), "volume_sum": float(vol.sum()).hex()},
                      open(os.path.join(HERE, "cfg2_cells.json"), "w"), indent=1)
    print("golden fixtures regenerated")


if __name__ == "__main__":
    main()
