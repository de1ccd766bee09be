#!/usr/bin/env python3
"""Driver of bridge.hip (GPU box): sets the headline scene up, runs a short segment so that the task records hold real first cells,
hands the device addresses of the engine's hot cell table to the synthetic walkers and prints lane-steps/s per variant.
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC profiles/microbench/bridge.hip -o profiles/microbench/libbridge.so
  python profiles/microbench/bridge.py [--source uniform]"""
import argparse
import ctypes as C
import os
import re
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

F = {"REAL": 1, "BODY": 2, "LDS": 4, "ROUNDS": 8, "IO": 16, "TRIM": 32, "OCT": 64}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--source", default="sersic")
    ap.add_argument("--steps", type=int, default=4000)
    args = ap.parse_args()
    from skirt9_amd.engine import Engine
    from skirt9_amd.host import Simulation
    ski = os.path.join(ROOT, "tests", "ski", "cfg2.ski")
    if args.source == "uniform":
        text = open(ski).read()
        new = ('<UniformBoxGeometry minX="-10000 pc" maxX="10000 pc" minY="-10000 pc" maxY="10000 pc" minZ="-1000 pc" maxZ="1000 pc"/>')
        text = re.sub(r"<SersicGeometry[^>]*/>", new, text, count=1)
        ski = os.path.join(tempfile.mkdtemp(), "cfg2u.ski")
        open(ski, "w").write(text)
    n = 2000000
    sim = Simulation(ski, num_packets=n).setup()
    eng = Engine(sim.scene, 0)
    eng.run_primary(0, n, 1)
    eng.sync()
    t = eng.debug_tables()
    B = C.CDLL(os.path.join(ROOT, "profiles", "microbench", "libbridge.so"))
    B.bridge_run.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_uint32]
    B.bridge_random_table.restype = C.c_void_p
    B.bridge_random_table.argtypes = [C.c_uint32]
    rnd = B.bridge_random_table(1 << 20)  # 32 MB of 32-byte records
    out = (C.c_double * 8)()

    def run(name, flags, block=768, refill=40, steps=args.steps, length=80, grid=256):
        f = sum(F[k] for k in flags)
        real = "REAL" in flags
        rc = B.bridge_run(f, t.cell_table if real else rnd, int(t.cell_slots) if real else (1 << 20), t.task_cell, min(int(t.num_slots), n), grid, block, steps,
                          refill, 4000 if real else length, out, int(getattr(t, 'loose_base', 0)))
        if rc:
            print(f"{name:70s} FAILED rc={rc}", flush=True)
            return
        ms, lane, wave, rounds, walks = out[0], out[1], out[2], out[3], out[4]
        print(f"{name:70s} {grid:4d} x {block // 64:2d} waves  {lane / ms / 1e8:6.3f}e11 lane-steps/s  lanes {100 * lane / (64 * wave):5.1f} %  "
              f"{ms * 1e6 / (wave / (grid * block // 64)):7.0f} ns/wave-step  steps/walk {lane / max(walks, 1):6.1f}  rounds/wave-step {rounds / wave:.4f}", flush=True)

    print(f"# bridge: scene {os.path.basename(ski)}, {t.cell_slots} cells, source {args.source}")
    B.bridge_plain_chase.restype = C.c_double
    B.bridge_plain_chase.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int]
    for block in (768, 512, 256):
        print(f"plain chase of gather_knee.hip, random 32 MB table, {block // 64} waves/CU: {B.bridge_plain_chase(rnd, 1 << 20, 256, block, 2000) / 1e11:.3f}e11 records/s")
    B.bridge_chase_variant.restype = C.c_double
    B.bridge_chase_variant.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int]
    for v, what in ((0, "as gather_knee"), (1, "all eight words used"), (2, "64-bit vector address"), (3, "eight words + vector address"),
                    (5, "eight words + link picked by axis / sign"), (7, "eight words + vector address + picked link"),
                    (9, "eight words + lanes pausing (partial exec)"), (15, "all four")):
        print(f"chase variant {v:2d} ({what}), 12 waves/CU: {B.bridge_chase_variant(v, rnd, 1 << 20, 256, 768, 2000) / 1e11:.3f}e11 records/s", flush=True)
    pow2 = 1 << (int(t.cell_slots).bit_length() - 1)
    print(f"plain chase over the first {pow2} records of the ENGINE's cell table (links as random numbers), 12 waves/CU: "
          f"{B.bridge_plain_chase(t.cell_table, pow2, 256, 768, 2000) / 1e11:.3f}e11 records/s")
    if os.environ.get("BRIDGE_OCTET"):
        # the octet-line table of the generation-8 experiment (engine built with it): the same walker on it
        for block in (768, 512):
            run("octet lines: real links, no body", ["REAL", "OCT"], block)
            run("octet lines: real + body + LDS", ["REAL", "OCT", "BODY", "LDS"], block)
            run("octet lines: real, rounds(40) free, no body", ["REAL", "OCT", "ROUNDS"], block)
            run("octet lines: all ingredients", ["REAL", "OCT", "BODY", "LDS", "ROUNDS", "IO", "TRIM"], block)
            run("octet lines: all but the recorded segments", ["REAL", "OCT", "BODY", "LDS", "ROUNDS", "IO"], block)
        return
    if os.environ.get("BRIDGE_SCALING"):
        # does the rate belong to the CU or to the chip?  workgroups (one per CU) x waves per workgroup
        for flags, name in ((["REAL"], "real table, real links, no body"), (["REAL", "BODY", "LDS", "ROUNDS", "IO", "TRIM"], "real, all ingredients"), ([], "random chase")):
            for grid in (32, 64, 128, 256, 512):
                for block in (128, 256, 512, 768, 1024):
                    if grid == 512 and block > 512:
                        continue
                    run(name, flags, block, grid=grid)
        return
    for block in (768, 512):
        run("random chase, 32 MB table (gather_knee)", [], block)
        run("+ real table, real links (walk locality)", ["REAL"], block)
        run("random chase + f64 body", ["BODY"], block)
        run("random chase + f64 body + 6 LDS reads", ["BODY", "LDS"], block)
        run("real + f64 body", ["REAL", "BODY"], block)
        run("real + f64 body + 6 LDS reads", ["REAL", "BODY", "LDS"], block)
        run("real + body + LDS + rounds at 40 waiting lanes (free rounds)", ["REAL", "BODY", "LDS", "ROUNDS"], block)
        run("real + body + LDS + rounds at 24 waiting lanes (free rounds)", ["REAL", "BODY", "LDS", "ROUNDS"], block, refill=24)
        run("real + body + LDS + rounds at 8 waiting lanes (free rounds)", ["REAL", "BODY", "LDS", "ROUNDS"], block, refill=8)
        run("real + body + LDS + rounds(40) + task loads / result stores", ["REAL", "BODY", "LDS", "ROUNDS", "IO"], block)
        run("real + body + LDS + rounds(24) + task loads / result stores", ["REAL", "BODY", "LDS", "ROUNDS", "IO"], block, refill=24)
        run("real + body + LDS + rounds(40) + IO + five recorded segments", ["REAL", "BODY", "LDS", "ROUNDS", "IO", "TRIM"], block)
        run("random + body + LDS + rounds(40) + IO + five recorded segments", ["BODY", "LDS", "ROUNDS", "IO", "TRIM"], block)
        run("random chase + rounds(40) (free)", ["ROUNDS"], block)
        run("real, rounds(40) (free), no body", ["REAL", "ROUNDS"], block)


if __name__ == "__main__":
    main()
