#!/usr/bin/env python3
"""Driver of xcd_own.hip (GPU box): the synthetic propagation walker of bridge.hip on the headline scene's own cell table, with and without
spatial ownership of the table per XCD (walks handed over between XCDs through queues of 128-byte records).
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC profiles/microbench/xcd_own.hip -o profiles/microbench/libxcdown.so
  python profiles/microbench/xcd_own.py [--source uniform]"""
import argparse
import ctypes as C
import os
import re
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


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
    B = C.CDLL(os.path.join(ROOT, "profiles", "microbench", "libxcdown.so"))
    B.xcdown_run.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    out = (C.c_double * 32)()
    print(f"# xcd_own: scene {os.path.basename(ski)}, {t.cell_slots} cells, source {args.source}")

    def run(name, mode, block=768, refill=40, steps=args.steps, grid=256):
        rc = B.xcdown_run(mode, t.cell_table, int(t.cell_slots), None if os.environ.get("XCD_OWN_RANDOM_STARTS") else t.task_cell, min(int(t.num_slots), n), grid, block,
                          steps, refill, 4000, out)
        if rc:
            print(f"{name:60s} FAILED rc={rc}", flush=True)
            return
        ms, lane, wave, rounds, walks, handed, failed, left = (out[i] for i in range(8))
        per = [out[8 + i] for i in range(8)]
        wgs = [int(out[16 + i]) for i in range(8)]
        print(f"{name:60s} {grid:4d} x {block // 64:2d} waves  {lane / ms / 1e8:6.3f}e11 lane-steps/s  lanes {100 * lane / (64 * wave):5.1f} %  "
              f"steps/walk {lane / max(walks, 1):6.1f}  hand-overs/walk {handed / max(walks, 1):5.2f}  failed claims {failed:.0f}  left in queues {left:.0f}\n"
              f"{'':60s} lane-steps per XCD (share): {' '.join(f'{100 * p / max(lane, 1):4.1f}' for p in per)}   workgroups per XCD: {wgs}", flush=True)

    run("warm-up", 0)
    print("fresh starts per eighth of the table:", [int(out[24 + i]) for i in range(8)])
    for block, refill, grid in ((768, 40, 256), (768, 24, 256), (512, 24, 256), (1024, 24, 256), (1024, 40, 256), (512, 24, 512), (256, 24, 1024)):
        run(f"no ownership (bridge walker), rounds at {refill}", 0, block, refill, grid=grid)
        run(f"XCD owns an eighth of the table, FREE hand-overs (upper bound), rounds at {refill}", 2, block, refill, grid=grid)
    if os.environ.get("XCD_OWN_QUEUES"):
        run("XCD owns an eighth of the table, hand-over queues (one CAS per wave and round), rounds at 40", 1, 768, 40)


if __name__ == "__main__":
    main()
