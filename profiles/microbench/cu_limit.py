#!/usr/bin/env python3
"""What one CU's vector memory pipeline sustains for lane-divergent gathers, by the SHAPE of the loads of a lane-step (GPU box; needs
libbridge.so, see bridge.py).  Private trajectories; 32 workgroups = 32 CUs (one per XCD slot in turn), so that L2 and fabric are far
from their limits; then the whole chip (256 workgroups)."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = C.CDLL(os.path.join(ROOT, "profiles", "microbench", "libbridge.so"))
B.bridge_shape.restype = C.c_double
B.bridge_shape.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int]
B.bridge_random_table.restype = C.c_void_p
B.bridge_random_table.argtypes = [C.c_uint32]
table = B.bridge_random_table(1 << 21)  # 64 MB
shapes = ["one 16 B load", "two 16 B loads, one 32 B record", "8 B + 4 B of one line", "8 B + 16 B + 8 B of one line", "two 16 B loads, two lines", "one 8 B load",
          "two 16 B loads, scalar base + 32-bit offsets"]
for kb in (1024, 16384):
    lines = kb * 1024 // 128
    for grid in (32, 256):
        print(f"# table {kb} KB, {grid} workgroups: lane-steps/s per CU (1e8) at 2 / 4 / 8 / 12 / 16 waves per CU")
        for sh, name in enumerate(shapes):
            row = []
            for block in (128, 256, 512, 768, 1024):
                r = B.bridge_shape(sh, table, lines, grid, block, 1000)
                row.append(f"{r / grid / 1e8:6.2f}")
            print(f"{name:46s} " + " ".join(row), flush=True)
