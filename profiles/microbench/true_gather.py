#!/usr/bin/env python3
"""The random-gather ceiling of the chip WITHOUT merging lanes (GPU box; needs libbridge.so, see bridge.py).
gather_knee.hip / gather_modes.hip let every lane follow the SAME random mapping idx -> f(idx) + c_step: two lanes that meet stay
together for ever, after 2000 steps a few thousand distinct trajectories are left among 2e5 lanes, and most "random" gathers are
served as duplicates.  Here a per-lane salt enters the next index; `distinct` = different final indices among the lanes."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = C.CDLL(os.path.join(ROOT, "profiles", "microbench", "libbridge.so"))
B.bridge_true_gather.restype = C.c_double
B.bridge_true_gather.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
B.bridge_random_table.restype = C.c_void_p
B.bridge_random_table.argtypes = [C.c_uint32]
table = B.bridge_random_table(1 << 23)  # 256 MB as 32-byte records
d = C.c_double(0)
for steps in (100, 500, 2000):
    r = B.bridge_true_gather(2, 0, table, 1 << 20, 256, 768, steps, C.byref(d))
    print(f"shared mapping (gather_knee), 32 MB, 12 waves/CU, {steps:5d} steps: {r / 1e11:.3f}e11 records/s, {int(d.value):7d} distinct of {256 * 768} lanes")
for steps in (500, 2000):
    r = B.bridge_true_gather(2, 1, table, 1 << 20, 256, 768, steps, C.byref(d))
    print(f"private trajectories,         32 MB, 12 waves/CU, {steps:5d} steps: {r / 1e11:.3f}e11 records/s, {int(d.value):7d} distinct of {256 * 768} lanes")
print("# private trajectories: records/s by table size, record shape and waves per CU (one workgroup per CU)")
for loads, rec in ((2, 32), (1, 16)):
    for kb in (512, 2048, 8192, 16384, 32768, 65536, 262144):
        records = kb * 1024 // rec
        if records > (1 << 23) * 32 // rec:
            continue
        row = []
        for block in (256, 512, 768, 1024):
            r = B.bridge_true_gather(loads, 1, table, records, 256, block, 1000, None)
            row.append(f"{block // 64:2d} waves {r / 1e11:.3f}e11")
        print(f"{rec:2d}-byte records ({loads} load{'s' if loads > 1 else ' '}), table {kb:6d} KB: " + "   ".join(row), flush=True)
