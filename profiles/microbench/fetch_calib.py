#!/usr/bin/env python3
"""Calibration of rocprofv3's FETCH_SIZE on a read pattern with a KNOWN number of lines from beyond L2 (GPU box; needs libbridge.so):
private-trajectory random gathers (bridge.hip chaseTrue: no two lanes ever share a trajectory) of 32-byte records (two 16-byte loads of one
64-byte sector) or 16-byte records from a 256 MB table -- 64x the L2 of an XCD, so all but ~1.6 % of the gathers miss L2 --, and a coalesced
streaming read of the same table for comparison (bridge_plain is not needed: torch is not used here; the stream is hipMemcpyDtoD).
Run under `rocprofv3 --pmc ...` by tools/fetch_calibration.sh, which divides the counters of the `chaseTrue` launches by the gather count."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = C.CDLL(os.path.join(ROOT, "profiles", "microbench", "libbridge.so"))
B.bridge_true_gather.restype = C.c_double
B.bridge_true_gather.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
B.bridge_random_table.restype = C.c_void_p
B.bridge_random_table.argtypes = [C.c_uint32]
table = B.bridge_random_table(1 << 23)  # 256 MB as 32-byte records
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
if os.environ.get("FETCH_CALIB_STREAMS"):
    # known byte counts of wide coalesced accesses and of scattered 8-byte stores (fetch_calib.hip)
    S = C.CDLL(os.path.join(ROOT, "profiles", "microbench", "libfetchcalib.so"))
    S.fetch_calib_run.argtypes = [C.c_size_t, C.POINTER(C.c_float)]
    ms = (C.c_float * 3)()
    nbytes = 1 << 31
    rc = S.fetch_calib_run(nbytes, ms)
    print(f"streams over {nbytes} bytes (rc {rc}): read {nbytes / ms[0] / 1e9:.2f} TB/s, write {nbytes / ms[1] / 1e9:.2f} TB/s, "
          f"scattered 8-byte stores ({nbytes // 64} sectors) {nbytes / 64 / ms[2] / 1e6:.2f}e9 /s")
    sys.exit(0)
for loads, records in ((2, 1 << 23), (1, 1 << 24), (2, 1 << 20)):
    # (every call launches chaseTrue twice: 10 warm-up steps, then `steps`)
    r = B.bridge_true_gather(loads, 1, table, records, 256, 768, steps, None)
    print(f"loads {loads} records {records} ({records * 16 * loads >> 20} MB): lanes {256 * 768} x steps {steps} (+10 warm-up) = {256 * 768 * (steps + 10)} gathers, {r / 1e11:.3f}e11 /s")
