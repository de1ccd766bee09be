#!/usr/bin/env python3
"""Registers and scratch of every kernel in a `hipcc -S --cuda-device-only` listing (tuning aid).  usage: kernel_regs.py file.s [substring...]"""
import re, sys
t = open(sys.argv[1]).read()
keys = sys.argv[2:]
for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)', t):
    n = m.group(1)
    if not keys or any(k in n for k in keys):
        print(f"{n[:72]:72s} scratch {m.group(2):>4s} sgpr {m.group(3):>3s} vgpr {m.group(4):>3s}")
