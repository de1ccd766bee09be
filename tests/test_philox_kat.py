"""Known-answer test of the per-history random stream (include/pmc_philox.h).

The engine and the oracle share that header, so GPU-vs-oracle parity cannot see a wrong generator.  The vectors are the
Philox4x32-10 entries of Random123's known-answer file (Salmon et al., SC'11: `kat_vectors`, lines "philox4x32 10 ...").
The header is compiled with gcc into a three-line program here: no GPU needed."""
import os
import subprocess

import numpy as np

from conftest import ROOT

KAT = [
    # counter (4 words), key (2 words), expected output (4 words)
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]

PROGRAM = r"""
#include "pmc_philox.h"
#include <stdio.h>
#include <stdlib.h>
int main(int argc, char** argv)
{
    if (argc == 7)
    {
        uint32_t c[4], k[2];
        for (int i = 0; i < 4; ++i) c[i] = (uint32_t)strtoul(argv[1 + i], 0, 16);
        for (int i = 0; i < 2; ++i) k[i] = (uint32_t)strtoul(argv[5 + i], 0, 16);
        pmc_philox4x32_10(c, k[0], k[1]);
        printf("%08x %08x %08x %08x\n", c[0], c[1], c[2], c[3]);
    }
    else
    {
        /* the first deviates of history argv[2] under seed argv[1], as hexadecimal floats */
        pmc_rng g;
        pmc_rng_init(&g, strtoull(argv[1], 0, 10), strtoull(argv[2], 0, 10));
        for (int i = 0; i < 4; ++i) printf("%a\n", pmc_rng_uniform(&g));
    }
    return 0;
}
"""


def _program(tmp_path):
    src = tmp_path / "kat.c"
    exe = tmp_path / "kat"
    src.write_text(PROGRAM)
    subprocess.check_call(["gcc", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    return str(exe)


def test_philox4x32_10_known_answers(tmp_path):
    exe = _program(tmp_path)
    for ctr, key, want in KAT:
        out = subprocess.check_output([exe] + ["%x" % w for w in ctr + key]).decode().split()
        assert tuple(int(w, 16) for w in out) == want


def test_uniform_deviates_follow_from_the_block(tmp_path):
    """a block gives two deviates (x + 1/2) 2^-52 from its 52 leading bits (pmc_bits_to_unit); counter = (history, block, "PMC1")"""
    exe = _program(tmp_path)
    seed, history = 0x0123456789abcdef, 0xfedcba9876543210
    got = [float.fromhex(s) for s in subprocess.check_output([exe, str(seed), str(history)]).decode().split()]
    want = []
    for block in range(2):
        ctr = (history & 0xffffffff, history >> 32, block, 0x504d4331)
        key = (seed & 0xffffffff, seed >> 32)
        c = _philox_python(ctr, key)
        for hi, lo in ((c[0], c[1]), (c[2], c[3])):
            want.append(((((hi << 32) | lo) >> 12) + 0.5) * 2.0 ** -52)
    assert got == want
    assert all(0. < u < 1. for u in got)


def _philox_python(ctr, key):
    """Philox4x32-10 as published (round: two 32x32 -> 64 multiplications, key bumped by the Weyl constants)"""
    c = [int(x) for x in ctr]
    k0, k1 = key
    for _ in range(10):
        p0 = 0xD2511F53 * c[0]
        p1 = 0xCD9E8D57 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k0, p1 & 0xffffffff, (p0 >> 32) ^ c[3] ^ k1, p0 & 0xffffffff]
        k0 = (k0 + 0x9E3779B9) & 0xffffffff
        k1 = (k1 + 0xBB67AE85) & 0xffffffff
    return c


def test_python_restatement_agrees_with_the_vectors():
    for ctr, key, want in KAT:
        assert tuple(_philox_python(ctr, key)) == want
    assert np.uint32(0x504d4331).tobytes()[::-1] == b"PMC1"
