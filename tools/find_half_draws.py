#!/usr/bin/env python3
"""Seeds for which a camera draw of (pixel 0, sample 0) is exactly 0.5 - rand01() = 0.5 makes r = 2 * rand01() exactly 1.0,
the point where render_pixel's tent filter changes branch (mod.rs:820-830).  A draw hits one given 24-bit value once in
2^24 tries: vectorised Philox4x32-7 over seeds (key = (seed, 0), counter = (0, 0, 0, 0)), in numpy.
    python tools/find_half_draws.py            ->  SEED_R1_IS_ONE / SEED_R2_IS_ONE of tests/kats_camera.py
"""
import numpy as np


def philox7_words01(seeds):
    m = np.uint64(0xFFFFFFFF)
    c0 = np.zeros_like(seeds)
    c1 = np.zeros_like(seeds)
    c2 = np.zeros_like(seeds)
    c3 = np.zeros_like(seeds)
    k0 = seeds & m
    k1 = seeds >> np.uint64(32)
    for _ in range(7):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = (p1 >> np.uint64(32)) ^ c1 ^ k0, p1 & m, (p0 >> np.uint64(32)) ^ c3 ^ k1, p0 & m
        k0 = (k0 + np.uint64(0x9E3779B9)) & m
        k1 = (k1 + np.uint64(0xBB67AE85)) & m
    return c0, c1


def main():
    found = [None, None]
    step = 1 << 22
    base = 0
    while None in found and base < (1 << 30):
        s = np.arange(base, base + step, dtype=np.uint64)
        w0, w1 = philox7_words01(s)
        for i, w in enumerate((w0, w1)):
            if found[i] is None:
                hit = np.nonzero((w >> np.uint64(8)) == np.uint64(0x800000))[0]
                if len(hit):
                    found[i] = int(s[hit[0]])
        base += step
    print("SEED_R1_IS_ONE = %d\nSEED_R2_IS_ONE = %d" % tuple(found))


if __name__ == "__main__":
    main()
