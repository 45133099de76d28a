"""Frame time of several builds of the library on one GPU: python tools/libs_probe.py <scene> <spp> lib.so ..."""
import os, subprocess, sys
scene, spp, libs = sys.argv[1], sys.argv[2], sys.argv[3:]
code = ("import sys; sys.path.insert(0, 'tools'); import ab_flags as f; import numpy as np\n"
        "img, n, t = f.render(sys.argv[2], int(sys.argv[3]), 0, 0, reps=2)\n"
        "print('%s %s: %.1f ms %.3f G bounces/s bounces %d hash %016x' % (sys.argv[1], sys.argv[2], t * 1e3, n / t / 1e9, n, "
        "int(np.bitwise_xor.reduce(img.view(np.uint32).astype(np.uint64) * np.arange(1, img.size + 1, dtype=np.uint64)))), flush=True)\n")
for rnd in range(2):
    for lib in libs:
        subprocess.run([sys.executable, "-c", code, os.path.basename(lib), scene, spp],
                       env=dict(os.environ, PT_LIB=os.path.abspath(lib)), check=True)
