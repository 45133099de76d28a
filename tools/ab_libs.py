"""A/B of two builds of libptrace_hip.so on one GPU, alternating, same frame: python tools/ab_libs.py a.so b.so [scene] [spp] [rounds]"""
import os
import subprocess
import sys

a, b = os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])
scene = sys.argv[3] if len(sys.argv) > 3 else "cornell"
spp = sys.argv[4] if len(sys.argv) > 4 else "1024"
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
code = ("import sys; sys.path.insert(0, 'tools'); import ab_flags as f; import numpy as np\n"
        "img, n, t = f.render(%r, %s, 0, 0, reps=2)\n"
        "print('%%s %%.1f ms %%.3f G bounces/s bounces %%d hash %%016x' %% (sys.argv[1], t * 1e3, n / t / 1e9, n, "
        "int(np.bitwise_xor.reduce(img.view(np.uint32).astype(np.uint64) * np.arange(1, img.size + 1, dtype=np.uint64)))))\n"
        % (scene, spp))
for r in range(rounds):
    for lib in (a, b):
        subprocess.run([sys.executable, "-c", code, os.path.basename(lib)], env=dict(os.environ, PT_LIB=lib), check=True)
