"""Frame time under different environment settings: python tools/env_probe.py <scene> <spp> <lib.so> VAR=a,b,c"""
import os, subprocess, sys
scene, spp, lib, spec = sys.argv[1:5]
var, vals = spec.split("=")
code = ("import sys; sys.path.insert(0, 'tools'); import ab_flags as f; import numpy as np\n"
        "img, n, t = f.render(sys.argv[2], int(sys.argv[3]), 0, 0, reps=2)\n"
        "print('%s %s: %.1f ms %.3f G bounces/s bounces %d hash %016x' % (sys.argv[1], sys.argv[2], t * 1e3, n / t / 1e9, n, "
        "int(np.bitwise_xor.reduce(img.view(np.uint32).astype(np.uint64) * np.arange(1, img.size + 1, dtype=np.uint64)))), flush=True)\n")
for v in vals.split(","):
    env = dict(os.environ, PT_LIB=os.path.abspath(lib))
    if v != "-":
        env[var] = v
    subprocess.run([sys.executable, "-c", code, "%s=%s" % (var, v), scene, spp], env=env, check=True)
