"""One frame (best of 2) per environment setting, each in a process of its own (the tuning variables are read when a context is
created): python tools/env_sweep.py <scene> <W> <H> <spp> "PT_STREAMS=8192,PT_RAYS_PER_PASS=805306368" "-" ...   ("-" = none)"""
import os, subprocess, sys
scene, W, H, spp, sets = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5:]
code = ("import sys; sys.path.insert(0, 'tools'); import ab_flags as f; import numpy as np\n"
        "f.W, f.H = int(sys.argv[4]), int(sys.argv[5])\n"
        "img, n, t = f.render(sys.argv[2], int(sys.argv[3]), 0, 0, reps=2)\n"
        "print('%s %sx%s @%s %s: %.1f ms %.3f G bounces/s' % (sys.argv[2], sys.argv[4], sys.argv[5], sys.argv[3], sys.argv[1], t * 1e3, n / t / 1e9), flush=True)\n")
for st in sets:
    env = dict(os.environ)
    if st != "-":
        for kv in st.split(","):
            k, v = kv.split("=")
            env[k] = v
    subprocess.run([sys.executable, "-c", code, st, scene, spp, W, H], env=env, check=True)
