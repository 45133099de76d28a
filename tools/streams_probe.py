"""Tuning probe: wavefront frame time against the stream-count target (PT_STREAMS) and the rays per pass.
usage: python tools/streams_probe.py streams <K> ... | rpp <rays_per_pass> ..."""
import os, subprocess, sys
mode, vals = sys.argv[1], sys.argv[2:]
code = ("import sys, ctypes as C; sys.path.insert(0, 'tools'); import ab_flags as f\n"
        "rpp = int(sys.argv[2])\n"
        "_r = f.PtConfig\n"
        "def cfg(*a):\n"
        "    c = _r(*a); c.rays_per_pass = rpp; return c\n"
        "f.PtConfig = cfg\n"
        "img, n, t = f.render(sys.argv[3], int(sys.argv[4]), 0, 0, reps=2)\n"
        "print('%s streams=%s rpp=%s: %.1f ms %.3f G bounces/s' % (sys.argv[3], sys.argv[1], sys.argv[2], t * 1e3, n / t / 1e9))\n")
scene = os.environ.get("SCENE", "cornell"); spp = os.environ.get("SPP", "1024")
for v in vals:
    env = dict(os.environ)
    k, rpp = ("auto", v) if mode == "rpp" else (v, "0")
    if mode == "streams":
        env["PT_STREAMS"] = v
    subprocess.run([sys.executable, "-c", code, k, rpp, scene, spp], env=env, check=True)
