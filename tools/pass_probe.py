import os, subprocess, sys
code = ("import sys; sys.path.insert(0, 'tools'); import ab_flags as f; import numpy as np\n"
        "img, n, t = f.render(sys.argv[2], int(sys.argv[3]), 0, 0, reps=2)\n"
        "print('PT_PASS_KERNEL=%s %s: %.1f ms %.3f G bounces/s bounces %d hash %016x' % (sys.argv[1], sys.argv[2], t * 1e3, n / t / 1e9, n, "
        "int(np.bitwise_xor.reduce(img.view(np.uint32).astype(np.uint64) * np.arange(1, img.size + 1, dtype=np.uint64)))))\n")
for scene, spp in (("cornell", "64"), ("cornell", "1024"), ("three-spheres", "256")):
    for v in ("0", "1", "0", "1"):
        subprocess.run([sys.executable, "-c", code, v, scene, spp], env=dict(os.environ, PT_PASS_KERNEL=v), check=True)
