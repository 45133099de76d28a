#!/usr/bin/env python3
"""The PPM the reference writes with MOCK_RANDOM = true, from the oracle: the one frame of the reference that is
reproducible run to run, so anyone holding a cargo toolchain can pin the oracle's render_pixel / radiance / Triangle /
tent-filter / PPM restatement on a whole frame (the reference's 7 unit tests do not reach those).

    python tools/mock_reference_ppm.py <scene-id> <res_y> <spp> > mock.ppm        # width = res_y*3/2 (mod.rs:872-879)

Reference side (not possible in this image: no rustc): set `const MOCK_RANDOM: bool = true;` in src/render/mod.rs:31,
`cargo run --release`, pick the same scene / samples / resolution in the GUI, render once right after start-up (the
counter MOCK_RANDOMS_INDEX is process-global and starts at 0, mod.rs:45), then
    tail -n +4 latest.ppm | cmp - <(tail -n +4 mock.ppm)
(the first three lines hold the rendering time).  A mismatch in the first differing pixel index, read from the end of
the file (pixels are written in reverse, mod.rs:1064), says which pixel - and with it which branch - differs.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ptlib  # noqa: E402


def main():
    sid, res_y, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    width = res_y * 3 // 2
    sc = ptlib.load_scene_py(ptlib.scene_path(sid))
    img, cnt, draws = ptlib.oracle_render_mock(sc, width, res_y, spp)
    L = ptlib.oracle()
    n = L.pto_format_ppm(ptlib._np_f(img), width, res_y, spp, sid.encode(), 0, None, 0)
    buf = C.create_string_buffer(n)
    L.pto_format_ppm(ptlib._np_f(img), width, res_y, spp, sid.encode(), 0, buf, n)
    sys.stdout.buffer.write(buf.raw[:n])
    print("%s %dx%d @%d spp: %d ray bounces, %d rand01() calls" % (sid, width, res_y, spp, cnt.ray_bounces, draws),
          file=sys.stderr)


if __name__ == "__main__":
    main()
