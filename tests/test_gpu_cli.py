"""The C++ host (`ptrace <spp> <res_y> <scene>`, the reference's dead cmd_render.rs UX) end to end on the GPU:
scene lookup by id and by index, width = res_y*3/2, P3 file + latest.ppm symlink, pixels = the oracle's frame."""
import os
import subprocess

import numpy as np
import pytest

import ptlib

ROOT = ptlib.ROOT

pytestmark = pytest.mark.gpu
CLI = os.path.join(ptlib.PKG, "ptrace")


def read_ppm(path):
    data = open(path).read()
    head, body = data.split("255\n", 1)
    lines = head.split("\n")
    assert lines[0] == "P3"
    w, h = [int(v) for v in lines[3].split()]
    vals = np.array(body.split(), dtype=np.int64).reshape(h * w, 3)
    return lines, w, h, vals


@pytest.mark.parametrize("scene_arg,sid", [("cornell", "cornell"), ("1", "cornell")])  # index 1 of the sorted listing
def test_cli_matches_oracle(tmp_path, scene_arg, sid):
    if not os.path.exists(CLI):
        pytest.skip("CLI not built")
    out = tmp_path / "out"
    r = subprocess.run([CLI, "6", "24", scene_arg, "--root", ptlib.ROOT, "--seed", "3", "--out", str(out), "--gpus", "2"],
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Rendering scene %s (11 objects), 6 samples per pixel, 36x24 resolution" % sid in r.stdout
    files = [f for f in os.listdir(out) if f.endswith(".ppm")]
    assert len(files) == 1 and files[0].endswith("-scene-%s-spp6-res24-.ppm" % sid)
    assert os.path.islink(tmp_path / "latest.ppm")
    lines, w, h, vals = read_ppm(str(out / files[0]))
    assert (w, h) == (36, 24) and lines[1] == "# samplesPerPixel: 6, resolution_y: 24, scene_id: %s" % sid
    sc = ptlib.load_scene_py(ptlib.scene_path(sid))
    img, _, _ = ptlib.oracle_render(sc, 36, 24, 6, 3)
    O = ptlib.oracle()
    want = np.array([[O.pto_to_int_with_gamma_correction(float(v)) for v in px] for px in img[::-1]], dtype=np.int64)
    diff = np.abs(vals - want)
    assert diff.max() <= 1 and (diff > 0).mean() < 0.002  # 1e-7 radiance differences may flip a rounding


def test_cli_errors(tmp_path):
    if not os.path.exists(CLI):
        pytest.skip("CLI not built")
    assert subprocess.run([CLI], capture_output=True).returncode == 1
    r = subprocess.run([CLI, "4", "16", "no-such-scene", "--root", ptlib.ROOT], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot load scene" in r.stderr
    r = subprocess.run([CLI, "4", "16", "99", "--root", ptlib.ROOT], capture_output=True, text=True)
    assert r.returncode == 1 and "out of range" in r.stderr


def test_cli_fills_an_empty_scenes_directory(tmp_path):
    """load_scene_ids (scenes.rs:28-38): with no scenes/*.json the built-in scenes are written out first; the frame
    rendered from the generated cornell.json is the frame of the shipped one."""
    if not os.path.exists(CLI):
        pytest.skip("CLI not built")
    root = tmp_path / "root"
    (root / "meshes").mkdir(parents=True)
    os.symlink(os.path.join(ptlib.ROOT, "meshes", "mctri.off"), root / "meshes" / "mctri.off")
    out = tmp_path / "out"
    r = subprocess.run([CLI, "4", "16", "cornell", "--root", str(root), "--seed", "5", "--out", str(out)],
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert sorted(os.listdir(root / "scenes")) == sorted(s + ".json" for s in
                                                         ["single-sphere", "cartesian", "two-spheres", "three-spheres",
                                                          "cornell", "mesh"])
    out2 = tmp_path / "out2"
    r = subprocess.run([CLI, "4", "16", "cornell", "--root", ptlib.ROOT, "--seed", "5", "--out", str(out2)],
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    a = read_ppm(str(out / os.listdir(out)[0]))[3]
    b = read_ppm(str(out2 / os.listdir(out2)[0]))[3]
    assert (a == b).all()


@pytest.mark.parametrize("gather", ["torch", "abi"])
def test_bench_collective_at_world_size_1(gather):
    """The collective branch of bench.py on hardware with the one GPU this box has: --force-collective initialises
    torch.distributed with backend "nccl" (= RCCL) and runs all_gather_into_tensor + the strided un-permute (--gather
    torch), or the C ABI's own communicator and pt_comm_gather_frame (--gather abi), at world size 1; the frame must be
    the one the plain run gives (bench.py compares the variants' images with the main run's itself, so here: same
    bounce count and a collective named in the line)."""
    import json
    import subprocess
    import sys

    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--spp", "8", "--width", "128",
            "--height", "96", "--no-variants", "--no-cpu-baseline"]
    plain = subprocess.run(base, capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0, plain.stderr[-3000:]
    coll = subprocess.run(base + ["--force-collective", "--gather", gather], capture_output=True, text=True, timeout=600)
    assert coll.returncode == 0, coll.stderr[-3000:]
    a = json.loads(plain.stdout.strip().splitlines()[-1])
    b = json.loads(coll.stdout.strip().splitlines()[-1])
    assert a["config"]["collective"] is None and b["config"]["collective"]
    assert ("pt_comm_gather_frame" in b["config"]["collective"]) == (gather == "abi")
    assert a["config"]["ray_bounces_per_frame"] == b["config"]["ray_bounces_per_frame"] > 0
    assert b["config"]["image_hash"] == a["config"]["image_hash"]
    # the diagnosis a multi-GPU run will need is in the line: per-rank render / gather times and bounces (min, max, mean over
    # the ranks - here one), the gather's own time, and the identity of the kernels the profiles must have been measured on
    assert a["per_rank"] is None and a["gather_ms"] is None and a["render_ms"] > 0
    pr = b["per_rank"]
    for k in ("render_ms", "gather_ms", "ray_bounces"):
        assert pr[k]["min"] == pr[k]["max"] == pr[k]["mean"]
    assert pr["ray_bounces"]["mean"] == b["config"]["ray_bounces_per_frame"] and pr["render_ms"]["mean"] > 0
    assert b["gather_ms"] is not None and 0 <= b["gather_ms"] < 1e3 and abs(b["gather_ms"] - pr["gather_ms"]["mean"]) < 1e-6
    assert len(b["config"]["kernel_isa_hash"]) == 16 and b["config"]["build_flags_complete"] is True
    assert isinstance(b["roofline"]["profile_matches_binary"], bool) and "hbm_counter_frac" in b["roofline"] and b["roofline"]["l2_frac"] > 0


def test_bench_two_rank_rehearsal_reports_per_rank_times():
    """bench.py with TWO ranks on this box's one GPU (PT_BENCH_DIST_BACKEND=gloo: the ranks share the device and the rows travel
    through host memory - a rehearsal of the N > 1 code path, not a measurement): interleaved rows, the all-gather, the
    reassembled frame - the single rank's image hash and bounce count - and the per-rank diagnosis line (render / gather
    milliseconds and bounces as min / max / mean over the ranks, gathered once after the timed region)."""
    import json
    import subprocess
    import sys

    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--spp", "16", "--width", "128",
            "--height", "96", "--no-variants", "--no-cpu-baseline"]
    one = subprocess.run(base, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    env = dict(os.environ, PT_BENCH_DIST_BACKEND="gloo")
    two = subprocess.run(base + ["--gpus", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert two.returncode == 0, (two.stdout + two.stderr)[-3000:]
    a = json.loads(one.stdout.strip().splitlines()[-1])
    b = json.loads([l for l in two.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert b["n_gpus"] == 2 and b["scaling"] == "strong" and "rehearsal" in b["config"]["collective"]
    assert b["config"]["image_hash"] == a["config"]["image_hash"]
    assert b["config"]["ray_bounces_per_frame"] == a["config"]["ray_bounces_per_frame"]
    pr = b["per_rank"]
    for k in ("render_ms", "gather_ms", "ray_bounces"):
        assert pr[k]["min"] <= pr[k]["mean"] <= pr[k]["max"], (k, pr[k])
    assert abs(2 * pr["ray_bounces"]["mean"] - b["config"]["ray_bounces_per_frame"]) < 1.0  # the ranks' rows make up the frame
    assert pr["render_ms"]["min"] > 0 and b["gather_ms"] is not None


@pytest.mark.gpu
def test_pass_kernel_lds_keeps_its_workgroups_per_cu():
    """k_pass_cand without walks is built for FIVE workgroups per CU, with walks for four: that holds while a workgroup's LDS stays
    under what `PT_LDS_PAD` measured (five workgroups of 32 144 B share a CU, five of 32 400 B do not; four of 40 928 B do) - a
    tenth of the frame rate otherwise, and nothing else would notice.  The library says what it asks for when PT_LDS_PAD is set."""
    import re
    import sys

    tool = os.path.join(ptlib.ROOT, "tools", "one_frame.py")
    for scene, spp, kernel, limit in (("cornell", 683, "k_pass_cand:", 32256), ("cornell", 64, "k_pass_cand:", 32256),
                                      ("mesh", 512, "k_pass_cand<BVH>:", 40960), ("mesh", 64, "k_pass_cand<BVH>:", 40960)):
        r = subprocess.run([sys.executable, tool, scene, str(spp)], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, PT_LDS_PAD="0"))
        assert r.returncode == 0, r.stderr
        m = re.search(re.escape(kernel) + r" (\d+) bytes of LDS per workgroup", r.stderr)
        assert m, r.stderr
        assert int(m.group(1)) <= limit, (scene, spp, m.group(0))
