#!/usr/bin/env python3
"""Turn the three PMC passes of tools/pmc_run.sh <tag> into profiles/<tag>_pmc_<what>.json and
profiles/r01_intersect_traffic.json (read by bench.py for roofline.traffic).
usage: python tools/make_traffic_json.py <tag> <what, e.g. cornell_1024x768_128spp> "<bench args of the run>" """
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarise

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, what, bench_args = sys.argv[1], sys.argv[2], sys.argv[3]
out = {k: summarise(os.path.join(ROOT, "gpurun_out", "pmc_%s_%s" % (tag, k))) for k in ("sq", "fetch", "write")}
line = [l for l in open(os.path.join(ROOT, "gpurun_out", "pmc_%s_sq.log" % tag)) if l.startswith("{")][-1]
rays = json.loads(line)["config"]["ray_bounces_per_frame"]
name = "%s_pmc_%s.json" % (tag, what)
with open(os.path.join(ROOT, "profiles", name), "w") as f:
    json.dump({"command": "tools/pmc_run.sh %s %s" % (tag, bench_args), "rays_per_frame": rays, "passes": out}, f, indent=1)


def kernel(d, frag):
    return next((v for k, v in d.items() if frag in k), None)


NOTE = ("SQ_INSTS_VALU x 64 / rays; SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES / 8 (8 = all 32 SIMDs of a shader engine issuing "
        "a VALU instruction every quad-cycle): the kernel is bound by VALU instruction issue, not by HBM")
done = {}
for kname in ("k_pass", "k_intersect", "k_shade"):
    sq = kernel(out["sq"], kname)
    if sq is None:
        continue
    fetch = kernel(out["fetch"], kname)["FETCH_SIZE"] * 1024.0  # counter is in KiB
    write = kernel(out["write"], kname)["WRITE_SIZE"] * 1024.0
    tr = {
        "kernel": kname,
        "rays": rays,
        "dispatches": sq["dispatches"],
        "FETCH_SIZE_bytes": fetch,
        "WRITE_SIZE_bytes": write,
        "hbm_bytes_per_ray": (2.0 * fetch + write) / rays,
        "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests of wide coalesced reads at 64 B, "
                      "MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; both in KiB",
        "source": "profiles/%s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py "
                  "--steps 1 --warmup 0 %s`" % (name, bench_args),
        "valu": {
            "insts_per_ray": sq["SQ_INSTS_VALU"] * 64.0 / rays,
            "busy_frac": sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_BUSY_CYCLES"] / 8.0,
            "note": NOTE,
        },
    }
    done[kname] = tr
    if kname != "k_shade":  # bench.py reads r01_<dominant kernel>_traffic.json
        with open(os.path.join(ROOT, "profiles", "r01_%s_traffic.json" % kname), "w") as f:
            json.dump(tr, f, indent=1)
print(json.dumps(done, indent=1))
