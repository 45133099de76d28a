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
    return next(v for k, v in d.items() if frag in k)


ki = kernel(out["sq"], "k_intersect")
fetch = kernel(out["fetch"], "k_intersect")["FETCH_SIZE"] * 1024.0  # counter is in KiB
write = kernel(out["write"], "k_intersect")["WRITE_SIZE"] * 1024.0
ks = kernel(out["sq"], "k_shade")
fs = kernel(out["fetch"], "k_shade")["FETCH_SIZE"] * 1024.0
ws = kernel(out["write"], "k_shade")["WRITE_SIZE"] * 1024.0
tr = {
    "kernel": "k_intersect",
    "rays": rays,
    "dispatches": ki["dispatches"],
    "FETCH_SIZE_bytes": fetch,
    "WRITE_SIZE_bytes": write,
    "hbm_bytes_per_ray": (2.0 * fetch + write) / rays,
    "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests of wide coalesced reads at 64 B, MI355X_MICROARCH.md "
                  "HBM section); WRITE_SIZE as reported; both in KiB",
    "source": "profiles/%s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 1 "
              "--warmup 0 %s`" % (name, bench_args),
    "valu": {
        "insts_per_ray": ki["SQ_INSTS_VALU"] * 64.0 / rays,
        "busy_frac": ki["SQ_ACTIVE_INST_VALU"] / ki["SQ_BUSY_CYCLES"] / 8.0,
        "note": "SQ_INSTS_VALU x 64 / rays; SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES / 8 (8 = all 32 SIMDs of a shader "
                "engine issuing a VALU instruction every quad-cycle): the kernel is bound by VALU instruction issue, "
                "not by HBM",
    },
    "k_shade": {
        "insts_per_ray": ks["SQ_INSTS_VALU"] * 64.0 / rays,
        "busy_frac": ks["SQ_ACTIVE_INST_VALU"] / ks["SQ_BUSY_CYCLES"] / 8.0,
        "hbm_bytes_per_ray": (2.0 * fs + ws) / rays,
    },
}
with open(os.path.join(ROOT, "profiles", "r01_intersect_traffic.json"), "w") as f:
    json.dump(tr, f, indent=1)
print(json.dumps(tr, indent=1))
