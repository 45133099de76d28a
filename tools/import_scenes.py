#!/usr/bin/env python3
"""Import the reference's scene DATA files (scenes/*.json, meshes/mctri.off) as compact fixtures.

The reference loads `scenes/{id}.json` (src/render/mod.rs:93-97) and MeshFile paths such as
`meshes/mctri.off` (scenes/mesh.json:6-9) relative to the CWD; this repo keeps the same layout so the
same scene ids work.  The files are data (serde dumps of SceneDescriptor / an OFF mesh), re-emitted
here in compact form with every number written as the shortest decimal that round-trips its f32 value
(serde_json reads f64 then casts to f32, OFF numbers are parsed straight to f32).
tests/test_scene_io.py re-checks the committed files against /root/reference when that exists.

Usage: python tools/import_scenes.py [/root/reference]
"""
import json
import os
import sys

import numpy as np


def f32s(x):
    v = np.float32(x)
    s = np.format_float_positional(v, unique=True, trim="0")
    if len(s) > 20:
        s = np.format_float_scientific(v, unique=True, trim="0")
    assert np.float32(float(s)) == v or (np.isnan(v)), (x, s)
    return s


class F(float):
    def __repr__(self):
        return f32s(float(self))


def conv(o):
    if isinstance(o, float):
        return F(o)
    if isinstance(o, list):
        return [conv(x) for x in o]
    if isinstance(o, dict):
        return {k: conv(v) for k, v in o.items()}
    return o


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    enc = json.JSONEncoder(separators=(",", ":"))
    json.encoder.float_repr = None
    for name in sorted(os.listdir(os.path.join(ref, "scenes"))):
        if not name.endswith(".json"):
            continue
        d = json.load(open(os.path.join(ref, "scenes", name)))
        # floats are emitted through F.__repr__ (json uses float.__repr__ for subclasses)
        out = enc.encode(conv(d))
        # one object per line keeps diffs readable
        out = out.replace('{"type_"', '\n{"type_"').replace('],"camera"', '\n],"camera"')
        with open(os.path.join(root, "scenes", name), "w") as f:
            f.write(out + "\n")
        print("wrote scenes/" + name, len(out), "bytes")
    # OFF: keep header/counts, normalise numbers
    src = [l.strip() for l in open(os.path.join(ref, "meshes", "mctri.off"))]
    lines = [l for l in src if l and not l.startswith("#")]
    assert lines[0] == "OFF"
    nv, nf, ne = map(int, lines[1].split())
    out = ["OFF", "# mctri: %d vertices, %d triangles (imported from the reference's meshes/mctri.off)" % (nv, nf),
           "%d %d %d" % (nv, nf, ne)]
    for l in lines[2:2 + nv]:
        out.append(" ".join(f32s(np.float32(t)) for t in l.split()))
    for l in lines[2 + nv:2 + nv + nf]:
        t = l.split()
        assert t[0] == "3"
        out.append(" ".join(t[:4]))
    with open(os.path.join(root, "meshes", "mctri.off"), "w") as f:
        f.write("\n".join(out) + "\n")
    print("wrote meshes/mctri.off", nv, nf)
    # hdodec.off: polygon faces (the reference cannot load it; used by the PT_LOAD_TRIANGULATE extension)
    src = [l.strip() for l in open(os.path.join(ref, "meshes", "hdodec.off"))]
    lines = [l for l in src if l and not l.startswith("#")]
    nv, nf, ne = map(int, lines[1].split())
    out = ["OFF",
           "# hdodec: %d vertices, %d pentagon faces (imported from the reference's meshes/hdodec.off; the reference's" % (nv, nf),
           "# load_off rejects non-triangle faces, this repo can fan-triangulate them on request: PT_LOAD_TRIANGULATE)",
           "%d %d %d" % (nv, nf, ne)]
    for l in lines[2:2 + nv]:
        out.append(" ".join(f32s(np.float32(t)) for t in l.split()))
    for l in lines[2 + nv:2 + nv + nf]:
        t = l.split()
        out.append(" ".join(t[:1 + int(t[0])]))
    with open(os.path.join(root, "meshes", "hdodec.off"), "w") as f:
        f.write("\n".join(out) + "\n")
    # a scene that uses it: mesh.json with the mesh file swapped (not a reference scene)
    d = json.load(open(os.path.join(ref, "scenes", "mesh.json")))
    d["id"] = "mesh-hdodec"
    d["objects"][0]["type_"] = {"MeshFile": {"path": "meshes/hdodec.off", "scale": 0.9}}
    d["objects"][0]["position"] = [0.0, -1.0, 0.0]
    out = enc.encode(conv(d)).replace('{"type_"', '\n{"type_"').replace('],"camera"', '\n],"camera"')
    with open(os.path.join(root, "scenes", "mesh-hdodec.json"), "w") as f:
        f.write(out + "\n")
    print("wrote meshes/hdodec.off and scenes/mesh-hdodec.json")


if __name__ == "__main__":
    main()
