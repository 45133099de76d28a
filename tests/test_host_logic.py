"""Host-side pieces of the device layout that can be checked without a GPU: the round-robin deal of pixels to ray
streams (pt_device.h: stream_pixel / stream_pixel_count / global_pixel) and the bookkeeping word."""
import ctypes as C
import os
import subprocess
import textwrap

import ptlib

SRC = textwrap.dedent(r"""
    #include <cstdio>
    #include <cstdint>
    #include <vector>
    #include "pt_device.h"
    using namespace pt;
    struct P { uint32_t idx_begin, chunk_pixels, chunk_first, chunk_step, k_begin; };
    int main() {
        // 1. every pixel of a call belongs to exactly one (stream, slot); the accumulator slot k_resolve reads back is
        //    the one k_pass / k_shade wrote
        const uint32_t cases[][2] = {{1, 1}, {7, 3}, {786432, 16050}, {98304, 16384}, {3072, 1536}, {1000, 999}, {5, 64}};
        for (auto &c : cases) {
            const uint32_t npix = c[0], K = c[1];
            const uint32_t m = (npix + K - 1) / K;
            std::vector<int> seen(npix, 0);
            uint64_t total = 0;
            for (uint32_t b = 0; b < K; ++b) {
                const uint32_t mb = stream_pixel_count(npix, K, b);
                if (mb > m) { printf("FAIL count %u %u %u\n", npix, K, b); return 1; }
                total += mb;
                for (uint32_t j = 0; j < mb; ++j) {
                    const uint32_t p = stream_pixel(K, b, j);
                    if (p >= npix || seen[p]++) { printf("FAIL deal %u %u\n", npix, K); return 1; }
                    if (p % K != b || p / K != j) { printf("FAIL resolve %u %u\n", npix, K); return 1; }  // k_resolve's inverse
                }
            }
            if (total != npix) { printf("FAIL total %u %u\n", npix, K); return 1; }
        }
        // 2. interleaved partition: ranks' global pixels are disjoint and cover the band
        {
            const uint32_t W = 37, H = 11, step = 3;
            std::vector<int> seen(W * H, 0);
            for (uint32_t r = 0; r < step; ++r) {
                P f{0, W, r, step};
                uint32_t owned = 0;
                for (uint32_t ck = r; ck * W < W * H; ck += step) owned += W;
                for (uint32_t k = 0; k < owned; ++k) {
                    const uint32_t g = global_pixel(f, k);
                    if (g >= W * H || seen[g]++) { printf("FAIL chunks\n"); return 1; }
                }
            }
            for (int v : seen) if (v != 1) { printf("FAIL cover\n"); return 1; }
        }
        // 3. the bookkeeping word round-trips at its limits
        const uint32_t w = pack_word(1023, 32767, 11, 7);
        if (word_pix(w) != 1023 || word_sample(w) != 32767 || word_depth(w) != 11 || word_branch(w) != 7) { printf("FAIL word\n"); return 1; }
        printf("OK\n");
        return 0;
    }
""")


def test_stream_deal_and_partition(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = str(tmp_path / "t")
    inc = os.path.join(ptlib.PKG, "csrc")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", inc, "-I", os.path.join(ptlib.ROOT, "include"), str(src),
                           "-o", exe])
    assert subprocess.check_output([exe]).decode().strip() == "OK"


def test_rust_shim_mirrors_the_header():
    """ffi/hip.rs cannot be compiled here (no rustc): hold its #[repr(C)] structs to include/ptrace.h field by field
    (names and order; emission is the header's spelling of the reference's `emmission`)."""
    import re

    root = ptlib.ROOT
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "ptrace.h")).read(), flags=re.S)
    rust = re.sub(r"//[^\n]*", "", open(os.path.join(root, "ffi", "hip.rs")).read())
    for c_name, r_name in (("pt_camera", "PtCamera"), ("pt_triangle", "PtTriangle"), ("pt_object", "PtObject"),
                           ("pt_config", "PtConfig"), ("pt_stats", "PtStats")):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (c_name, c_name), header, flags=re.S).group(1)
        c_fields = re.findall(r"\b(?:float|double|uint32_t|uint64_t)\s+(\w+)", body)
        rbody = re.search(r"pub struct %s \{(.*?)\n\}" % r_name, rust, flags=re.S).group(1)
        r_fields = re.findall(r"pub (\w+):", rbody)
        assert c_fields == r_fields, (c_name, c_fields, r_fields)
    for name in ("PT_OK", "PT_CANCELLED"):
        c_val = int(re.search(r"#define %s \(?(-?\d+)\)?" % name, header).group(1))
        r_val = int(re.search(r"pub const %s: i32 = (-?\d+);" % name, rust).group(1))
        assert c_val == r_val
    assert len(re.findall(r"pub fn pt_render\(", rust)) == 1 and "pub fn flatten(" in rust
    # every function the shim binds is declared in the header with the same number of parameters, in the same order by
    # kind (pointer / integer / float), and the preview path uses the resident form
    ext = re.search(r'extern "C" \{(.*?)\n\}', rust, flags=re.S).group(1)
    bound = re.findall(r"pub fn (\w+)\((.*?)\)\s*(?:->\s*([\w\s\*]+))?;", ext, flags=re.S)
    assert {"pt_ctx_create", "pt_ctx_destroy", "pt_ctx_set_scene", "pt_ctx_render", "pt_ctx_snapshot", "pt_device_malloc",
            "pt_device_free", "pt_device_download", "pt_render"} <= {b[0] for b in bound}

    def kind_rust(t):
        t = t.strip()
        return "p" if t.startswith("*") or t.startswith("Option<") else ("f" if t in ("f32", "f64") else "i")

    def kind_c(t):
        t = t.strip()
        return "p" if "*" in t or "[" in t or t.startswith("pt_progress_fn") else ("f" if t.split()[0] in ("float", "double") else "i")

    for fname, params, _ in bound:
        m = re.search(r"\b%s\((.*?)\);" % fname, header, flags=re.S)
        assert m, fname + " is not declared in include/ptrace.h"
        c_params = [q for q in m.group(1).split(",") if q.strip() and q.strip() != "void"]
        r_params = [q.split(":", 1)[1] for q in params.split(",") if ":" in q]
        assert [kind_c(q) for q in c_params] == [kind_rust(q) for q in r_params], (fname, c_params, r_params)
    body = rust[rust.index("pub fn render_pixels_hip"):]
    for call in ("pt_ctx_create", "pt_ctx_set_scene", "pt_ctx_render", "pt_device_download", "pt_ctx_destroy"):
        assert call + "(" in body, call
    assert "pt_ctx_snapshot(" in rust[rust.index('extern "C" fn on_progress'):rust.index("pub fn render_pixels_hip")]

FLATTEN_SRC = r"""
#include <cstdio>
#include <cstdint>
#include <string>
#include <vector>
#include "ptrace.h"
#include "pt_host.h"
using namespace pt;
static int depth_of(const host::FlatScene &fs, int32_t ref, std::vector<int> &leaf_hits, uint32_t pair_lo, uint32_t pair_hi, bool &ok) {
    if (ref < 0) {
        const uint32_t code = (uint32_t)~ref, first = leaf_first(code), cnt = leaf_count(code);
        if (cnt < 1 || cnt > kBvhLeafPairs || first < pair_lo || first + cnt > pair_hi) ok = false;
        for (uint32_t r = 0; r < cnt && ok; ++r)
            for (int hf = 0; hf < 2; ++hf)
                if (fs.tri_pairs[first + r].id[hf] != kNoTri) leaf_hits[fs.tri_pairs[first + r].id[hf]]++;
        return 0;
    }
    const BvhNode &n = fs.bvh_nodes[ref];
    const int a = depth_of(fs, n.c[0], leaf_hits, pair_lo, pair_hi, ok), b = depth_of(fs, n.c[1], leaf_hits, pair_lo, pair_hi, ok);
    return 1 + (a > b ? a : b);
}
// the four-wide tree below ref4: every leaf reference it holds, and that each of its boxes is a box of the binary tree
// (the child boxes of some binary node: collected in `bin_boxes` as lo.x of each, a cheap identity)
static void walk4(const host::FlatScene &fs, int32_t ref4, std::vector<int32_t> &leaves, bool &ok, int depth, int &deepest) {
    if (ref4 < 0) { leaves.push_back(ref4); return; }
    if ((size_t)ref4 >= fs.bvh_nodes4.size() || depth > 64) { ok = false; return; }
    deepest = depth > deepest ? depth : deepest;
    const BvhNode4 &n = fs.bvh_nodes4[(size_t)ref4];
    int live = 0;
    for (int j = 0; j < 4; ++j) {
        const bool filler = n.lox[j] != n.lox[j];  // NaN
        if (filler) {
            // a filler box is NaN in every bound (never hit: pt_device.h) and repeats child 0's reference; fillers come last
            if (n.hix[j] == n.hix[j] || n.loy[j] == n.loy[j] || n.hiz[j] == n.hiz[j] || n.c[j] != n.c[0] || j < 2) ok = false;
            continue;
        }
        if (live != j) ok = false;
        ++live;
        bool found = false;  // the box is one of the binary tree's child boxes, with the same reference kind
        for (const BvhNode &b : fs.bvh_nodes)
            for (int h = 0; h < 2; ++h)
                if (b.lox[h] == n.lox[j] && b.loy[h] == n.loy[j] && b.loz[h] == n.loz[j] && b.hix[h] == n.hix[j] && b.hiy[h] == n.hiy[j] &&
                    b.hiz[h] == n.hiz[j] && ((b.c[h] < 0) == (n.c[j] < 0)) && (b.c[h] >= 0 || b.c[h] == n.c[j]))
                    found = true;
        if (!found) ok = false;
        walk4(fs, n.c[j], leaves, ok, depth + 1, deepest);
    }
}
static void leaves2(const host::FlatScene &fs, int32_t ref, std::vector<int32_t> &leaves) {
    if (ref < 0) { leaves.push_back(ref); return; }
    leaves2(fs, fs.bvh_nodes[(size_t)ref].c[0], leaves);
    leaves2(fs, fs.bvh_nodes[(size_t)ref].c[1], leaves);
}
int main(int argc, char **argv) {
    pt_scene *sc = nullptr;
    if (pt_scene_load(argv[1], argv[2], &sc) != 0) { printf("FAIL load %s\n", pt_last_error()); return 1; }
    uint32_t n_objs, n_tris;
    const pt_object *objs = pt_scene_objects(sc, &n_objs);
    const pt_triangle *tris = pt_scene_triangles(sc, &n_tris);
    host::FlatScene fs;
    std::string err;
    if (!host::flatten_scene(*pt_scene_camera(sc), objs, n_objs, tris, n_tris, fs, err)) { printf("FAIL flatten %s\n", err.c_str()); return 1; }
    // ranks: rank_id and tri_rank are inverse on the triangles; objects from the last to the first
    if (fs.tri_rank.size() < n_tris) { printf("FAIL tri_rank size\n"); return 1; }
    for (uint32_t k = 0; k < n_tris; ++k)
        if (fs.rank_id[fs.tri_rank[k]] != n_objs + k) { printf("FAIL tri_rank %u\n", k); return 1; }
    // the BVH mesh list: exactly the objects with a BVH, in visiting order
    size_t q = 0;
    uint32_t deepest = 0;
    for (uint32_t v = 0; v < n_objs; ++v) {
        const ObjRec &r = fs.objs[n_objs - 1u - v];
        if (r.kind != kKindMesh || r.bvh_root == kNoBvh) continue;
        if (q >= fs.bvh_meshes.size() || fs.bvh_meshes[q].root != r.bvh_root || fs.bvh_meshes[q].rr != r.rr ||
            fs.bvh_meshes[q].cx != r.cx) { printf("FAIL bvh_meshes %zu\n", q); return 1; }
        ++q;
        // the tree: every triangle of the mesh in exactly one leaf, leaves within the mesh's records, depth within the stack
        std::vector<int> hits(n_tris, 0);
        bool ok = true;
        const int d = depth_of(fs, r.bvh_root, hits, r.pair_begin, r.pair_begin + r.pair_count, ok);
        if (!ok) { printf("FAIL leaf\n"); return 1; }
        for (uint32_t k = 0; k < n_tris; ++k)
            if (hits[k] != ((k >= r.tri_begin && k < r.tri_begin + r.tri_count) ? 1 : 0)) { printf("FAIL cover %u\n", k); return 1; }
        deepest = (uint32_t)d > deepest ? (uint32_t)d : deepest;
        // the four-wide form of the same tree (the walk queue's): the same leaves in the same order, each once, every box one
        // of the binary tree's, at most half as deep (+1)
        std::vector<int32_t> l2, l4;
        leaves2(fs, r.bvh_root, l2);
        int deep4 = 0;
        walk4(fs, fs.bvh_meshes[q - 1].root4, l4, ok, 1, deep4);
        if (!ok || l2 != l4 || 2 * deep4 > d + 2) { printf("FAIL four-wide tree %d %zu %zu %d %d\n", (int)ok, l2.size(), l4.size(), deep4, d); return 1; }
        if (fs.bvh_nodes4.size() * 2 > fs.bvh_nodes.size() + 2) { printf("FAIL four-wide node count %zu %zu\n", fs.bvh_nodes4.size(), fs.bvh_nodes.size()); return 1; }
    }
    if (q != fs.bvh_meshes.size()) { printf("FAIL bvh_meshes count\n"); return 1; }
    if (q != 0 && (fs.bvh_stack < deepest + 1u || fs.bvh_stack > kBvhStack)) { printf("FAIL stack %u %u\n", fs.bvh_stack, deepest); return 1; }
    // the candidate records are those of the meshes without a BVH
    size_t want = 0;
    for (uint32_t i = 0; i < n_objs; ++i)
        if (fs.objs[i].kind == kKindMesh && fs.objs[i].bvh_root == kNoBvh) want += fs.objs[i].pair_count;
    if (fs.cand_pairs.size() != want || !fs.cand_ok) { printf("FAIL cand %zu %zu\n", fs.cand_pairs.size(), want); return 1; }
    printf("OK %zu %u %u\n", fs.bvh_meshes.size(), fs.bvh_stack, deepest);
    return 0;
}
"""


def test_flatten_tables_of_the_walk_queue(tmp_path):
    """flatten_scene's tables for k_pass_cand<.., BVH> on mesh.json and cornell.json: tri_rank inverts rank_id, the BVH mesh
    list holds the meshes with a BVH in visiting order, every triangle of such a mesh sits in exactly one leaf of at most
    kBvhLeafPairs records inside the mesh's record range, the advertised stack depth covers the tree, and the candidate
    records are exactly those of the meshes without a BVH."""
    src = tmp_path / "f.cpp"
    src.write_text(FLATTEN_SRC)
    exe = str(tmp_path / "f")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ptlib.PKG, "csrc"), "-I", os.path.join(ptlib.ROOT, "include"),
                           str(src), "-o", exe, "-L", ptlib.PKG, "-lptrace_hip", "-Wl,-rpath," + ptlib.PKG])
    out = subprocess.check_output([exe, ptlib.scene_path("mesh"), ptlib.ROOT]).decode().split()
    assert out[0] == "OK" and out[1] == "1" and int(out[2]) > int(out[3]) >= 5  # one BVH mesh; stack deeper than the tree
    out = subprocess.check_output([exe, ptlib.scene_path("cornell"), ptlib.ROOT]).decode().split()
    assert out[:2] == ["OK", "0"]


def test_bvh_reference_limit():
    """The BVH walkers pack a node index or a leaf code (first pair record << 1 | records - 1: leaves of one or two records) into 26 bits of a queue entry
    (csrc/pt_device.h: WalkQueue, LeafLds); flatten_scene refuses a scene beyond that instead of letting references wrap."""
    L = ptlib.product()
    L.pt_bvh_refs_fit.argtypes = [C.c_uint64, C.c_uint64]
    assert L.pt_bvh_refs_fit(141, 300) == 1                      # mesh.json
    assert L.pt_bvh_refs_fit((1 << 26) - 1, 1000) == 1
    assert L.pt_bvh_refs_fit(1 << 26, 1000) == 0                 # node indices need 27 bits
    assert L.pt_bvh_refs_fit(1000, (1 << 25) - 1) == 1
    assert L.pt_bvh_refs_fit(1000, 1 << 25) == 0                 # leaf codes need 27 bits


def test_build_flags_are_reported():
    L = ptlib.product()
    L.pt_build_flags.restype = C.c_char_p
    flags = L.pt_build_flags().decode()
    # "<set of the general unit> | flat: <set of k_pass_cand without walks>" (Makefile: MLLVM, MLLVM_FLAT)
    general, sep, flat = flags.partition("| flat:")
    assert sep and all(tok.startswith("-") or tok == "" for part in (general, flat) for tok in part.split(" ")), flags


SIGN_SRC = r"""
#include <cstdio>
#include <string>
#include <vector>
#include "ptrace.h"
#include "pt_host.h"
using namespace pt;
// argv[1] = 0: two quads in the plane z = 1 made of well-shaped triangles; 1: the same plane, but one triangle whose two edge
// products nearly cancel (a sliver: e1 = (1, 1), e2 = (1, 1.0005)): the plane normal's component is 0.0005 of the products
int main(int argc, char **argv) {
    const bool sliver = argv[1][0] == '1';
    pt_camera cam = {{0, 0, 5}, {0, 0, -1}, 0.035f, 0.036f, 1.5f};
    std::vector<pt_triangle> tris;
    tris.push_back({{0, 0, 1}, {1, 0, 1}, {0, 1, 1}});
    tris.push_back({{1, 0, 1}, {1, 1, 1}, {0, 1, 1}});
    if (sliver) tris[1] = {{0, 0, 1}, {1, 1, 1}, {1, 1.0005f, 1}};
    pt_object o{};
    o.kind = PT_MESH;
    o.tri_count = 2;
    o.bs_radius = 100.0f;
    host::FlatScene fs;
    std::string err;
    if (!host::flatten_scene(cam, &o, 1, tris.data(), 2, fs, err)) { printf("FAIL %s\n", err.c_str()); return 1; }
    if (fs.flat_pairs.size() != 1) { printf("FAIL flat_pairs %zu\n", fs.flat_pairs.size()); return 1; }
    printf("OK axis %u sign_exact %u\n", fs.flat_pairs[0].axis, fs.flat_pairs[0].sign_exact);
    return 0;
}
"""


def test_flat_filter_sign_rule_needs_well_shaped_triangles(tmp_path):
    """FlatPairRec.sign_exact (filter_flat drops rays that do not move towards the plane: the sign of Triangle::intersect's
    distance is then known exactly) is only set when the two edge products whose difference is the plane normal do not
    nearly cancel; a sliver keeps the conservative distance test.  cornell.json's walls all qualify."""
    src = tmp_path / "s.cpp"
    src.write_text(SIGN_SRC)
    exe = str(tmp_path / "s")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ptlib.PKG, "csrc"), "-I", os.path.join(ptlib.ROOT, "include"),
                           str(src), "-o", exe, "-L", ptlib.PKG, "-lptrace_hip", "-Wl,-rpath," + ptlib.PKG])
    assert subprocess.check_output([exe, "0"]).decode().split() == ["OK", "axis", "2", "sign_exact", "1"]
    assert subprocess.check_output([exe, "1"]).decode().split() == ["OK", "axis", "2", "sign_exact", "0"]


PLAN_SRC = r"""
#include <cstdio>
#include <cstdlib>
#include "ptrace.h"
#include "pt_host.h"
using namespace pt;
// argv: npix spp want default(0/1) stack_form stack_park cand_scan has_bvh streams per_stream wave_stack n_cus budget groups_per_cu
// prints the plan after following plan_pass's retries the way render_wavefront does: OK spp_pass m K cap bytes0 bytes1 retries
int main(int argc, char **argv) {
    if (argc != 15) return 2;
    host::PassPlanIn in;
    in.npix = strtoull(argv[1], 0, 10);
    in.spp = (uint32_t)strtoul(argv[2], 0, 10);
    in.want = strtoull(argv[3], 0, 10);
    in.want_is_default = argv[4][0] == '1';
    in.stack_form = argv[5][0] == '1';
    in.stack_park = argv[6][0] == '1';
    in.cand_scan = argv[7][0] == '1';
    in.has_bvh = argv[8][0] == '1';
    in.streams = strtoull(argv[9], 0, 10);
    in.per_stream = (uint32_t)strtoul(argv[10], 0, 10);
    in.wave_stack = (uint32_t)strtoul(argv[11], 0, 10);
    in.n_cus = (uint32_t)strtoul(argv[12], 0, 10);
    in.stack_budget = (size_t)strtoull(argv[13], 0, 10);
    in.groups_per_cu = (uint32_t)strtoul(argv[14], 0, 10);
    host::PassPlan p;
    int retries = 0;
    for (;;) {
        uint64_t next = in.want;
        const int rc = host::plan_pass(in, p, &next);
        if (rc == host::kPlanOk) break;
        if (rc == host::kPlanTooLarge) { printf("TOOLARGE\n"); return 0; }
        if (next >= in.want || ++retries > 64) { printf("STUCK\n"); return 1; }
        in.want = next;
    }
    printf("OK %u %u %u %u %zu %zu %d\n", p.spp_pass, p.m, p.K, p.cap, p.bytes0, p.bytes1, retries);
    return 0;
}
"""


def test_pass_plan(tmp_path):
    """host::plan_pass - how render_wavefront cuts a frame into passes and streams - on the CPU: the bench frame (six passes of
    683 samples, stacks of 1024 slots per wave, nothing in the second container), mesh.json (parking areas), a frame of few
    samples (at most 64 pixels per stream), the memory budget (passes halved until the stacks fit), small stacks on request,
    tiny passes (smaller stacks), the level-by-level forms (slices of 4 slots per primary ray, two containers), and a pass
    that 32-bit slot indices cannot hold."""
    src = tmp_path / "p.cpp"
    src.write_text(PLAN_SRC)
    exe = str(tmp_path / "p")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ptlib.PKG, "csrc"), "-I", os.path.join(ptlib.ROOT, "include"),
                           str(src), "-o", exe, "-L", ptlib.PKG, "-lptrace_hip", "-Wl,-rpath," + ptlib.PKG])

    def plan(npix, spp, want, default=1, stack=1, park=0, cand=1, bvh=0, streams=0, per_stream=0, wave_stack=0, n_cus=256, budget=0, groups=None):
        groups = groups or (4 if bvh or not stack else 5)  # workgroups a CU holds: k_pass_cand without walks runs five waves per SIMD
        out = subprocess.check_output([exe] + [str(v) for v in (npix, spp, want, default, stack, park, cand, bvh, streams, per_stream,
                                                                wave_stack, n_cus, budget, groups)]).decode().split()
        if out[0] != "OK":
            return out[0]
        spp_pass, m, K, cap, b0, b1, retries = map(int, out[1:])
        assert K * m >= npix > (K - 1) * m and 1 <= m <= 1024 and 1 <= spp_pass <= spp and cap % 4 == 0
        if stack:
            w = cap // 4
            most = min(m * spp_pass, -(-(-(-m * spp_pass // 64)) // 4) * 64)  # primaries of the wave with the most chunks of 64
            assert w & (w - 1) == 0 and 128 <= w <= 1024 and (4 * most + 3 <= w or w == (wave_stack or 1024))
            assert b0 == K * cap * 40 and b1 == (K * 4 * 128 * 48 if park else 0)
        else:
            assert cap >= 4 * m * spp_pass + 16 and cap % 256 == 0 and b0 == b1 == K * cap * 40
        return spp_pass, m, K, cap, retries

    npix = 1024 * 768
    spp_pass, m, K, cap, _ = plan(npix, 4096, 512 << 20)
    assert spp_pass == 683 and cap == 4096 and -(-4096 // spp_pass) == 6 and 4096 - 5 * spp_pass > 600  # six equal passes
    assert (m, K) == (22, 35747)  # streams of about 16 Ki primaries (24 pixels), nudged to the fewest rounds of 1280 resident workgroups x pixels
    assert plan(npix, 4096, 512 << 20, groups=4)[1:3] == (24, 32768)  # (rounds of 1024: 32.0)
    spp_pass, m, K, cap, _ = plan(npix, 1024, 512 << 20, park=1, bvh=1)  # mesh.json: two passes of 512, streams of 22 Ki primaries for scenes with walks, no nudge
    assert spp_pass == 512 and m == -(-npix // -(-npix * 512 // 22528)) and cap == 4096
    spp_pass, m, K, cap, _ = plan(npix, 128, 512 << 20)  # few samples: one pass, short streams of at most 64 (+ nudge) pixels
    assert spp_pass == 128 and m <= 72 and K >= 10922
    spp_pass, m, K, cap, retries = plan(128 * 96, 64, 512 << 20, budget=8 << 20)  # the budget test of the GPU suite
    assert spp_pass == 1 and retries == 6 and cap == 512
    spp_pass, m, K, cap, _ = plan(npix, 4096, 512 << 20, wave_stack=512)
    assert cap == 2048
    spp_pass, m, K, cap, _ = plan(48 * 32, 8, 512 << 20)  # a tiny frame: 2048 streams of one pixel, 8 + 8 <= 128 slots
    assert (spp_pass, m, cap) == (8, 1, 512)
    spp_pass, m, K, cap, _ = plan(npix, 4096, 96 << 20, stack=0)  # level by level: 128 samples per pass, 4 slots per primary
    assert spp_pass == 128 and cap >= 4 * m * 128
    assert plan(1 << 20, 32767, 1 << 31, default=0, stack=0) == "TOOLARGE"
    spp_pass, m, K, cap, retries = plan(1 << 20, 32767, 1 << 31, default=1, stack=0)  # the same as a default: halved until it fits
    assert retries >= 1 and K * cap <= 0x7fffffff


PASS_SRC = r"""
#include <cstdio>
#include <cstdint>
#include "pt_host.h"
using namespace pt;
int main() {
    const uint64_t npix = 1024u * 768u, probe = 1u << 20;
    // nothing measured yet: the probe (1 Mi primary rays = one sample per pixel of this frame)
    if (host::next_pass_samples(0.0, 100.0, npix, probe, 0, 4096, 683) != 1) { printf("FAIL probe\n"); return 1; }
    // a tiny frame's probe is never less than one sample
    if (host::next_pass_samples(0.0, 100.0, 4096u * 4096u, probe, 0, 100, 32) != 1) { printf("FAIL probe floor\n"); return 1; }
    // a measured rate is followed, but a pass grows at most sixteen-fold (short passes measure overheads), plus the fifth by which a
    // pass may be stretched: 19 after 1, 292 after 16
    const double rate = 5.5e6;  // primary samples per ms: cornell.json on an MI355X
    if (host::next_pass_samples(rate, 100.0, npix, probe, 1, 4095, 683) != 19) { printf("FAIL growth\n"); return 1; }
    if (host::next_pass_samples(rate, 100.0, npix, probe, 16, 4079, 683) != 292) { printf("FAIL growth 2 %u\n", host::next_pass_samples(rate, 100.0, npix, probe, 16, 4079, 683)); return 1; }
    // steady state: 699 samples would fit 100 ms, the plan allows 683: the bench frame is six equal passes of 683
    if (host::next_pass_samples(rate, 100.0, npix, probe, 683, 4096, 683) != 683) { printf("FAIL steady\n"); return 1; }
    // equal passes over what is left, each at most a fifth longer than the target rather than one pass more:
    // mesh.json's rate fits 448 samples into 100 ms; 1024 left -> two passes of 512 (114 ms), not three of 342
    if (host::next_pass_samples(3.52e6, 100.0, npix, probe, 512, 1024, 683) != 512) { printf("FAIL stretch\n"); return 1; }
    if (host::next_pass_samples(3.52e6, 100.0, npix, probe, 512, 1100, 683) != 367) { printf("FAIL equal %u\n", host::next_pass_samples(3.52e6, 100.0, npix, probe, 512, 1100, 683)); return 1; }
    // a scene fifty times dearer: passes of a tenth of a second are a handful of samples; never zero, never beyond what is left
    if (host::next_pass_samples(rate / 50.0, 100.0, npix, probe, 8, 10000, 683) != 15) { printf("FAIL dear %u\n", host::next_pass_samples(rate / 50.0, 100.0, npix, probe, 8, 10000, 683)); return 1; }
    if (host::next_pass_samples(1.0, 100.0, npix, probe, 1, 3, 683) != 1) { printf("FAIL floor\n"); return 1; }
    if (host::next_pass_samples(rate, 100.0, npix, probe, 683, 5, 683) != 5) { printf("FAIL left\n"); return 1; }
    printf("OK\n");
    return 0;
}
"""


def test_pass_length_follows_the_measured_rate(tmp_path):
    """host::next_pass_samples (render_wavefront / render_mega): the probe, the sixteen-fold growth limit, the plan's cap, equal
    passes stretched by at most a fifth, floors - the arithmetic behind "a cancel comes back within a tenth of a second"."""
    src = tmp_path / "p.cpp"
    src.write_text(PASS_SRC)
    exe = str(tmp_path / "p")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ptlib.PKG, "csrc"), "-I", os.path.join(ptlib.ROOT, "include"),
                           str(src), "-o", exe, "-L", ptlib.PKG, "-lptrace_hip", "-Wl,-rpath," + ptlib.PKG])
    assert subprocess.check_output([exe]).decode().strip() == "OK"
