"""Host-side pieces of the device layout that can be checked without a GPU: the round-robin deal of pixels to ray
streams (pt_device.h: stream_pixel / stream_pixel_count / global_pixel) and the bookkeeping word."""
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
