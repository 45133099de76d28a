// ptrace_cli.cpp — command-line host above the C ABI with the UX of the reference's (dead) CLI,
// src/cmd_render.rs:16-44 / .vscode/launch.json: `ptrace <spp> <res_y> <scene id or index>`.
// Width defaults to res_y*3/2 (src/render/mod.rs:872-879, src/main.rs:174-177); the scene is looked up as
// scenes/{id}.json (mod.rs:94) or by index into the sorted scenes/*.json listing (scenes.rs:10-41); the
// image goes to out/<timestamp>-scene-<id>-spp<N>-res<H>-.ppm plus a latest.ppm symlink (mod.rs:1031-1088).
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "../../include/ptrace.h"

static void usage() {
    fprintf(stderr,
            "usage: ptrace <samplesPerPixel> <y-resolution> <scene id|index> [--width W] [--backend wavefront|megakernel]\n"
            "              [--seed S] [--gpus N] [--root DIR] [--out DIR] [--no-ppm]\n");
}

static std::vector<std::string> scene_ids(const std::string &root) {
    std::vector<std::string> ids;
    DIR *d = opendir((root + "/scenes").c_str());
    if (!d) return ids;
    while (dirent *e = readdir(d)) {
        std::string n = e->d_name;
        if (n.size() > 5 && n.substr(n.size() - 5) == ".json") ids.push_back(n.substr(0, n.size() - 5));
    }
    closedir(d);
    std::sort(ids.begin(), ids.end());
    return ids;
}

static void progress(void *, float f) {
    fprintf(stderr, "\rRendering ... %5.1f%%", 100.0 * f);
    fflush(stderr);
}

int main(int argc, char **argv) {
    if (argc < 4) {
        usage();
        return 1;
    }
    const uint32_t spp = (uint32_t)strtoul(argv[1], nullptr, 10);
    const uint32_t res_y = (uint32_t)strtoul(argv[2], nullptr, 10);
    std::string scene_arg = argv[3], root = ".", out_dir = "out", backend = "wavefront";
    uint32_t width = res_y * 3 / 2;
    uint64_t seed = (uint64_t)time(nullptr);
    bool write_ppm = true;
    uint32_t gpus = 1;
    for (int i = 4; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--width") width = (uint32_t)strtoul(next(), nullptr, 10);
        else if (a == "--backend") backend = next();
        else if (a == "--seed") seed = strtoull(next(), nullptr, 10);
        else if (a == "--gpus") gpus = (uint32_t)strtoul(next(), nullptr, 10);
        else if (a == "--root") root = next();
        else if (a == "--out") out_dir = next();
        else if (a == "--no-ppm") write_ppm = false;
        else {
            usage();
            return 1;
        }
    }
    if (!spp || !res_y || !width) {
        usage();
        return 1;
    }
    // load_scene_ids (scenes.rs:28-38): a scenes/ directory without any *.json is filled with the built-in scenes
    if (scene_ids(root).empty()) {
        mkdir((root + "/scenes").c_str(), 0777);
        for (uint32_t i = 0; i < pt_builtin_scene_count(); ++i) {
            const char *bid = pt_builtin_scene_id(i);
            pt_scene *b = nullptr;
            if (pt_scene_builtin(bid, root.c_str(), &b) != PT_OK ||
                pt_scene_save(b, (root + "/scenes/" + bid + ".json").c_str()) != PT_OK)
                fprintf(stderr, "Failed to save scene '%s': %s\n", bid, pt_last_error());
            pt_scene_free(b);
        }
    }
    // SceneId::Int(i) -> nth scene of the listing, SceneId::String -> by id (cmd_render.rs:19-30)
    std::string id = scene_arg;
    char *endp = nullptr;
    unsigned long idx = strtoul(scene_arg.c_str(), &endp, 10);
    if (*endp == '\0' && !scene_arg.empty()) {
        std::vector<std::string> ids = scene_ids(root);
        if (idx >= ids.size()) {
            fprintf(stderr, "scene index %lu out of range (%zu scenes)\n", idx, ids.size());
            return 1;
        }
        id = ids[idx];
    }
    pt_scene *sc = nullptr;
    int rc = pt_scene_load((root + "/scenes/" + id + ".json").c_str(), root.c_str(), &sc);
    if (rc) {
        fprintf(stderr, "cannot load scene '%s': %s\n", id.c_str(), pt_last_error());
        return 1;
    }
    uint32_t n_objs = 0, n_tris = 0;
    const pt_object *objs = pt_scene_objects(sc, &n_objs);
    const pt_triangle *tris = pt_scene_triangles(sc, &n_tris);
    printf("Rendering scene %s (%u objects), %u samples per pixel, %ux%u resolution\n", pt_scene_id(sc), n_objs, spp,
           width, res_y);  // mod.rs:987-995
    pt_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.width = width;
    cfg.height = res_y;
    cfg.spp = spp;
    cfg.backend = backend == "megakernel" ? PT_BACKEND_MEGAKERNEL : PT_BACKEND_WAVEFRONT;
    cfg.seed = seed;
    std::vector<float> img((size_t)width * res_y * 3, 0.0f);
    pt_stats st;
    rc = pt_render_multi(&cfg, gpus ? gpus : 1, pt_scene_camera(sc), objs, n_objs, tris, n_tris, img.data(), nullptr,
                         progress, nullptr, &st);
    fprintf(stderr, "\n");
    if (rc) {
        fprintf(stderr, "render failed (%d): %s\n", rc, pt_last_error());
        pt_scene_free(sc);
        return 2;
    }
    printf("Rendering complete\n");
    printf("{\"ray_bounces\": %llu, \"samples\": %llu, \"ms_total\": %.3f, \"ms_device\": %.3f, \"ray_bounces_per_sec\": %.4g}\n",
           (unsigned long long)st.ray_bounces, (unsigned long long)st.samples, st.ms_total, st.ms_device,
           st.ray_bounces / (st.ms_total * 1e-3));
    if (write_ppm) {
        mkdir(out_dir.c_str(), 0755);  // create_dir_all("out"), mod.rs:1032
        char stamp[64];
        time_t now = time(nullptr);
        strftime(stamp, sizeof stamp, "%Y-%m-%d_%H:%M:%S", localtime(&now));
        const std::string path = out_dir + "/" + stamp + "-scene-" + pt_scene_id(sc) + "-spp" + std::to_string(spp) +
                                 "-res" + std::to_string(res_y) + "-.ppm";  // mod.rs:1035-1041
        rc = pt_write_ppm(path.c_str(), img.data(), width, res_y, spp, pt_scene_id(sc), (uint64_t)(st.ms_total / 1000.0));
        if (rc) {
            fprintf(stderr, "cannot write %s: %s\n", path.c_str(), pt_last_error());
            pt_scene_free(sc);
            return 3;
        }
        unlink("latest.ppm");  // mod.rs:1079-1088
        if (symlink(path.c_str(), "latest.ppm") != 0)
            printf("Could not create symlink to latest image. You can find it at %s\n", path.c_str());
        printf("wrote %s\n", path.c_str());
    }
    pt_scene_free(sc);
    return 0;
}
