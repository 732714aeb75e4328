// Command-line driver: the reference's main() (src/main.cpp:24-73) with the render
// threads replaced by rtx_render().  Same console lines ("[ TRI ] ...", "[ BVH ] ...",
// three frame times in seconds with 3 decimals), same P6 output, same error
// convention (message on stderr, exit code -1).  What the reference fixes at compile
// time (scene, canvas size, recursion depth; src/main.cpp:8-9,28,83) are flags here.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rtx.h"

int main(int argc, char **argv) {
    std::string scene = "teapot", obj, out = "out/canvas.ppm", mode = "pruned", builder = "device";
    bool tolerant = false, overlap = false, knot = false;
    int width = 1920, height = 1920, depth = 3, spp = 1, repeats = 3, device = -1;
    unsigned long long standin = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--scene") scene = val();
        else if (a == "--obj") obj = val();
        else if (a == "--out") out = val();
        else if (a == "--mode") mode = val();
        else if (a == "--width") width = atoi(val());
        else if (a == "--height") height = atoi(val());
        else if (a == "--depth") depth = atoi(val());
        else if (a == "--spp") spp = atoi(val());
        else if (a == "--repeats") repeats = atoi(val());
        else if (a == "--device") device = atoi(val());
        else if (a == "--standin") standin = strtoull(val(), nullptr, 10);
        else if (a == "--builder") builder = val();
        else if (a == "--tolerant") tolerant = true;
        else if (a == "--knot") knot = true;
        else if (a == "--overlap") overlap = true;
        else {
            fprintf(stderr,
                    "usage: %s [--scene teapot|bunny|erato|dragon|aurelius] [--obj FILE | --standin NTRIS [--knot]] [--width W] "
                    "[--height H] [--depth D] [--spp N] [--mode reference|pruned] [--builder device|host] [--tolerant] [--overlap] [--repeats N] "
                    "[--device K] [--out FILE]\n",
                    argv[0]);
            return 2;
        }
    }
    if (obj.empty() && !standin) obj = "scenes/" + scene + ".obj";     // scene.hpp:262

    // With the device builder the host scene carries no tree: it is built, and stays, on the GPU.
    const bool on_device = builder != "host";
    const uint32_t flags = RTX_LOAD_VERBOSE | (on_device ? RTX_LOAD_NO_BVH : 0) | (tolerant ? RTX_LOAD_TOLERANT : 0) | (knot ? RTX_STANDIN_KNOT : 0);
    rtx_host_scene *hs = nullptr;
    int rc = standin ? rtx_host_scene_standin_ex(scene.c_str(), standin, 0, nullptr, flags, device, &hs)
                     : rtx_host_scene_load_obj_ex(obj.c_str(), scene.c_str(), flags, device, &hs);
    if (rc != RTX_OK) { fprintf(stderr, "%s", rtx_last_error()); return -1; }
    rtx_scene *sc = nullptr;
    if (rtx_scene_create_from_host(hs, device, &sc) != RTX_OK) { fprintf(stderr, "%s\n", rtx_last_error()); return -1; }
    if (on_device) {                                                    // Scene::Scene prints this (scene.hpp:49-51)
        const double s = rtx_scene_build_seconds(sc);
        printf("[ BVH ] %g seconds.\n", (double)(long long)(s * 1000.0) / 1000.0);
    }

    rtx_render_params p;
    rtx_render_params_default(&p, width, height);
    p.depth = depth;
    p.spp = spp;
    p.traversal = mode == "pruned" ? RTX_TRAVERSAL_PRUNED : RTX_TRAVERSAL_REFERENCE;
    // the kernel keeps a pixel's column and row in 16 bits each; anything else ends the way the reference's errors do
    if (width < 1 || height < 1 || width > 65535 || height > 65535 || repeats < 0) { fprintf(stderr, "canvas size must be 1..65535 in each dimension\n"); return -1; }
    std::vector<uint8_t> rgb((size_t)width * height * 3);
    auto save = [&](const uint8_t *pixels) {
        const bool png = out.size() > 4 && out.compare(out.size() - 4, 4, ".png") == 0;
        return png ? rtx_save_png(out.c_str(), width, height, pixels) : rtx_save_ppm(out.c_str(), width, height, pixels);
    };
    if (!overlap) {
        for (int n = 0; n < repeats; n++) {
            auto t0 = std::chrono::high_resolution_clock::now();
            if (rtx_render(sc, &p, rgb.data(), nullptr, nullptr) != RTX_OK) { fprintf(stderr, "%s\n", rtx_last_error()); return -1; }
            auto t1 = std::chrono::high_resolution_clock::now();
            long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count();
            printf("%.3f ", (double)ms / 1000);                             // src/main.cpp:40,57
        }
        printf("\n");
        if (save(rgb.data()) != RTX_OK) { fprintf(stderr, "%s", rtx_last_error()); return -1; }
    } else {
        // The repeat loop with the frames overlapped (rtx_render_begin / rtx_render_end): frame n + 1 is queued before
        // frame n is waited for, so a frame's copy to host memory (and the end of its kernel, when lanes run dry)
        // overlaps the next frame's render.  The times printed are the intervals between delivered frames.
        auto last = std::chrono::high_resolution_clock::now();
        if (repeats > 0 && rtx_render_begin(sc, &p, 0, 0) != RTX_OK) { fprintf(stderr, "%s\n", rtx_last_error()); return -1; }
        for (int n = 0; n < repeats; n++) {
            if (n + 1 < repeats && rtx_render_begin(sc, &p, (n + 1) & 1, 0) != RTX_OK) { fprintf(stderr, "%s\n", rtx_last_error()); return -1; }
            if (rtx_render_end(sc, n & 1, rgb.data(), nullptr) != RTX_OK) { fprintf(stderr, "%s\n", rtx_last_error()); return -1; }
            const auto now = std::chrono::high_resolution_clock::now();
            printf("%.4f ", std::chrono::duration<double>(now - last).count());
            last = now;
        }
        printf("\n");
        if (repeats > 0 && save(rgb.data()) != RTX_OK) { fprintf(stderr, "%s", rtx_last_error()); return -1; }
    }
    rtx_scene_destroy(sc);
    rtx_host_scene_free(hs);
    return 0;
}
