/* Source-level mirror of the reference's driver interface, for a program shaped like the reference's src/main.cpp.
 *
 * The reference's driver consists of (src/main.cpp:8-9,16-22,75-116; include/scene.hpp:66-76):
 *     #define CANVAS_WIDTH / CANVAS_HEIGHT           the canvas size, fixed at compile time
 *     Scene::teapot() ... Scene::aurelius()           factories that load scenes/<name>.obj and throw a string literal
 *     void render(Scene scene, int xMin, int xMax)    one column strip x in [xMin, xMax), x = 0 in the middle
 *     void saveCanvas()                               canvas -> CANVAS_PATH as P6, throwing "Failed to open file.\n"
 * The same names with the same argument meaning and the same error behaviour are defined here on top of the C ABI
 * of rtx.h: a translation unit defines CANVAS_WIDTH, CANVAS_HEIGHT (and optionally CANVAS_PATH) and includes this
 * header once.  render() runs the strip on the GPU; rendering the whole width in one call is the intended use, the
 * eight strips of the reference's thread fan-out work too (each is its own launch).
 *
 * Header-only C++17; the C ABI stays the boundary, this file only spells it the reference's way. */
#ifndef RTX_REFERENCE_API_HPP
#define RTX_REFERENCE_API_HPP

#ifndef CANVAS_WIDTH
#error "define CANVAS_WIDTH and CANVAS_HEIGHT before including rtx_reference_api.hpp (src/main.cpp:8-9)"
#endif
#ifndef CANVAS_PATH
#define CANVAS_PATH "out/canvas.ppm"      /* src/main.cpp:7 */
#endif

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include <unistd.h>

#include "rtx.h"

namespace rtx_reference_api_detail {
struct Handles {
    rtx_host_scene *host = nullptr;
    rtx_scene *device = nullptr;
    rtx_scene_group *group = nullptr;      /* more than one GPU: replicas on all of them instead of `device` */
    ~Handles() {
        rtx_scene_group_destroy(group);
        rtx_scene_destroy(device);
        rtx_host_scene_free(host);
    }
};
/* GPUs a Scene loaded from now on is spread over (the reference's threadsSize, src/main.cpp:35, counts host
 * threads); 1 = one GPU and no RCCL, 0 = every GPU of the node. */
inline int &gpus() {
    static int n = 1;
    return n;
}
}  // namespace rtx_reference_api_detail

/* include/scene.hpp:15-76: here a handle to the scene on the host and in HBM; copies share it (the reference
 * copies its Scene into every thread, src/main.cpp:46). */
struct Scene {
    std::shared_ptr<rtx_reference_api_detail::Handles> handles;

    static Scene load(const char *name) {
        auto h = std::make_shared<rtx_reference_api_detail::Handles>();
        const std::string path = std::string("scenes/") + name + ".obj";                       /* scene.hpp:262 */
        /* verbose: the "[ TRI ]" line of parseOBJ (scene.hpp:248); no tree on the host, it is built on the GPU */
        if (rtx_host_scene_load_obj_ex(path.c_str(), name, RTX_LOAD_VERBOSE | RTX_LOAD_NO_BVH, -1, &h->host) != RTX_OK)
            throw rtx_last_error();                                                            /* scene.hpp:216-218 */
        const int n_gpus = rtx_reference_api_detail::gpus();
        if (n_gpus == 1) {
            if (rtx_scene_create_from_host(h->host, -1, &h->device) != RTX_OK) throw rtx_last_error();
        } else {
            /* RCCL may announce its version on standard output while its communicators are created.  This driver's
             * standard output carries the reference's lines and nothing else, so for that one call -- made before any
             * other thread of the driver exists -- descriptor 1 points at standard error. */
            fflush(stdout);
            const int saved = dup(1);
            const bool redirected = saved >= 0 && dup2(2, 1) >= 0;
            const int rc = rtx_scene_group_create_from_host(h->host, nullptr, n_gpus, &h->group);
            fflush(stdout);
            if (redirected && dup2(saved, 1) < 0) throw "cannot restore standard output\n";
            if (saved >= 0) close(saved);
            if (rc != RTX_OK) throw rtx_last_error();
        }
        const double s = rtx_scene_build_seconds(h->group ? rtx_scene_group_scene(h->group, 0) : h->device);
        printf("[ BVH ] %g seconds.\n", static_cast<double>(static_cast<long long>(s * 1000.0)) / 1000.0);   /* scene.hpp:49-51 */
        Scene scene;
        scene.handles = h;
        return scene;
    }
    static Scene teapot() { return load("teapot"); }        /* scene.hpp:255-293 */
    static Scene bunny() { return load("bunny"); }          /* scene.hpp:295-332 */
    static Scene erato() { return load("erato"); }          /* scene.hpp:334-371 */
    static Scene dragon() { return load("dragon"); }        /* scene.hpp:373-410 */
    static Scene aurelius() { return load("aurelius"); }    /* scene.hpp:412-448 */
};

/* src/main.cpp:16: the canvas, here already clamped bytes (saveCanvas clamps, src/main.cpp:107) */
inline std::vector<uint8_t> &canvas() {
    static std::vector<uint8_t> pixels(static_cast<size_t>(CANVAS_WIDTH) * CANVAS_HEIGHT * 3);
    return pixels;
}

/* src/main.cpp:75-88: columns x in [xMin, xMax) of the frame, x = -CANVAS_WIDTH/2 being the left edge; depth 3. */
inline void render(Scene scene, int xMin, int xMax) {
    rtx_render_params p;
    rtx_render_params_default(&p, CANVAS_WIDTH, CANVAS_HEIGHT);
    p.traversal = RTX_TRAVERSAL_PRUNED;
    p.x0 = xMin + CANVAS_WIDTH / 2;                     /* putPixel: column = x + W/2 (src/main.cpp:90-94) */
    p.x1 = xMax + CANVAS_WIDTH / 2;
    p.y0 = 0;
    p.y1 = CANVAS_HEIGHT;
    auto frame_into = [&](uint8_t *rgb) {                /* on one GPU, or tiles over all GPUs of the group + RCCL gather */
        const int rc = scene.handles->group ? rtx_render_sharded(scene.handles->group, &p, rgb)
                                            : rtx_render(scene.handles->device, &p, rgb, nullptr, nullptr);
        if (rc != RTX_OK) throw rtx_last_error();
    };
    if (p.x0 == 0 && p.x1 == CANVAS_WIDTH) {            /* the whole frame: straight into the canvas */
        frame_into(canvas().data());
        return;
    }
    std::vector<uint8_t> frame(canvas().size());         /* a strip: rendered into a frame of its own, its columns kept */
    frame_into(frame.data());
    for (int row = 0; row < CANVAS_HEIGHT; row++) {
        const size_t at = (static_cast<size_t>(row) * CANVAS_WIDTH + static_cast<size_t>(p.x0)) * 3;
        std::copy(frame.begin() + at, frame.begin() + at + static_cast<size_t>(p.x1 - p.x0) * 3, canvas().begin() + at);
    }
}

/* src/main.cpp:96-116 */
inline void saveCanvas() {
    if (rtx_save_ppm(CANVAS_PATH, CANVAS_WIDTH, CANVAS_HEIGHT, canvas().data()) != RTX_OK) throw rtx_last_error();
}

#endif /* RTX_REFERENCE_API_HPP */
