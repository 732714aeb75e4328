// Records rtx_render_device_noalloc() into a HIP graph and looks at what was recorded.  The graph is never launched.
//
//   graph_capture_probe <teapot.obj>
//
// Step 1 renders the frame directly.  Step 2 captures the same call on a capturing stream (RTX_ALLOW_CAPTURE=1 lifts the
// library's refusal), ends the capture and prints every node: its type and, for kernel nodes, the launch geometry, the
// dynamic LDS size and the recorded bytes of the two argument blocks beside the fields they must hold (canvas size,
// recursion depth, output pointer).
//
// build: hipcc -std=c++17 -O2 -I include -I ray_tracer_amd/csrc tools/graph_capture_probe.cpp -o ray_tracer_amd/graph_capture_probe \
//        -L ray_tracer_amd -lrtx -Wl,-rpath,'$ORIGIN'
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "device_layout.h"
#include "rtx.h"

#define CHECK(expr)                                                                            \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_)); return 2; } \
    } while (0)
#define RTX(expr)                                                                              \
    do {                                                                                       \
        if ((expr) != RTX_OK) { fprintf(stderr, "%s: %s\n", #expr, rtx_last_error()); return 2; } \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: graph_capture_probe <teapot.obj>\n"); return 1; }
    const int W = 640, H = 400;
    rtx_host_scene *hs = nullptr;
    RTX(rtx_host_scene_load_obj_ex(argv[1], "teapot", RTX_LOAD_NO_BVH, -1, &hs));
    rtx_scene *scene = nullptr;
    RTX(rtx_scene_create_from_host(hs, -1, &scene));
    rtx_render_params p;
    rtx_render_params_default(&p, W, H);
    uint8_t *d_rgb = nullptr;
    CHECK(hipMalloc(&d_rgb, (size_t)W * H * 3));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));

    // 1. the direct launch
    CHECK(hipMemsetAsync(d_rgb, 0, (size_t)W * H * 3, st));
    RTX(rtx_render_device_noalloc(scene, &p, d_rgb, nullptr, nullptr, nullptr, nullptr, st));
    CHECK(hipStreamSynchronize(st));
    std::vector<uint8_t> want((size_t)W * H * 3);
    CHECK(hipMemcpy(want.data(), d_rgb, want.size(), hipMemcpyDeviceToHost));
    size_t lit = 0;
    for (uint8_t b : want) lit += b != 0;
    printf("direct launch: %zu of %zu bytes non-zero\n", lit, want.size());

    // 2. capture only
    setenv("RTX_ALLOW_CAPTURE", "1", 1);
    hipGraph_t graph = nullptr;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = rtx_render_device_noalloc(scene, &p, d_rgb, nullptr, nullptr, nullptr, nullptr, st);
    const hipError_t end = hipStreamEndCapture(st, &graph);
    if (rc != RTX_OK) { fprintf(stderr, "call under capture: %s\n", rtx_last_error()); return 2; }
    CHECK(end);
    size_t n = 0;
    CHECK(hipGraphGetNodes(graph, nullptr, &n));
    std::vector<hipGraphNode_t> nodes(n);
    CHECK(hipGraphGetNodes(graph, nodes.data(), &n));
    printf("captured graph: %zu nodes\n", n);
    for (size_t i = 0; i < n; i++) {
        hipGraphNodeType type;
        CHECK(hipGraphNodeGetType(nodes[i], &type));
        size_t n_deps = 0;
        CHECK(hipGraphNodeGetDependencies(nodes[i], nullptr, &n_deps));
        printf(" node %zu: type %d, %zu dependencies\n", i, (int)type, n_deps);
        if (type != hipGraphNodeTypeKernel) continue;
        hipKernelNodeParams kp;
        memset(&kp, 0, sizeof kp);
        CHECK(hipGraphKernelNodeGetParams(nodes[i], &kp));
        printf("   kernel: grid %u x %u x %u, block %u x %u x %u, dynamic LDS %u bytes, kernelParams %p, extra %p\n", kp.gridDim.x,
               kp.gridDim.y, kp.gridDim.z, kp.blockDim.x, kp.blockDim.y, kp.blockDim.z, kp.sharedMemBytes, (void *)kp.kernelParams,
               (void *)kp.extra);
        if (kp.blockDim.x == 256 && kp.kernelParams) {
            // the render kernel: (DevScene, DevRender) by value
            const rtx::DevScene *sc = (const rtx::DevScene *)kp.kernelParams[0];
            const rtx::DevRender *r = (const rtx::DevRender *)kp.kernelParams[1];
            printf("   argument 0 (DevScene, %zu bytes): nodes %p tris %p n_tris %u bvh_depth %d n_top_pairs %u\n", sizeof *sc,
                   (const void *)sc->nodes, (const void *)sc->tris, sc->n_tris, sc->bvh_depth, sc->n_top_pairs);
            printf("   argument 1 (DevRender, %zu bytes): W %d H %d depth %d spp %d rgb %p (want %p) work %p tile_order %p total_blocks %u\n",
                   sizeof *r, r->W, r->H, r->depth, r->spp, (void *)r->rgb, (void *)d_rgb, (void *)r->work, (const void *)r->tile_order,
                   r->total_blocks);
        }
    }
    (void)hipGraphDebugDotPrint(graph, "/tmp/rtx_noalloc_graph.dot", hipGraphDebugDotFlagsVerbose);
    printf("capture only: the graph is never launched\n");
    return 0;
}
