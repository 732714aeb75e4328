// A driver with the shape of the reference's src/main.cpp:24-73, written against include/rtx_reference_api.hpp:
// Scene::teapot(), render(scene, xMin, xMax) over eight column strips, three timed repeats printed in seconds with
// three decimals, saveCanvas(), string literals thrown on failure and turned into exit code -1.  Run from a
// directory that holds scenes/teapot.obj and out/, as the reference is.  `--gpus N` (N = 0: all) spreads every
// frame over N GPUs of the node: the place of the reference's eight threads (src/main.cpp:34-52) is taken by
// screen tiles dealt to the GPUs and one RCCL gather, all inside the library.
#define CANVAS_WIDTH 1920
#define CANVAS_HEIGHT 1920
#include "rtx_reference_api.hpp"

#include <chrono>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <thread>

int main(int argc, char **argv) {
    const bool one_call = argc > 1 && std::string(argv[1]) == "--whole-frame";   // one render() for the whole width
    const bool threaded = argc > 1 && std::string(argv[1]) == "--threads";       // the strips from eight threads at once, as the reference starts them
    for (int a = 1; a + 1 < argc; a++)
        if (std::string(argv[a]) == "--gpus") rtx_reference_api_detail::gpus() = std::atoi(argv[a + 1]);
    Scene scene;
    try {
        scene = Scene::teapot();
        const int strips = one_call ? 1 : 8, width = CANVAS_WIDTH / strips;
        std::cout << std::fixed << std::setprecision(3);
        for (int repeat = 0; repeat < 3; repeat++) {
            const auto begin = std::chrono::high_resolution_clock::now();
            if (threaded) {
                std::vector<std::thread> pool;
                std::vector<const char *> failed(strips, nullptr);
                for (int k = 0, x = -CANVAS_WIDTH / 2; k < strips; k++, x += width)
                    pool.emplace_back([&failed, scene, k, x, width]() {
                        try { render(scene, x, x + width); } catch (const char *message) { failed[k] = message; }
                    });
                for (std::thread &t : pool) t.join();
                for (const char *message : failed)
                    if (message) throw "a render thread failed\n";
            } else {
                for (int k = 0, x = -CANVAS_WIDTH / 2; k < strips; k++, x += width) render(scene, x, x + width);
            }
            const auto spent = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - begin);
            std::cout << static_cast<double>(spent.count()) / 1000 << " ";
        }
        std::cout << "\n";
        saveCanvas();
    } catch (const char *message) {
        std::cerr << message;
        return -1;
    }
    return 0;
}
