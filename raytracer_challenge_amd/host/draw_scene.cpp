// Render a scene file through the HIP library and write a P3 image:
//   draw_scene <scene file> <out.ppm> [hsize vsize [depth]]
// e.g. draw_scene scenes/draw_world.scene result.ppm 3840 2160   (the reference's depth is 5)
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "scene_text.hpp"

using namespace rtc_host;

int main(int argc, char** argv) {
    if (argc < 3) {
        std::cerr << "usage: draw_scene <scene file> <out.ppm> [hsize vsize [depth]]\n";
        return 2;
    }
    const size_t hsize = argc > 4 ? std::strtoul(argv[3], nullptr, 10) : 0;
    const size_t vsize = argc > 4 ? std::strtoul(argv[4], nullptr, 10) : 0;
    const int depth = argc > 5 ? std::atoi(argv[5]) : 5;
    try {
        std::ifstream in(argv[1]);
        if (!in) throw SceneTextError(std::string("cannot open ") + argv[1]);
        const Scene scene = parseScene(in, hsize, vsize);
        RtcStats stats{};
        // the colours come back as the bytes canvasToPPM (ppm.zig:11-38) would print
        const ByteCanvas canvas = scene.camera.renderBytes(scene.world, depth, &stats);
        std::ofstream file(argv[2]);
        bytesToPPM(canvas, file);
        std::cerr << scene.world.objects.size() << " shapes, " << canvas.width << "x" << canvas.height << ": "
                  << stats.kernel_ms << " ms on the GPU, " << stats.radiance_rays << " radiance rays -> " << argv[2] << "\n";
    } catch (const SceneTextError& e) {
        std::cerr << "scene error: " << e.what() << "\n";
        return 2;
    } catch (const RenderError& e) {
        std::cerr << "render failed (" << e.code << "): " << e.what() << "\n";
        return 1;
    }
    return 0;
}
