// The reference's production program (draw_world.zig:23-202) against the HIP library: builds
// the Ch. 11 reflection/refraction room, renders it with camera.render(world) and writes
// result.ppm.  usage: draw_world [hsize vsize [out.ppm]]   (the reference uses 900 450 result.ppm)
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "rtc_host.hpp"

using namespace rtc_host;

int main(int argc, char** argv) {
    const size_t hsize = argc > 2 ? std::strtoul(argv[1], nullptr, 10) : 900;
    const size_t vsize = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 450;
    const char* out_path = argc > 3 ? argv[3] : "result.ppm";

    World world;
    Camera camera = Camera::init(hsize, vsize, 1.152);
    const Vec4 from = initVector(-2.6, 1.5, -3.9);
    const Vec4 to = initVector(-0.6, 1, -0.8);
    const Vec4 up = initVector(0, 1, 0);
    camera.transform = viewTransform(from, to, up);

    world.light = PointLight{initPoint(-4.9, 4.9, -1), White};

    Pattern wall_pattern;
    wall_pattern.kind = Pattern::stripe;
    wall_pattern.a = Color::init(0.45, 0.45, 0.45);
    wall_pattern.b = Color::init(0.55, 0.55, 0.55);
    wall_pattern.transform = Mat4::identity().scale(0.25, 0.25, 0.25).rotateY(1.5708);
    Material wall_material;
    wall_material.pattern = wall_pattern;
    wall_material.ambient = 0;
    wall_material.diffuse = 0.4;
    wall_material.specular = 0;
    wall_material.reflective = 0.3;

    Pattern floor_pattern;
    floor_pattern.kind = Pattern::checkers;
    floor_pattern.a = Color::init(0.35, 0.35, 0.35);
    floor_pattern.b = Color::init(0.65, 0.65, 0.65);
    floor_pattern.transform = Mat4::identity().rotateY(0.31415);
    Shape floor;
    floor.geo = Shape::plane;
    floor.material.pattern = floor_pattern;
    floor.material.specular = 0;
    floor.material.reflective = 0.4;
    world.objects.push_back(floor);

    Shape ceiling;
    ceiling.geo = Shape::plane;
    ceiling.material.color = Color::init(0.8, 0.8, 0.8);
    ceiling.material.ambient = 0.3;
    ceiling.material.specular = 0;
    ceiling.transform = Mat4::identity().translate(0, 5, 0);
    world.objects.push_back(ceiling);

    auto wall = [&](const Mat4& t) {
        Shape s;
        s.geo = Shape::plane;
        s.material = wall_material;
        s.transform = t;
        world.objects.push_back(s);
    };
    wall(Mat4::identity().rotateY(1.5708).rotateZ(1.5708).translate(-5, 0, 0));  // west
    wall(Mat4::identity().rotateY(1.5708).rotateZ(1.5708).translate(5, 0, 0));   // east
    wall(Mat4::identity().rotateX(1.5708).translate(0, 0, 5));                   // north
    wall(Mat4::identity().rotateX(1.5708).translate(0, 0, -5));                  // south

    auto ball = [&](double s, double x, double y, double z, const Color& c) {
        Shape b;
        b.transform = Mat4::identity().scale(s, s, s).translate(x, y, z);
        b.material.color = c;
        b.material.shininess = 50;
        world.objects.push_back(b);
    };
    ball(0.4, 4.6, 0.4, 1.0, Color::init(0.8, 0.5, 0.3));
    ball(0.3, 4.7, 0.3, 0.4, Color::init(0.9, 0.4, 0.5));
    ball(0.5, -1, 0.5, 4.5, Color::init(0.4, 0.9, 0.6));
    ball(0.3, -1.7, 0.3, 4.7, Color::init(0.4, 0.6, 0.9));

    Shape red;
    red.transform = Mat4::identity().translate(-0.6, 1, 0.6);
    red.material.color = Color::init(1.0, 0.3, 0.2);
    red.material.specular = 0.4;
    red.material.shininess = 5;
    world.objects.push_back(red);

    auto glass = [&](double s, double x, double y, double z, const Color& c) {
        Shape g;
        g.transform = Mat4::identity().scale(s, s, s).translate(x, y, z);
        g.material.color = c;
        g.material.ambient = 0;
        g.material.diffuse = 0.4;
        g.material.specular = 0.9;
        g.material.shininess = 300;
        g.material.reflective = 0.9;
        g.material.transparency = 0.9;
        g.material.refractive_index = 1.5;
        world.objects.push_back(g);
    };
    glass(0.7, 0.6, 0.7, -0.6, Color::init(0, 0, 0.2));
    glass(0.5, -0.7, 0.5, -0.8, Color::init(0, 0.2, 0));

    try {
        RtcStats stats{};
        const Canvas canvas = camera.render(world, 5, &stats);
        std::ofstream file(out_path);
        canvasToPPM(canvas, file);
        std::cerr << "rendered " << hsize << "x" << vsize << " in " << stats.kernel_ms << " ms on the GPU: "
                  << stats.radiance_rays << " radiance rays, " << stats.shadow_rays << " shadow rays -> " << out_path
                  << "\n";
    } catch (const RenderError& e) {
        std::cerr << "render failed (" << e.code << "): " << e.what() << "\n";
        return 1;
    }
    return 0;
}
