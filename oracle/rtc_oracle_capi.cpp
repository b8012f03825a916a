// TEST INFRASTRUCTURE ONLY — C entry points of the CPU oracle (see rtc_oracle.hpp).
// Takes the same plain-old-data descriptors as the product's C-ABI (include/rtc_hip.h)
// so that a test can hand identical inputs to both and compare the outputs.
#include "rtc_oracle.hpp"

#include "../include/rtc_hip.h"

#include <chrono>

namespace {

orc::Mat4 toMat4(const double* m) {
    orc::Mat4 r{};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.mat[i][j] = m[i * 4 + j];
    return r;
}

orc::Shape toShape(const RtcShapeDesc& d) {
    orc::Shape s;
    s.geo = d.geo;
    s.transform = toMat4(d.transform);
    const RtcMaterialDesc& m = d.material;
    s.material.color = {m.color[0], m.color[1], m.color[2]};
    s.material.ambient = m.ambient;
    s.material.diffuse = m.diffuse;
    s.material.specular = m.specular;
    s.material.shininess = m.shininess;
    s.material.reflective = m.reflective;
    s.material.transparency = m.transparency;
    s.material.refractive_index = m.refractive_index;
    s.material.has_pattern = m.has_pattern != 0;
    if (s.material.has_pattern) {
        s.material.pattern.kind = m.pattern.kind;
        s.material.pattern.a = {m.pattern.a[0], m.pattern.a[1], m.pattern.a[2]};
        s.material.pattern.b = {m.pattern.b[0], m.pattern.b[1], m.pattern.b[2]};
        s.material.pattern.transform = toMat4(m.pattern.transform);
    }
    return s;
}

orc::World toWorld(const RtcShapeDesc* shapes, uint64_t n, const RtcLightDesc* light) {
    orc::World w;
    for (uint64_t i = 0; i < n; ++i) w.objects.push_back(toShape(shapes[i]));
    w.light.position = {light->position[0], light->position[1], light->position[2], light->position[3]};
    w.light.intensity = {light->intensity[0], light->intensity[1], light->intensity[2]};
    return w;
}

orc::Camera toCamera(const RtcCameraDesc* c) {
    orc::Camera cam;
    cam.hsize = c->hsize;
    cam.vsize = c->vsize;
    cam.field_of_view = c->field_of_view;
    cam.half_width = c->half_width;
    cam.half_height = c->half_height;
    cam.pixel_size = c->pixel_size;
    cam.transform = toMat4(c->transform);
    return cam;
}

void fillStats(const orc::Stats& s, RtcStats* out, uint64_t pixels, double ms) {
    if (!out) return;
    out->radiance_rays = s.radiance_rays;
    out->shadow_rays = s.shadow_rays;
    out->sphere_tests = s.sphere_tests;
    out->plane_tests = s.plane_tests;
    out->sphere_hits = s.sphere_hits;
    out->plane_hits = s.plane_hits;
    out->shaded_hits = s.shaded_hits;
    out->pixels = pixels;
    out->kernel_ms = ms;
    out->sphere_bound_skips = 0;  // the reference evaluates every pair
    out->plane_side_skips = 0;
    out->cube_tests = s.cube_tests;
    out->cube_hits = s.cube_hits;
    out->sphere_prefilter_skips = 0;
}

}  // namespace

extern "C" {

// Render rows row_begin, row_begin+row_step, ... < row_end of the canvas.  mode 0 =
// LITERAL, 1 = STREAMING.  out_rgb is the full hsize*vsize*3 canvas; unvisited rows are
// left untouched.  stats->kernel_ms receives the wall time of the render loop.
int orc_render(const RtcShapeDesc* shapes, uint64_t n_shapes, const RtcLightDesc* light,
               const RtcCameraDesc* camera, int32_t depth, int32_t mode, uint64_t row_begin, uint64_t row_end,
               uint64_t row_step, double* out_rgb, RtcStats* stats) {
    orc::World w = toWorld(shapes, n_shapes, light);
    const orc::Camera cam = toCamera(camera);
    if (row_end > cam.vsize) row_end = cam.vsize;
    if (row_step == 0) row_step = 1;
    const auto t0 = std::chrono::steady_clock::now();
    orc::render(cam, w, depth, mode == 0 ? orc::LITERAL : orc::STREAMING, out_rgb, row_begin, row_end, row_step);
    const auto t1 = std::chrono::steady_clock::now();
    uint64_t rows = row_end > row_begin ? (row_end - row_begin + row_step - 1) / row_step : 0;
    fillStats(w.stats, stats, rows * cam.hsize, std::chrono::duration<double, std::milli>(t1 - t0).count());
    return 0;
}

// Render an explicit list of pixels (canvas indices y*hsize + x); out_rgb receives
// n_pixels*3 doubles in list order.
int orc_render_pixels(const RtcShapeDesc* shapes, uint64_t n_shapes, const RtcLightDesc* light,
                      const RtcCameraDesc* camera, int32_t depth, int32_t mode, const uint64_t* pixels,
                      uint64_t n_pixels, double* out_rgb, RtcStats* stats) {
    orc::World w = toWorld(shapes, n_shapes, light);
    const orc::Camera cam = toCamera(camera);
    const orc::Mode m = mode == 0 ? orc::LITERAL : orc::STREAMING;
    if (m == orc::STREAMING) w.prepare();
    for (uint64_t i = 0; i < n_pixels; ++i) {
        const uint64_t x = pixels[i] % cam.hsize, y = pixels[i] / cam.hsize;
        const orc::Color c = orc::worldColorAt(w, cam.rayForPixel(x, y), depth, m);
        out_rgb[i * 3 + 0] = c.red;
        out_rgb[i * 3 + 1] = c.green;
        out_rgb[i * 3 + 2] = c.blue;
    }
    fillStats(w.stats, stats, n_pixels, 0.0);
    return 0;
}

// calc.worldColorAt for caller-supplied rays (6 doubles each: origin xyz, direction xyz).
int orc_color_at(const RtcShapeDesc* shapes, uint64_t n_shapes, const RtcLightDesc* light, const double* rays,
                 uint64_t n_rays, int32_t depth, int32_t mode, double* out_rgb, RtcStats* stats) {
    orc::World w = toWorld(shapes, n_shapes, light);
    const orc::Mode m = mode == 0 ? orc::LITERAL : orc::STREAMING;
    if (m == orc::STREAMING) w.prepare();
    for (uint64_t i = 0; i < n_rays; ++i) {
        const double* r = rays + i * 6;
        const orc::Ray ray{orc::initPoint(r[0], r[1], r[2]), orc::initVector(r[3], r[4], r[5])};
        const orc::Color c = orc::worldColorAt(w, ray, depth, m);
        out_rgb[i * 3 + 0] = c.red;
        out_rgb[i * 3 + 1] = c.green;
        out_rgb[i * 3 + 2] = c.blue;
    }
    fillStats(w.stats, stats, n_rays, 0.0);
    return 0;
}

// Mat4.inverse (matrix.zig:286-300) and the determinant, row-major 16 doubles.
int orc_mat4_inverse(const double* m, double* out_inv, double* out_det) {
    const orc::Mat4 a = toMat4(m);
    if (out_det) *out_det = a.determinant();
    const orc::Mat4 inv = a.inverse();
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out_inv[i * 4 + j] = inv.mat[i][j];
    return 0;
}

// Camera.rayForPixel (camera.zig:48-72): out receives origin xyz, direction xyz.
int orc_ray_for_pixel(const RtcCameraDesc* camera, uint64_t x, uint64_t y, double* out6) {
    const orc::Ray r = toCamera(camera).rayForPixel(x, y);
    out6[0] = r.origin.x;
    out6[1] = r.origin.y;
    out6[2] = r.origin.z;
    out6[3] = r.direction.x;
    out6[4] = r.direction.y;
    out6[5] = r.direction.z;
    return 0;
}

// The loop of draw_sphere.zig:38-83 on a canvas_size x canvas_size canvas; out_rays may be NULL.
int orc_draw_sphere(uint64_t canvas_size, double* out_rgb, double* out_rays) {
    orc::drawSphere(canvas_size, out_rgb, out_rays);
    return 0;
}

// std.math.pow as restated in rtc_oracle.hpp.
double orc_pow(double x, double y) { return orc::zigmath::pow(x, y); }

// byteColorValue (ppm.zig:6-9) over an array.
int orc_quantize_u8(const double* rgb, uint64_t n, uint8_t* out) {
    for (uint64_t i = 0; i < n; ++i) out[i] = orc::byteColorValue(rgb[i]);
    return 0;
}

}  // extern "C"
