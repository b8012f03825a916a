// Host side above the C-ABI, in C++: the reference's scene API with the same names, argument
// meaning and defaults, so that code written against the Zig types reads the same here.
// The reference is Zig (no Zig toolchain in the build image); in a real integration this layer
// stays Zig and only the body of Camera.render changes.
//
//   Vec4 / initPoint / initVector   vector.zig:4-118        Color            color.zig:4-59
//   Mat4 + fluent builders          matrix.zig:99-236       Pattern          pattern.zig:64-75
//   Material                        material.zig:6-18       Shape            shape.zig:73-82
//   PointLight                      light.zig:9-14          World            world.zig:15-59
//   Camera (init, render)           camera.zig:17-90        Canvas           canvas.zig:4-39
//   viewTransform                   calc.zig:721-738        canvasToPPM      ppm.zig:6-38
//
// Everything below Camera::render is NOT here: render() flattens the world into the C-ABI
// descriptors and calls librtc_hip.so.  Compile with -ffp-contract=off (the builders must not
// fuse multiply-adds either: the reference does not).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <optional>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rtc_hip.h"

namespace rtc_host {

struct Vec4 {
    double x, y, z, w;
    Vec4 add(const Vec4& o) const { return {x + o.x, y + o.y, z + o.z, w + o.w}; }
    Vec4 sub(const Vec4& o) const { return {x - o.x, y - o.y, z - o.z, w - o.w}; }
    Vec4 negate() const { return {-x, -y, -z, -w}; }
    Vec4 scale(double s) const { return {s * x, s * y, s * z, s * w}; }
    double length() const { return std::sqrt(x * x + y * y + z * z + w * w); }
    Vec4 normalize() const {
        const double len = length();
        return {x / len, y / len, z / len, w / len};
    }
    double dot(const Vec4& o) const { return x * o.x + y * o.y + z * o.z + w * o.w; }
    Vec4 cross(const Vec4& o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x, 0.0}; }
};
inline Vec4 initPoint(double x, double y, double z) { return {x, y, z, 1.0}; }
inline Vec4 initVector(double x, double y, double z) { return {x, y, z, 0.0}; }

struct Color {
    double red, green, blue;
    static Color init(double r, double g, double b) { return {r, g, b}; }
};
inline const Color Black{0, 0, 0};
inline const Color White{1, 1, 1};

struct Mat4 {
    double mat[4][4];
    static Mat4 identity() { return {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}}; }
    Mat4 mult(const Mat4& o) const {  // matrix.zig:197-213
        Mat4 r{};
        for (int row = 0; row < 4; ++row)
            for (int col = 0; col < 4; ++col)
                r.mat[row][col] = mat[row][0] * o.mat[0][col] + mat[row][1] * o.mat[1][col] +
                                  mat[row][2] * o.mat[2][col] + mat[row][3] * o.mat[3][col];
        return r;
    }
    // the fluent builders pre-multiply: m.translate(..) is T*m (matrix.zig:116-180)
    Mat4 translate(double x, double y, double z) const {
        return Mat4{{{1, 0, 0, x}, {0, 1, 0, y}, {0, 0, 1, z}, {0, 0, 0, 1}}}.mult(*this);
    }
    Mat4 scale(double x, double y, double z) const {
        return Mat4{{{x, 0, 0, 0}, {0, y, 0, 0}, {0, 0, z, 0}, {0, 0, 0, 1}}}.mult(*this);
    }
    Mat4 rotateX(double r) const {
        return Mat4{{{1, 0, 0, 0}, {0, std::cos(r), -std::sin(r), 0}, {0, std::sin(r), std::cos(r), 0}, {0, 0, 0, 1}}}
            .mult(*this);
    }
    Mat4 rotateY(double r) const {
        return Mat4{{{std::cos(r), 0, std::sin(r), 0}, {0, 1, 0, 0}, {-std::sin(r), 0, std::cos(r), 0}, {0, 0, 0, 1}}}
            .mult(*this);
    }
    Mat4 rotateZ(double r) const {
        return Mat4{{{std::cos(r), -std::sin(r), 0, 0}, {std::sin(r), std::cos(r), 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}}
            .mult(*this);
    }
    Mat4 shear(double xy, double xz, double yx, double yz, double zx, double zy) const {
        return Mat4{{{1, xy, xz, 0}, {yx, 1, yz, 0}, {zx, zy, 1, 0}, {0, 0, 0, 1}}}.mult(*this);
    }
};

struct Pattern {
    enum Kind { point = RTC_PATTERN_POINT, stripe, gradient, ring, checkers };
    Kind kind = point;
    Color a = Black, b = Black;
    Mat4 transform = Mat4::identity();
};

struct Material {
    Color color = Color::init(1, 1, 1);
    double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200.0;
    double reflective = 0.0, transparency = 0.0, refractive_index = 1.0;
    std::optional<Pattern> pattern = std::nullopt;
};

struct Shape {
    enum Geo { sphere = RTC_GEO_SPHERE, plane = RTC_GEO_PLANE, cube = RTC_GEO_CUBE /* extension, not in shape.zig:76-79 */ };
    Geo geo = sphere;
    Mat4 transform = Mat4::identity();
    Material material{};
};

struct PointLight {
    Vec4 position = initPoint(0, 0, 0);
    Color intensity = White;
};

struct World {
    std::vector<Shape> objects;
    PointLight light{};
    static World initDefault() {  // world.zig:30-54
        World w;
        w.light = {initPoint(-10, 10, -10), White};
        Shape s1;
        s1.material.color = Color::init(0.8, 1.0, 0.6);
        s1.material.diffuse = 0.7;
        s1.material.specular = 0.2;
        w.objects.push_back(s1);
        Shape s2;
        s2.transform = Mat4::identity().scale(0.5, 0.5, 0.5);
        w.objects.push_back(s2);
        return w;
    }
};

inline Mat4 viewTransform(const Vec4& from, const Vec4& to, const Vec4& up) {  // calc.zig:721-738
    const Vec4 forward = to.sub(from).normalize();
    const Vec4 left = forward.cross(up.normalize());
    const Vec4 true_up = left.cross(forward);
    const Mat4 orientation{{{left.x, left.y, left.z, 0},
                            {true_up.x, true_up.y, true_up.z, 0},
                            {-forward.x, -forward.y, -forward.z, 0},
                            {0, 0, 0, 1}}};
    return orientation.mult(Mat4::identity().translate(-from.x, -from.y, -from.z));
}

struct Canvas {  // canvas.zig:4-39
    size_t width = 0, height = 0;
    std::vector<Color> pixels;
    Canvas(size_t w, size_t h) : width(w), height(h), pixels(w * h, Color{0, 0, 0}) {}
    void set(size_t x, size_t y, const Color& c) { pixels.at(y * width + x) = c; }
    Color at(size_t x, size_t y) const { return pixels.at(y * width + x); }
};
static_assert(sizeof(Color) == 24, "the library writes canvas.pixels in place as 3 consecutive doubles");

inline uint8_t byteColorValue(double val) {  // ppm.zig:6-9
    double v = val * 256.0;
    v = v < 255.0 ? v : 255.0;
    v = v > 0.0 ? v : 0.0;
    return (uint8_t)v;
}

inline void canvasToPPM(const Canvas& canvas, std::ostream& writer) {  // ppm.zig:11-38
    writer << "P3\n" << canvas.width << " " << canvas.height << "\n255\n";
    for (size_t y = 0; y < canvas.height; ++y) {
        for (size_t x = 0; x < canvas.width; ++x) {
            if (x > 0) writer << (x % 5 == 0 ? '\n' : ' ');
            const Color p = canvas.at(x, y);
            writer << (int)byteColorValue(p.red) << ' ' << (int)byteColorValue(p.green) << ' '
                   << (int)byteColorValue(p.blue);
        }
        writer << '\n';
    }
}

// The canvas after byteColorValue (ppm.zig:6-9): what canvasToPPM prints, three bytes per pixel.
struct ByteCanvas {
    size_t width = 0, height = 0;
    std::vector<uint8_t> rgb;
    ByteCanvas(size_t w, size_t h) : width(w), height(h), rgb(w * h * 3, 0) {}
};

// The text half of canvasToPPM (ppm.zig:11-38) for colours that were quantised on the device.
inline void bytesToPPM(const ByteCanvas& canvas, std::ostream& writer) {
    std::string text = "P3\n" + std::to_string(canvas.width) + " " + std::to_string(canvas.height) + "\n255\n";
    text.reserve(text.size() + canvas.rgb.size() * 4);
    char num[256][4];
    int len[256];
    for (int v = 0; v < 256; ++v) len[v] = std::snprintf(num[v], sizeof num[v], "%d", v);
    const uint8_t* p = canvas.rgb.data();
    for (size_t y = 0; y < canvas.height; ++y) {
        for (size_t x = 0; x < canvas.width; ++x, p += 3) {
            if (x > 0) text.push_back(x % 5 == 0 ? '\n' : ' ');
            text.append(num[p[0]], (size_t)len[p[0]]);
            text.push_back(' ');
            text.append(num[p[1]], (size_t)len[p[1]]);
            text.push_back(' ');
            text.append(num[p[2]], (size_t)len[p[2]]);
        }
        text.push_back('\n');
    }
    writer.write(text.data(), (std::streamsize)text.size());
}

struct RenderError : std::runtime_error {
    int code;
    RenderError(int c, const std::string& msg) : std::runtime_error(msg), code(c) {}
};

namespace detail {
inline void flattenMat(double dst[16], const Mat4& m) {
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) dst[r * 4 + c] = m.mat[r][c];
}
inline RtcShapeDesc flattenShape(const Shape& s) {
    RtcShapeDesc d{};
    d.geo = s.geo;
    flattenMat(d.transform, s.transform);
    const Material& m = s.material;
    d.material.color[0] = m.color.red;
    d.material.color[1] = m.color.green;
    d.material.color[2] = m.color.blue;
    d.material.ambient = m.ambient;
    d.material.diffuse = m.diffuse;
    d.material.specular = m.specular;
    d.material.shininess = m.shininess;
    d.material.reflective = m.reflective;
    d.material.transparency = m.transparency;
    d.material.refractive_index = m.refractive_index;
    flattenMat(d.material.pattern.transform, Mat4::identity());
    if (m.pattern) {
        d.material.has_pattern = 1;
        d.material.pattern.kind = m.pattern->kind;
        const Color &a = m.pattern->a, &b = m.pattern->b;
        d.material.pattern.a[0] = a.red, d.material.pattern.a[1] = a.green, d.material.pattern.a[2] = a.blue;
        d.material.pattern.b[0] = b.red, d.material.pattern.b[1] = b.green, d.material.pattern.b[2] = b.blue;
        flattenMat(d.material.pattern.transform, m.pattern->transform);
    }
    return d;
}
}  // namespace detail

struct Camera {
    size_t hsize, vsize;
    double field_of_view, half_width, half_height, pixel_size;
    Mat4 transform = Mat4::identity();

    static Camera init(size_t hsize, size_t vsize, double field_of_view) {  // camera.zig:29-46
        const double half_view = std::tan(field_of_view * 0.5);
        const double aspect = (double)hsize / (double)vsize;
        Camera c{};
        c.hsize = hsize;
        c.vsize = vsize;
        c.field_of_view = field_of_view;
        c.half_width = aspect >= 1.0 ? half_view : half_view * aspect;
        c.half_height = aspect < 1.0 ? half_view : half_view / aspect;
        c.pixel_size = (2.0 * c.half_width) / (double)hsize;
        c.transform = Mat4::identity();
        return c;
    }

    // Camera.render (camera.zig:74-89) through the HIP library; the reference's depth is 5.
    // `integrator`: RTC_INTEGRATOR_POSTORDER (the reference's order of summation) or RTC_INTEGRATOR_WEIGHTED.
    Canvas render(const World& world, int depth = 5, RtcStats* stats = nullptr,
                  int integrator = RTC_INTEGRATOR_POSTORDER) const {
        // an empty canvas has no first pixel to point at: the library refuses it by name
        if (hsize == 0 || vsize == 0) throw RenderError(RTC_ERR_INVALID_ARGUMENT, "camera hsize/vsize must be non-zero");
        Canvas canvas(hsize, vsize);
        through_library(world, [&](RtcScene* scene, const RtcCameraDesc& cam) {
            if (int rc = rtc_scene_set_integrator(scene, integrator)) return rc;
            return rtc_render(scene, &cam, depth, &canvas.pixels[0].red, stats);
        });
        return canvas;
    }

    // Camera.render and the byteColorValue half of canvasToPPM (ppm.zig:6-9) in one call: the
    // colours are quantised on the device and bytesToPPM only formats.
    ByteCanvas renderBytes(const World& world, int depth = 5, RtcStats* stats = nullptr,
                           int integrator = RTC_INTEGRATOR_POSTORDER) const {
        if (hsize == 0 || vsize == 0) throw RenderError(RTC_ERR_INVALID_ARGUMENT, "camera hsize/vsize must be non-zero");
        ByteCanvas canvas(hsize, vsize);
        through_library(world, [&](RtcScene* scene, const RtcCameraDesc& cam) {
            if (int rc = rtc_scene_set_integrator(scene, integrator)) return rc;
            return rtc_render_u8(scene, &cam, depth, canvas.rgb.data(), stats);
        });
        return canvas;
    }

    // scene upload, the call, scene release; a failure becomes a RenderError
    template <class Call>
    void through_library(const World& world, Call&& call) const {
        std::vector<RtcShapeDesc> shapes;
        shapes.reserve(world.objects.size());
        for (const Shape& s : world.objects) shapes.push_back(detail::flattenShape(s));
        RtcLightDesc light{};
        light.position[0] = world.light.position.x;
        light.position[1] = world.light.position.y;
        light.position[2] = world.light.position.z;
        light.position[3] = world.light.position.w;
        light.intensity[0] = world.light.intensity.red;
        light.intensity[1] = world.light.intensity.green;
        light.intensity[2] = world.light.intensity.blue;
        RtcCameraDesc cam{};
        cam.hsize = hsize;
        cam.vsize = vsize;
        cam.field_of_view = field_of_view;
        cam.half_width = half_width;
        cam.half_height = half_height;
        cam.pixel_size = pixel_size;
        detail::flattenMat(cam.transform, transform);
        RtcScene* scene = nullptr;
        int rc = rtc_scene_create(shapes.data(), shapes.size(), &light, &scene);
        if (rc != RTC_OK) throw RenderError(rc, rtc_last_error());
        rc = call(scene, cam);
        const std::string msg = rc != RTC_OK ? rtc_last_error() : "";
        rtc_scene_destroy(scene);
        if (rc != RTC_OK) throw RenderError(rc, msg);
    }
};

}  // namespace rtc_host
