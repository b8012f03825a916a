"""A tiny line-based scene format, so that worlds can be handed to the render path without
the reference's Zig scene-construction code (draw_world.zig:29-192 builds its world in code).
The same format is read by the C++ host side (host/scene_text.hpp).

    # comment
    camera <hsize> <vsize> <field_of_view>
    view <from x y z> <to x y z> <up x y z>       camera.transform = viewTransform (calc.zig:721-738)
    light <x y z> <r g b>
    material <name> key=value ...                  a named material (material.zig:6-18 defaults)
    sphere|plane|cube [material=<name>] key=value ... [transform=<ops>]

keys    color=r,g,b ambient= diffuse= specular= shininess= reflective= transparency=
        refractive_index= pattern=point|stripe|gradient|ring|checkers pattern_a=r,g,b
        pattern_b=r,g,b pattern_transform=<ops>
<ops>   a chain like scale(0.5,0.5,0.5).rotate_y(1.5708).translate(0,1,0), applied left to right
        exactly like the reference's fluent builders (each op pre-multiplies, matrix.zig:116-180);
        ops: translate scale rotate_x rotate_y rotate_z shear matrix(16 row-major values).
Numbers are decimal literals parsed correctly rounded (as Zig parses its own literals); values
contain no spaces.
"""
from .host import (Camera, Color, Mat4, Material, Pattern, PointLight, Shape, World, init_point, init_vector,
                   view_transform)

_MATERIAL_KEYS = ("ambient", "diffuse", "specular", "shininess", "reflective", "transparency", "refractive_index")


class SceneTextError(ValueError):
    pass


def _floats(text, n, what):
    parts = text.split(",")
    if len(parts) != n:
        raise SceneTextError(f"{what}: expected {n} comma-separated numbers, got {text!r}")
    return [float(p) for p in parts]


def parse_ops(text):
    """'scale(..).rotate_y(..)' -> Mat4, applied left to right with the fluent builders."""
    m = Mat4.identity()
    rest = text.strip()
    while rest:
        if rest[0] == ".":
            rest = rest[1:]
        name, sep, tail = rest.partition("(")
        args, sep2, rest = tail.partition(")")
        if not sep or not sep2:
            raise SceneTextError(f"malformed transform {text!r}")
        name = name.strip()
        if name == "translate":
            m = m.translate(*_floats(args, 3, name))
        elif name == "scale":
            m = m.scale(*_floats(args, 3, name))
        elif name in ("rotate_x", "rotate_y", "rotate_z"):
            m = getattr(m, name)(*_floats(args, 1, name))
        elif name == "shear":
            m = m.shear(*_floats(args, 6, name))
        elif name == "matrix":
            v = _floats(args, 16, name)
            m = Mat4([v[0:4], v[4:8], v[8:12], v[12:16]]).mult(m)
        else:
            raise SceneTextError(f"unknown transform op {name!r}")
    return m


def _apply_material_keys(mat, kv):
    pat_kind = kv.pop("pattern", None)
    pat_a, pat_b, pat_t = kv.pop("pattern_a", None), kv.pop("pattern_b", None), kv.pop("pattern_transform", None)
    if "color" in kv:
        mat.color = Color(*_floats(kv.pop("color"), 3, "color"))
    for k in _MATERIAL_KEYS:
        if k in kv:
            setattr(mat, k, float(kv.pop(k)))
    if pat_kind is not None:
        mat.pattern = Pattern(pat_kind,
                              Color(*_floats(pat_a, 3, "pattern_a")) if pat_a else None,
                              Color(*_floats(pat_b, 3, "pattern_b")) if pat_b else None,
                              parse_ops(pat_t) if pat_t else None)
    elif pat_a or pat_b or pat_t:
        raise SceneTextError("pattern_a / pattern_b / pattern_transform without pattern=")


def _copy_material(m):
    return Material(m.color, m.ambient, m.diffuse, m.specular, m.shininess, m.reflective, m.transparency,
                    m.refractive_index, m.pattern)


def loads(text, hsize=None, vsize=None):
    """-> (World, Camera).  hsize/vsize override the file's canvas size (the aspect rule of
    Camera.init, camera.zig:33-34, does the rest)."""
    world, materials = World(), {}
    cam_args, view, light_seen = None, None, False
    for no, raw in enumerate(text.splitlines(), 1):
        line = raw.split("#", 1)[0].split()
        if not line:
            continue
        try:
            head, rest = line[0], line[1:]
            if head == "camera":
                cam_args = (int(rest[0]), int(rest[1]), float(rest[2]))
            elif head == "view":
                v = [float(x) for x in rest]
                if len(v) != 9:
                    raise SceneTextError("view takes 9 numbers")
                view = v
            elif head == "light":
                v = [float(x) for x in rest]
                if len(v) != 6:
                    raise SceneTextError("light takes 6 numbers")
                world.light = PointLight(init_point(*v[:3]), Color(*v[3:]))
                light_seen = True
            elif head in ("material", "sphere", "plane", "cube"):
                if head == "material":
                    name, rest = rest[0], rest[1:]
                kv = {}
                for item in rest:
                    k, sep, val = item.partition("=")
                    if not sep:
                        raise SceneTextError(f"expected key=value, got {item!r}")
                    kv[k] = val
                base = kv.pop("material", None)
                if base is not None and base not in materials:
                    raise SceneTextError(f"unknown material {base!r}")
                mat = _copy_material(materials[base]) if base is not None else Material()
                transform = parse_ops(kv.pop("transform")) if "transform" in kv else Mat4.identity()
                _apply_material_keys(mat, kv)
                if kv:
                    raise SceneTextError(f"unknown key(s) {sorted(kv)}")
                if head == "material":
                    materials[name] = mat
                else:
                    world.objects.append(Shape(head, transform, mat))
            else:
                raise SceneTextError(f"unknown directive {head!r}")
        except (IndexError, ValueError) as e:
            raise SceneTextError(f"line {no}: {e}") from None
    if cam_args is None or not light_seen:
        raise SceneTextError("a scene needs a camera and a light")
    camera = Camera(hsize or cam_args[0], vsize or cam_args[1], cam_args[2])
    if view is not None:
        # like draw_world.zig:36-39 the three are built with initVector; only differences are used
        camera.transform = view_transform(init_vector(*view[0:3]), init_vector(*view[3:6]), init_vector(*view[6:9]))
    return world, camera


def load(path, hsize=None, vsize=None):
    with open(path) as f:
        return loads(f.read(), hsize, vsize)


def _matrix_op(m):
    return "matrix(" + ",".join(repr(v) for row in m.mat for v in row) + ")"


def dumps(world, camera, view=None):
    """Serialise any world: transforms are written as explicit matrices (17 significant digits
    round-trip exactly).  view = (from, to, up) triples writes a `view` line; without it the
    camera transform is not representable and must be the identity."""
    out = [f"camera {camera.hsize} {camera.vsize} {camera.field_of_view!r}"]
    if view is not None:
        out.append("view " + " ".join(repr(float(x)) for triple in view for x in triple))
    elif camera.transform.mat != Mat4.identity().mat:
        raise SceneTextError("dumps needs view=(from, to, up) for a transformed camera")
    p, i = world.light.position, world.light.intensity
    out.append(f"light {p.x!r} {p.y!r} {p.z!r} {i.red!r} {i.green!r} {i.blue!r}")
    for s in world.objects:
        m = s.material
        items = [s.geo, f"color={m.color.red!r},{m.color.green!r},{m.color.blue!r}"]
        items += [f"{k}={getattr(m, k)!r}" for k in _MATERIAL_KEYS]
        if m.pattern is not None:
            q = m.pattern
            items += [f"pattern={q.kind}", f"pattern_a={q.a.red!r},{q.a.green!r},{q.a.blue!r}",
                      f"pattern_b={q.b.red!r},{q.b.green!r},{q.b.blue!r}", f"pattern_transform={_matrix_op(q.transform)}"]
        items.append(f"transform={_matrix_op(s.transform)}")
        out.append(" ".join(items))
    return "\n".join(out) + "\n"
