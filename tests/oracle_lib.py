"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSET_DIR = os.path.join(ROOT, "assets", "models")
IMAGE_DIR = os.path.join(ROOT, "assets", "img")


class Material(C.Structure):
    _fields_ = [("ka", C.c_float * 3), ("kd", C.c_float * 3), ("ks", C.c_float * 3), ("pad0", C.c_float),
                ("shininess", C.c_double), ("type", C.c_int32), ("ior", C.c_float),
                ("transparency", C.c_float), ("reflectivity", C.c_float),
                ("texture", C.c_int32), ("normal_map", C.c_int32), ("material_scale", C.c_double)]


class Camera(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("zoom", C.c_float),
                ("quat", C.c_float * 4), ("fovy_deg", C.c_float), ("aspect", C.c_float),
                ("znear", C.c_float), ("zfar", C.c_float)]


class Params(C.Structure):
    _fields_ = [("img_w", C.c_int32), ("img_h", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
                ("light_samples", C.c_int32), ("shadow_mode", C.c_int32), ("jitter", C.c_int32),
                ("seed", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays_primary", "rays_secondary", "rays_shadow", "box_tests",
                                          "tri_tests", "node_visits", "sh_box_tests", "sh_tri_tests",
                                          "sh_node_visits")]


HIT_DTYPE = np.dtype([("exists", "<i4"), ("obj", "<i4"), ("tri", "<i4"), ("t", "<f4"),
                      ("pos", "<f4", 3), ("nrm", "<f4", 3), ("u", "<f4"), ("v", "<f4")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.orc_scene_new.restype = C.c_void_p
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_set_asset_dir.argtypes = [C.c_char_p]
        L.orc_scene_setup.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_build.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_dump_tree.restype = C.c_int64
        L.orc_dump_tree.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.orc_render.restype = C.c_double
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_void_p, C.c_void_p, C.POINTER(Stats), C.c_int]
        L.orc_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_shadow.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(Params), C.c_void_p,
                                C.c_void_p, C.c_void_p]
        L.orc_camera_rays.argtypes = [C.POINTER(Camera), C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p]
        L.orc_quantize.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_rng.restype = C.c_uint32
        L.orc_rng.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_counts.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
        L.orc_get_sphere.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(Material)]
        L.orc_get_square.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(Material)]
        L.orc_get_light.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_void_p]
        L.orc_get_mesh_counts.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.orc_get_mesh.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Material)]
        L.orc_add_light.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.orc_add_square.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Material)]
        L.orc_add_sphere.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.POINTER(Material)]
        L.orc_add_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(Material)]
        L.orc_rotmatrix.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_load_off.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                   C.POINTER(C.c_int)]
        L.orc_set_image_dir.argtypes = [C.c_char_p]
        L.orc_add_texture.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_get_texture.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]
        L.orc_load_ppm.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_int64]
        L.orc_set_asset_dir(ASSET_DIR.encode())
        L.orc_set_image_dir(IMAGE_DIR.encode())
        _lib = L
    return _lib


def default_camera(aspect=1.0, zoom=3.0, z=-3.1):
    """Camera() defaults (Camera.cpp:24-38) + camera.move(0,0,-3.1) (main.cpp:385)."""
    c = Camera()
    c.x, c.y, c.z, c.zoom = 0.0, 0.0, z, zoom
    c.quat[:] = [0.0, 0.0, 0.0, 1.0]
    c.fovy_deg, c.aspect, c.znear, c.zfar = 45.0, aspect, 4.1, 10000.0
    return c


def make_params(w, h, spp=1, max_depth=50, light_samples=35, shadow_mode=0, jitter=0, seed=0):
    return Params(w, h, spp, max_depth, light_samples, shadow_mode, jitter, seed)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def load_ppm(path):
    """Binary PPM -> uint8 array (h, w, 3) through the oracle's reader."""
    w, h = C.c_int(), C.c_int()
    if lib().orc_load_ppm(path.encode(), C.byref(w), C.byref(h), None, 0) != 0:
        raise RuntimeError(f"cannot read {path}")
    px = np.zeros((h.value, w.value, 3), np.uint8)
    assert lib().orc_load_ppm(path.encode(), C.byref(w), C.byref(h), _ptr(px), px.size) == 0
    return px


class Scene:
    def __init__(self, name=None, top_depth=3, mesh_depth=12):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_scene_new())
        if name is not None:
            rc = self.L.orc_scene_setup(self.h, name.encode())
            if rc != 0:
                raise RuntimeError(f"orc_scene_setup({name}) = {rc}")
            self.build(top_depth, mesh_depth)

    def __del__(self):
        try:
            self.L.orc_scene_free(self.h)
        except Exception:
            pass

    def build(self, top_depth=3, mesh_depth=12):
        self.L.orc_build(self.h, top_depth, mesh_depth)

    def counts(self):
        v = [C.c_int() for _ in range(4)]
        self.L.orc_counts(self.h, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)

    def sphere(self, i):
        c = np.zeros(3, np.float32)
        r = C.c_float()
        m = Material()
        assert self.L.orc_get_sphere(self.h, i, _ptr(c), C.byref(r), C.byref(m)) == 0
        return c, r.value, m

    def square(self, i):
        v = np.zeros(12, np.float32)
        m = Material()
        assert self.L.orc_get_square(self.h, i, _ptr(v), C.byref(m)) == 0
        return v.reshape(4, 3), m

    def light(self, i):
        p = np.zeros(3, np.float32)
        col = np.zeros(3, np.float32)
        r = C.c_float()
        assert self.L.orc_get_light(self.h, i, _ptr(p), C.byref(r), _ptr(col)) == 0
        return p, r.value, col

    def mesh(self, i):
        nv, nt = C.c_int64(), C.c_int64()
        assert self.L.orc_get_mesh_counts(self.h, i, C.byref(nv), C.byref(nt)) == 0
        v = np.zeros((nv.value, 3), np.float32)
        n = np.zeros((nv.value, 3), np.float32)
        t = np.zeros((nt.value, 3), np.uint32)
        m = Material()
        self.L.orc_get_mesh(self.h, i, _ptr(v), _ptr(n), _ptr(t), C.byref(m))
        return v, n, t, m

    def tree(self, which):
        """(leaf flags, member counts, boxes[n,6], items) in DFS pre-order."""
        ni = C.c_int64()
        nn = self.L.orc_dump_tree(self.h, which, None, None, None, 0, None, 0, C.byref(ni))
        leaf = np.zeros(nn, np.int32)
        cnt = np.zeros(nn, np.int32)
        boxes = np.zeros((nn, 6), np.float32)
        items = np.zeros(max(ni.value, 1), np.int64)
        self.L.orc_dump_tree(self.h, which, _ptr(leaf), _ptr(cnt), _ptr(boxes), nn, _ptr(items), ni.value, C.byref(ni))
        return leaf, cnt, boxes, items[:ni.value]

    def add_light(self, pos, radius, color=(1, 1, 1)):
        p = np.asarray(pos, np.float32)
        c = np.asarray(color, np.float32)
        self.L.orc_add_light(self.h, _ptr(p), radius, _ptr(c))

    def texture(self, i):
        """Texel array (h, w, channels) of image i (0-based) of the scene."""
        w, h, ch = C.c_int(), C.c_int(), C.c_int()
        assert self.L.orc_get_texture(self.h, i, C.byref(w), C.byref(h), C.byref(ch), None) == 0
        px = np.zeros((h.value, w.value, ch.value), np.uint8)
        assert self.L.orc_get_texture(self.h, i, C.byref(w), C.byref(h), C.byref(ch), _ptr(px)) == 0
        return px

    def add_texture(self, texels):
        """texels: uint8 array (h, w, channels>=3), rows top to bottom. Returns the value for Material.texture / normal_map."""
        t = np.ascontiguousarray(texels, np.uint8)
        idx = self.L.orc_add_texture(self.h, t.shape[1], t.shape[0], t.shape[2], _ptr(t))
        if idx < 0:
            raise RuntimeError("orc_add_texture failed")
        return idx + 1

    def add_square(self, verts, mat):
        v = np.ascontiguousarray(verts, np.float32).reshape(12)
        if self.L.orc_add_square(self.h, _ptr(v), C.byref(mat)) != 0:
            raise RuntimeError("orc_add_square failed")

    def add_sphere(self, center, radius, mat):
        c = np.asarray(center, np.float32)
        if self.L.orc_add_sphere(self.h, _ptr(c), radius, C.byref(mat)) != 0:
            raise RuntimeError("orc_add_sphere failed")

    def add_mesh(self, verts, tris, mat):
        v = np.ascontiguousarray(verts, np.float32)
        t = np.ascontiguousarray(tris, np.uint32)
        if self.L.orc_add_mesh(self.h, _ptr(v), len(v), _ptr(t), len(t), C.byref(mat)) != 0:
            raise RuntimeError("orc_add_mesh failed")

    def intersect(self, org, dirs):
        org = np.ascontiguousarray(org, np.float32)
        dirs = np.ascontiguousarray(dirs, np.float32)
        out = np.zeros(len(org), HIT_DTYPE)
        self.L.orc_intersect(self.h, _ptr(org), _ptr(dirs), len(org), _ptr(out))
        return out

    def shadow(self, pts, lights):
        pts = np.ascontiguousarray(pts, np.float32)
        lights = np.ascontiguousarray(lights, np.float32)
        out = np.zeros(len(pts), np.int32)
        self.L.orc_shadow(self.h, _ptr(pts), _ptr(lights), len(pts), _ptr(out))
        return out

    def trace(self, org, dirs, params, pixel=None, sample=None):
        org = np.ascontiguousarray(org, np.float32)
        dirs = np.ascontiguousarray(dirs, np.float32)
        n = len(org)
        pixel = np.arange(n, dtype=np.uint32) if pixel is None else np.ascontiguousarray(pixel, np.uint32)
        sample = np.zeros(n, np.uint32) if sample is None else np.ascontiguousarray(sample, np.uint32)
        rgb = np.zeros((n, 3), np.float32)
        self.L.orc_trace(self.h, _ptr(org), _ptr(dirs), n, C.byref(params), _ptr(pixel), _ptr(sample), _ptr(rgb))
        return rgb

    def render(self, cam, params, x0=0, y0=0, w=None, h=None, threads=8):
        w = params.img_w if w is None else w
        h = params.img_h if h is None else h
        rgb = np.zeros((h, w, 3), np.float32)
        prim = np.zeros((h, w, 2), np.int32)
        st = Stats()
        secs = self.L.orc_render(self.h, C.byref(cam), C.byref(params), x0, y0, w, h, _ptr(rgb), _ptr(prim),
                                 C.byref(st), threads)
        return rgb, prim, st, secs


def camera_rays(cam, params, x0, y0, w, h):
    n = w * h * params.spp
    org = np.zeros((n, 3), np.float32)
    dirs = np.zeros((n, 3), np.float32)
    lib().orc_camera_rays(C.byref(cam), C.byref(params), x0, y0, w, h, _ptr(org), _ptr(dirs))
    return org, dirs


def quantize(rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    out = np.zeros(rgb.shape, np.int32)
    lib().orc_quantize(_ptr(rgb), rgb.size, _ptr(out))
    return out


def rng(seed, pixel, sample, code, light, i, which):
    return lib().orc_rng(seed, pixel, sample, code, light, i, which)
