"""ray-tracing_amd -- MI355X-native render loop for Crhonopost/Ray-tracing.

Thin ctypes layer over ``librtgpu.so`` (hand-written HIP kernels + C-ABI, include/rtgpu.h,
and the host-side Scene/Camera/Mesh mirror, include/rthost.h).  There is no CPU fallback:
if the library is missing, or no GPU is visible when a device call is made, it raises.

Import with ``importlib.import_module("ray-tracing_amd")`` (the hyphen is the repo's name).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("RT_LIB", os.path.join(_HERE, "librtgpu.so"))   # RT_LIB: developer A/B knob
# data files of the reference's setup_* scenes (its assets/models/*.off and assets/img/**.ppm; data, not source)
DEFAULT_ASSET_DIR = os.environ.get("RT_ASSET_DIR", os.path.join(ROOT, "assets", "models"))
DEFAULT_IMAGE_DIR = os.environ.get("RT_IMAGE_DIR", os.path.join(ROOT, "assets", "img"))

RT_OK = 0


class RtError(RuntimeError):
    pass


class Material(C.Structure):
    _fields_ = [("ka", C.c_float * 3), ("kd", C.c_float * 3), ("ks", C.c_float * 3), ("pad0", C.c_float),
                ("shininess", C.c_double), ("type", C.c_int32), ("ior", C.c_float),
                ("transparency", C.c_float), ("reflectivity", C.c_float),
                ("texture", C.c_int32), ("normal_map", C.c_int32), ("material_scale", C.c_double)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32), ("pad0", C.c_int32),
                ("texels", C.POINTER(C.c_uint8))]


class Light(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("radius", C.c_float), ("color", C.c_float * 3), ("pad0", C.c_float)]


class Square(C.Structure):
    _fields_ = [("v0", C.c_float * 3), ("v1", C.c_float * 3), ("v3", C.c_float * 3), ("material", C.c_int32)]


class Sphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("material", C.c_int32), ("pad0", C.c_int32 * 3)]


class BvhNode(C.Structure):
    _fields_ = [("lmin", C.c_float * 3), ("lmax", C.c_float * 3), ("rmin", C.c_float * 3), ("rmax", C.c_float * 3),
                ("left", C.c_int32), ("right", C.c_int32), ("pad0", C.c_int32 * 2)]


class BvhLeaf(C.Structure):
    _fields_ = [("first", C.c_uint32), ("count", C.c_uint32)]


class Mesh(C.Structure):
    _fields_ = [("vert_first", C.c_int64), ("vert_count", C.c_int64), ("tri_first", C.c_int64), ("tri_count", C.c_int64),
                ("node_first", C.c_int64), ("node_count", C.c_int64), ("leaf_first", C.c_int64), ("leaf_count", C.c_int64),
                ("root", C.c_int32), ("material", C.c_int32), ("depth", C.c_int32), ("pad0", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("n_materials", C.c_int32), ("materials", C.POINTER(Material)),
                ("n_lights", C.c_int32), ("lights", C.POINTER(Light)),
                ("n_squares", C.c_int32), ("squares", C.POINTER(Square)),
                ("n_spheres", C.c_int32), ("spheres", C.POINTER(Sphere)),
                ("n_meshes", C.c_int32), ("meshes", C.POINTER(Mesh)),
                ("n_verts", C.c_int64), ("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)),
                ("n_tris", C.c_int64), ("tri_indices", C.POINTER(C.c_uint32)),
                ("mesh_items", C.POINTER(C.c_uint32)),
                ("n_mesh_nodes", C.c_int64), ("mesh_nodes", C.POINTER(BvhNode)),
                ("n_mesh_leaves", C.c_int64), ("mesh_leaves", C.POINTER(BvhLeaf)),
                ("n_top_nodes", C.c_int32), ("top_nodes", C.POINTER(BvhNode)),
                ("n_top_leaves", C.c_int32), ("top_leaves", C.POINTER(BvhLeaf)),
                ("n_top_items", C.c_int32), ("top_items", C.POINTER(C.c_uint32)),
                ("top_root", C.c_int32), ("top_depth", C.c_int32),
                ("n_textures", C.c_int32), ("textures", C.POINTER(Texture))]


class Camera(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("zoom", C.c_float),
                ("quat", C.c_float * 4), ("fovy_deg", C.c_float), ("aspect", C.c_float),
                ("znear", C.c_float), ("zfar", C.c_float)]


class RenderParams(C.Structure):
    _fields_ = [("img_w", C.c_int32), ("img_h", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
                ("light_samples", C.c_int32), ("shadow_mode", C.c_int32), ("jitter", C.c_int32),
                ("seed", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_secondary", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("nonfinite", C.c_uint64), ("iterations", C.c_uint64), ("rays_shadow_culled", C.c_uint64),
                ("ms_total", C.c_double), ("ms_closest", C.c_double), ("ms_shadow", C.c_double),
                ("launches_closest", C.c_uint64), ("launches_shadow", C.c_uint64), ("peak_states", C.c_uint64),
                ("rays_shadow_grouped", C.c_uint64)]

    @property
    def rays(self):
        """Rays traced (one ray = one BVH_Node::intersectObjects query); shadow queries settled by the group test are not rays."""
        return self.rays_primary + self.rays_secondary + self.rays_shadow - self.rays_shadow_grouped

    @property
    def shadow_queries(self):
        """Scene::isInShadow calls of the CPU path for the same frame: traced + culled (see rtgpu.h, rays_shadow_culled)."""
        return self.rays_shadow + self.rays_shadow_culled


HIT_DTYPE = np.dtype([("exists", "<i4"), ("obj", "<i4"), ("tri", "<i4"), ("t", "<f4"),
                      ("pos", "<f4", 3), ("nrm", "<f4", 3), ("u", "<f4"), ("v", "<f4")])

# every symbol include/rtgpu.h and include/rthost.h declare
EXPORTS = ["rt_init", "rt_tile_owner", "rt_shutdown", "rt_last_error", "rt_upload_scene", "rt_set_camera", "rt_begin_frame", "rt_begin_frame_samples", "rt_export_accumulators", "rt_import_accumulators", "rt_continue_frame",
           "rt_render_tile", "rt_render_tiles", "rt_flush", "rt_download_framebuffer", "rt_resolve_device", "rt_resolve_rgba8_device", "rt_get_stats",
           "rt_intersect", "rt_occluded", "rt_group_clear", "rt_trace", "rt_trace_static", "rt_camera_rays", "rt_rng", "rt_uses_quantised_shadow_nodes", "rt_hit_excursion", "rt_filter_passes",
           "rth_scene_new", "rth_scene_free", "rth_set_asset_dir", "rth_set_image_dir", "rth_add_texture", "rth_scene_setup", "rth_add_light",
           "rth_add_square", "rth_add_sphere", "rth_add_mesh", "rth_build", "rth_build_mode", "rth_scene_desc",
           "rth_ray_trace_from_camera", "rth_write_ppm", "rth_load_off", "rth_scene_save", "rth_scene_load"]

_lib = None


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def lib():
    """Loads librtgpu.so (never a fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(f"{LIB_PATH} is missing: build it with `make -C ray-tracing_amd/csrc` "
                      "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    L.rt_init.argtypes = [C.POINTER(C.c_int), i32, C.POINTER(vp)]
    L.rt_tile_owner.argtypes = [i32, i32, i32]
    L.rt_shutdown.argtypes = [vp]
    L.rt_shutdown.restype = None
    L.rt_last_error.argtypes = [vp]
    L.rt_last_error.restype = C.c_char_p
    L.rt_upload_scene.argtypes = [vp, C.POINTER(SceneDesc)]
    L.rt_set_camera.argtypes = [vp, C.POINTER(Camera)]
    L.rt_begin_frame.argtypes = [vp, C.POINTER(RenderParams)]
    L.rt_continue_frame.argtypes = [vp, i32]
    L.rt_begin_frame_samples.argtypes = [vp, C.POINTER(RenderParams), i32, i32]
    L.rt_export_accumulators.argtypes = [vp, vp, vp]
    L.rt_import_accumulators.argtypes = [vp, vp, vp]
    L.rt_render_tile.argtypes = [vp, i32, i32, i32, i32]
    L.rt_render_tiles.argtypes = [vp, vp, i64]
    L.rt_flush.argtypes = [vp]
    L.rt_download_framebuffer.argtypes = [vp, vp, vp, C.POINTER(Stats)]
    L.rt_resolve_device.argtypes = [vp, vp]
    L.rt_resolve_rgba8_device.argtypes = [vp, vp, i32]
    L.rt_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.rt_uses_quantised_shadow_nodes.argtypes = [vp]
    L.rt_hit_excursion.argtypes = [vp, vp, vp, C.c_float]
    L.rt_hit_excursion.restype = C.c_double
    L.rt_filter_passes.argtypes = [vp, vp, vp, C.c_float, vp, vp]
    L.rt_filter_passes.restype = C.c_int
    L.rt_intersect.argtypes = [vp, vp, vp, i64, vp]
    L.rt_occluded.argtypes = [vp, vp, vp, i64, vp]
    L.rt_group_clear.argtypes = [vp, vp, vp, vp, i64, vp]
    L.rt_trace.argtypes = [vp, vp, vp, vp, vp, i64, C.POINTER(RenderParams), vp]
    L.rt_trace_static.argtypes = [vp, vp, vp, vp, vp, i64, C.POINTER(RenderParams), vp]
    L.rt_camera_rays.argtypes = [vp, C.POINTER(RenderParams), i32, i32, i32, i32, vp, vp]
    L.rt_rng.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
    L.rt_rng.restype = C.c_uint32
    L.rth_scene_new.restype = vp
    L.rth_scene_free.argtypes = [vp]
    L.rth_scene_free.restype = None
    L.rth_set_asset_dir.argtypes = [C.c_char_p]
    L.rth_set_asset_dir.restype = None
    L.rth_set_image_dir.argtypes = [C.c_char_p]
    L.rth_set_image_dir.restype = None
    L.rth_add_texture.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
    L.rth_scene_setup.argtypes = [vp, C.c_char_p]
    L.rth_add_light.argtypes = [vp, vp, C.c_float, vp]
    L.rth_add_square.argtypes = [vp, vp, C.POINTER(Material)]
    L.rth_add_sphere.argtypes = [vp, vp, C.c_float, C.POINTER(Material)]
    L.rth_add_mesh.argtypes = [vp, vp, i64, vp, i64, C.POINTER(Material)]
    L.rth_build.argtypes = [vp, i32, i32]
    L.rth_build_mode.argtypes = [vp, i32, i32, i32]
    L.rth_scene_desc.argtypes = [vp]
    L.rth_scene_desc.restype = C.POINTER(SceneDesc)
    L.rth_ray_trace_from_camera.argtypes = [vp, vp, C.POINTER(Camera), C.POINTER(RenderParams), C.c_char_p,
                                            C.POINTER(C.c_double)]
    L.rth_write_ppm.argtypes = [C.c_char_p, vp, i32, i32]
    L.rth_load_off.argtypes = [C.c_char_p, vp, i64, vp, i64, C.POINTER(i64), C.POINTER(i64)]
    L.rth_scene_save.argtypes = [vp, C.c_char_p]
    L.rth_scene_load.argtypes = [vp, C.c_char_p]
    L.rth_set_asset_dir(DEFAULT_ASSET_DIR.encode())
    L.rth_set_image_dir(DEFAULT_IMAGE_DIR.encode())
    _lib = L
    return L


def default_camera(aspect=1.0, zoom=3.0, z=-3.1):
    """Camera() defaults (Camera.cpp:24-38) after camera.move(0,0,-3.1) (main.cpp:385)."""
    c = Camera()
    c.x, c.y, c.z, c.zoom = 0.0, 0.0, z, zoom
    c.quat[:] = [0.0, 0.0, 0.0, 1.0]
    c.fovy_deg, c.aspect, c.znear, c.zfar = 45.0, aspect, 4.1, 10000.0
    return c


def make_params(w, h, spp=30, max_depth=50, light_samples=35, shadow_mode=0, jitter=1, seed=0):
    """Defaults = the reference's literals (main.cpp:189, Scene.h:214,284)."""
    return RenderParams(w, h, spp, max_depth, light_samples, shadow_mode, jitter, seed)


def make_material(kd=(1, 1, 1), ks=None, ka=(0, 0, 0), shininess=0.0, type=0, ior=1.0, transparency=0.0,
                  reflectivity=0.0):
    m = Material()
    ks = kd if ks is None else ks
    m.ka[:] = ka
    m.kd[:] = kd
    m.ks[:] = ks
    m.shininess, m.type, m.ior, m.transparency, m.reflectivity = shininess, type, ior, transparency, reflectivity
    return m


def interleaved_tiles(w, h, rank=0, world=1, tile=32):
    """Image-space tiles dealt round-robin to ranks (the ThreadPool's pixel tasks, sharded)."""
    out = []
    k = 0
    for y in range(0, h, tile):
        for x in range(0, w, tile):
            if k % world == rank:
                out.append((x, y, min(tile, w - x), min(tile, h - y)))
            k += 1
    return out


class HostScene:
    """Host-side Scene mirror (rthost.h): named setup_* scenes or free-form geometry."""

    def __init__(self, name=None, top_depth=3, mesh_depth=12, tree_mode=0):
        self.L = lib()
        self.h = C.c_void_p(self.L.rth_scene_new())
        if name is not None:
            rc = self.L.rth_scene_setup(self.h, name.encode())
            if rc != RT_OK:
                raise RtError(f"rth_scene_setup({name!r}) failed ({rc}); asset dir ok?")
            self.build(top_depth, mesh_depth, tree_mode)

    @classmethod
    def from_file(cls, path, top_depth=3, mesh_depth=12):
        """A scene description file (rthost.h, Scene::load), built like a named scene."""
        s = cls()
        rc = s.L.rth_scene_load(s.h, os.fsencode(path))
        if rc != RT_OK:
            raise RtError(f"rth_scene_load({path!r}) failed ({rc})")
        s.build(top_depth, mesh_depth)
        return s

    def save(self, path):
        rc = self.L.rth_scene_save(self.h, os.fsencode(path))
        if rc != RT_OK:
            raise RtError(f"rth_scene_save({path!r}) failed ({rc})")

    def __del__(self):
        try:
            self.L.rth_scene_free(self.h)
        except Exception:
            pass

    def add_light(self, pos, radius, color=(1, 1, 1)):
        p = np.asarray(pos, np.float32)
        c = np.asarray(color, np.float32)
        self.L.rth_add_light(self.h, _ptr(p), radius, _ptr(c))

    def add_texture(self, texels):
        """texels: uint8 array (h, w, channels>=3), rows top to bottom (stbi_load layout).
        Returns the value for Material.texture / normal_map."""
        t = np.ascontiguousarray(texels, np.uint8)
        rc = self.L.rth_add_texture(self.h, t.shape[1], t.shape[0], t.shape[2], _ptr(t))
        if rc <= 0:
            raise RtError(f"rth_add_texture failed ({rc})")
        return rc

    def add_square(self, verts, mat):
        v = np.ascontiguousarray(verts, np.float32).reshape(12)
        rc = self.L.rth_add_square(self.h, _ptr(v), C.byref(mat))
        if rc != RT_OK:
            raise RtError(f"rth_add_square failed ({rc})")

    def add_sphere(self, center, radius, mat):
        c = np.asarray(center, np.float32)
        rc = self.L.rth_add_sphere(self.h, _ptr(c), radius, C.byref(mat))
        if rc != RT_OK:
            raise RtError(f"rth_add_sphere failed ({rc})")

    def add_mesh(self, verts, tris, mat):
        v = np.ascontiguousarray(verts, np.float32)
        t = np.ascontiguousarray(tris, np.uint32)
        rc = self.L.rth_add_mesh(self.h, _ptr(v), len(v), _ptr(t), len(t), C.byref(mat))
        if rc != RT_OK:
            raise RtError(f"rth_add_mesh failed ({rc})")

    def build(self, top_depth=3, mesh_depth=12, tree_mode=0):
        """tree_mode 0: the reference's tree builder (parity); 1: binned SAH on the host, 2: linear BVH built on the GPU (both free topology: no parity)."""
        rc = self.L.rth_build_mode(self.h, top_depth, mesh_depth, tree_mode)
        if rc != RT_OK:
            raise RtError(f"rth_build failed ({rc})")

    @property
    def desc(self):
        return self.L.rth_scene_desc(self.h)

    # ---- numpy views of the flattened scene (copies) ----
    def arrays(self):
        d = self.desc.contents

        def arr(ptr, n, dtype, width=1):
            if n == 0:
                return np.zeros((0, width) if width > 1 else (0,), dtype)
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * width * np.dtype(dtype).itemsize,))
            a = a.view(dtype).copy()
            return a.reshape(n, width) if width > 1 else a

        node_dt = np.dtype([("lmin", "<f4", 3), ("lmax", "<f4", 3), ("rmin", "<f4", 3), ("rmax", "<f4", 3),
                            ("left", "<i4"), ("right", "<i4"), ("pad", "<i4", 2)])
        leaf_dt = np.dtype([("first", "<u4"), ("count", "<u4")])
        out = {
            "positions": arr(d.positions, d.n_verts, np.float32, 3),
            "normals": arr(d.normals, d.n_verts, np.float32, 3),
            "tris": arr(d.tri_indices, d.n_tris, np.uint32, 3),
            "mesh_items": arr(d.mesh_items, d.n_tris, np.uint32),
            "mesh_nodes": arr(d.mesh_nodes, d.n_mesh_nodes, node_dt),
            "mesh_leaves": arr(d.mesh_leaves, d.n_mesh_leaves, leaf_dt),
            "top_nodes": arr(d.top_nodes, d.n_top_nodes, node_dt),
            "top_leaves": arr(d.top_leaves, d.n_top_leaves, leaf_dt),
            "top_items": arr(d.top_items, d.n_top_items, np.uint32),
            "top_root": d.top_root,
            "meshes": [d.meshes[i] for i in range(d.n_meshes)],
            "spheres": [d.spheres[i] for i in range(d.n_spheres)],
            "squares": [d.squares[i] for i in range(d.n_squares)],
            "lights": [d.lights[i] for i in range(d.n_lights)],
            "materials": [d.materials[i] for i in range(d.n_materials)],
            "textures": [np.ctypeslib.as_array(d.textures[i].texels, shape=(d.textures[i].height, d.textures[i].width,
                                                                             d.textures[i].channels)).copy()
                         for i in range(d.n_textures)],
        }
        return out


def load_off(path):
    L = lib()
    nv, nt = C.c_int64(), C.c_int64()
    rc = L.rth_load_off(path.encode(), None, 0, None, 0, C.byref(nv), C.byref(nt))
    if rc != RT_OK:
        raise RtError(f"cannot read {path}")
    v = np.zeros((nv.value, 3), np.float32)
    t = np.zeros((nt.value, 3), np.uint32)
    L.rth_load_off(path.encode(), _ptr(v), nv.value, _ptr(t), nt.value, C.byref(nv), C.byref(nt))
    return v, t


class Context:
    """A render context (rt_ctx) on one GPU, or on several (devices=[0, 1, ...]: tiles dealt inside the library)."""

    def __init__(self, device=0, devices=None):
        self.L = lib()
        self.h = C.c_void_p()
        devs = [device] if devices is None else list(devices)
        ids = (C.c_int * len(devs))(*devs)
        rc = self.L.rt_init(ids, len(devs), C.byref(self.h))
        if rc != RT_OK:
            raise RtError(f"rt_init(devices={devs}) failed ({rc}): no MI355X visible? (no CPU fallback)")
        self.params = None

    def close(self):
        if self.h:
            self.L.rt_shutdown(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != RT_OK:
            raise RtError(f"{what} failed ({rc}): {self.L.rt_last_error(self.h).decode()}")

    def upload_scene(self, scene):
        self._scene = scene
        self._check(self.L.rt_upload_scene(self.h, scene.desc), "rt_upload_scene")

    def set_camera(self, cam):
        self._check(self.L.rt_set_camera(self.h, C.byref(cam)), "rt_set_camera")

    def begin_frame(self, params):
        self.params = params
        self._check(self.L.rt_begin_frame(self.h, C.byref(params)), "rt_begin_frame")

    def begin_frame_samples(self, params, first_sample, n_samples):
        """Sample sharding: this context renders samples first_sample .. first_sample + n_samples - 1 of a frame of params.spp."""
        self.params = params
        self._check(self.L.rt_begin_frame_samples(self.h, C.byref(params), int(first_sample), int(n_samples)), "rt_begin_frame_samples")

    def export_accumulators(self, acc_ptr, flags_ptr):
        """The frame's fixed-point sums (int64, 3 per pixel) and flag words (uint32 per pixel) into device buffers of the caller."""
        self._check(self.L.rt_export_accumulators(self.h, C.c_void_p(acc_ptr), C.c_void_p(flags_ptr)), "rt_export_accumulators")

    def import_accumulators(self, acc_ptr, flags_ptr):
        self._check(self.L.rt_import_accumulators(self.h, C.c_void_p(acc_ptr), C.c_void_p(flags_ptr)), "rt_import_accumulators")

    def continue_frame(self, more_spp):
        """Progressive accumulation: the tiles queued from here on add more_spp samples per pixel to the open frame."""
        self._check(self.L.rt_continue_frame(self.h, int(more_spp)), "rt_continue_frame")

    def render_tile(self, x0, y0, w, h):
        self._check(self.L.rt_render_tile(self.h, x0, y0, w, h), "rt_render_tile")

    def render_tiles(self, tiles):
        """Queues many tiles with one call; tiles: sequence of (x0, y0, w, h) or an int32 array of shape (n, 4)."""
        a = np.ascontiguousarray(tiles, np.int32).reshape(-1, 4)
        self._check(self.L.rt_render_tiles(self.h, _ptr(a), len(a)), "rt_render_tiles")

    def flush(self):
        self._check(self.L.rt_flush(self.h), "rt_flush")

    def stats(self):
        s = Stats()
        self._check(self.L.rt_get_stats(self.h, C.byref(s)), "rt_get_stats")
        return s

    def download(self):
        p = self.params
        rgb = np.zeros((p.img_h, p.img_w, 3), np.float32)
        prim = np.zeros((p.img_h, p.img_w, 2), np.uint32)
        st = Stats()
        self._check(self.L.rt_download_framebuffer(self.h, _ptr(rgb), _ptr(prim), C.byref(st)), "rt_download_framebuffer")
        return rgb, prim, st

    def resolve_device(self, device_ptr):
        self._check(self.L.rt_resolve_device(self.h, C.c_void_p(device_ptr)), "rt_resolve_device")

    def resolve_rgba8_device(self, device_ptr, bottom_up=False):
        """The displayable frame (the PPM's bytes as RGBA8) into device memory of the caller, e.g. a mapped GL buffer."""
        self._check(self.L.rt_resolve_rgba8_device(self.h, C.c_void_p(device_ptr), 1 if bottom_up else 0), "rt_resolve_rgba8_device")

    def render(self, cam, params, tiles=None):
        """Whole frame (or the given tiles): begin_frame, render_tile*, download."""
        self.set_camera(cam)
        self.begin_frame(params)
        if tiles is None:
            tiles = interleaved_tiles(params.img_w, params.img_h)
        self.render_tiles(tiles)
        return self.download()

    def intersect(self, org, dirs):
        org = np.ascontiguousarray(org, np.float32)
        dirs = np.ascontiguousarray(dirs, np.float32)
        out = np.zeros(len(org), HIT_DTYPE)
        self._check(self.L.rt_intersect(self.h, _ptr(org), _ptr(dirs), len(org), _ptr(out)), "rt_intersect")
        return out

    def occluded(self, pts, lights):
        pts = np.ascontiguousarray(pts, np.float32)
        lights = np.ascontiguousarray(lights, np.float32)
        out = np.zeros(len(pts), np.int32)
        self._check(self.L.rt_occluded(self.h, _ptr(pts), _ptr(lights), len(pts), _ptr(out)), "rt_occluded")
        return out

    def group_clear(self, pts, centres, radii):
        """1 where every shadow ray from pts[i] to the sphere (centres[i], radii[i]) is proven unoccluded by the group test (rt_group_clear)."""
        pts = np.ascontiguousarray(pts, np.float32)
        centres = np.ascontiguousarray(centres, np.float32)
        radii = np.ascontiguousarray(radii, np.float32)
        out = np.zeros(len(pts), np.int32)
        self._check(self.L.rt_group_clear(self.h, _ptr(pts), _ptr(centres), _ptr(radii), len(pts), _ptr(out)), "rt_group_clear")
        return out

    def trace(self, org, dirs, params, pixel=None, sample=None, static=False):
        """Scene::rayTrace on explicit rays; static=True goes through rt_trace_static (no allocation, small inputs)."""
        org = np.ascontiguousarray(org, np.float32)
        dirs = np.ascontiguousarray(dirs, np.float32)
        n = len(org)
        pixel = np.arange(n, dtype=np.uint32) if pixel is None else np.ascontiguousarray(pixel, np.uint32)
        sample = np.zeros(n, np.uint32) if sample is None else np.ascontiguousarray(sample, np.uint32)
        rgb = np.zeros((n, 3), np.float32)
        fn = self.L.rt_trace_static if static else self.L.rt_trace
        self._check(fn(self.h, _ptr(org), _ptr(dirs), _ptr(pixel), _ptr(sample), n, C.byref(params), _ptr(rgb)),
                    "rt_trace_static" if static else "rt_trace")
        return rgb

    def camera_rays(self, params, x0, y0, w, h):
        n = w * h * params.spp
        org = np.zeros((n, 3), np.float32)
        dirs = np.zeros((n, 3), np.float32)
        self._check(self.L.rt_camera_rays(self.h, C.byref(params), x0, y0, w, h, _ptr(org), _ptr(dirs)), "rt_camera_rays")
        return org, dirs

    def ray_trace_from_camera(self, scene, cam, params, ppm_path=None):
        """Drop-in for main.cpp's ray_trace_from_camera(): renders and writes the P3 file."""
        secs = C.c_double()
        self.params = params
        rc = self.L.rth_ray_trace_from_camera(self.h, scene.h, C.byref(cam), C.byref(params),
                                              ppm_path.encode() if ppm_path else None, C.byref(secs))
        self._check(rc, "rth_ray_trace_from_camera")
        return secs.value


def rng(seed, pixel, sample, code, light, i, which):
    return lib().rt_rng(seed, pixel, sample, code, light, i, which)


def write_ppm(path, rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w = rgb.shape[:2]
    rc = lib().rth_write_ppm(path.encode(), _ptr(rgb), w, h)
    if rc != RT_OK:
        raise RtError(f"cannot write {path}")
