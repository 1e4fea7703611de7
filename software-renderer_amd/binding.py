"""ctypes binding of include/swr.h (plumbing for tests / bench; the product is the .so)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# SWR_LIBRARY: tools/ only — load another build of the same C-ABI (e.g. lib/libswr_hip_ablation.so of `make ablation`)
_LIB = os.environ.get("SWR_LIBRARY") or os.path.join(_PKG, "lib", "libswr_hip.so")

FLAG_DEPTH_TEST = 1
FLAG_NO_COLOR = 2
FLAG_METAL_RULES = 4
FLAG_REAL_LINES = 8      # .line primitives: the reference's DDA (Renderer.swift:405-419) instead of its empty stub (:289-293)

# every symbol include/swr.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "swr_abi_version", "swr_version", "swr_context_create", "swr_context_destroy", "swr_last_error",
    "swr_render", "swr_scene_upload", "swr_target_set", "swr_draw", "swr_draw_primitives", "swr_sync", "swr_read_color",
    "swr_read_depth", "swr_timing_enable", "swr_get_timings", "swr_timing_totals", "swr_timing_reset", "swr_pipeline_enable", "swr_tile_rows", "swr_tile_cols",
    "swr_band_rows", "swr_scene_attributes", "swr_material_set", "swr_texture_upload",
    "swr_timing_sample", "swr_context_bands", "swr_context_band_info", "swr_host_alloc", "swr_host_free",
    "swr_host_register", "swr_host_unregister", "swr_present", "swr_present_wait", "swr_device_count",
    "swr_render_timings", "swr_debug_fault", "swr_debug_set",
]
# swr_debug_set keys (test hooks, include/swr.h)
DEBUG_STREAM_ORDER, DEBUG_CULL, DEBUG_BIN_MODE, DEBUG_ONESHOT_MIN_TRIS, DEBUG_DEPTH_KEYS32, DEBUG_RASTER_SORT = 1, 2, 3, 4, 5, 6
BIN_MODE_AUTO, BIN_MODE_EXACT, BIN_MODE_FIXED, BIN_MODE_ATOMIC = 0, 1, 2, 3


class SwrError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"swr error {code}: {msg}")
        self.code = code


class RenderPass(ctypes.Structure):
    """swr_render_pass (include/swr.h) == RenderPass (Renderer.swift:191-200)."""
    _fields_ = [
        ("color", ctypes.c_void_p), ("depth", ctypes.c_void_p),
        ("width", ctypes.c_int64), ("height", ctypes.c_int64),
        ("color_bytes_per_row", ctypes.c_int64), ("depth_bytes_per_row", ctypes.c_int64),
        ("vertices", ctypes.c_void_p), ("vertex_count", ctypes.c_int64),
        ("indices", ctypes.c_void_p), ("index_count", ctypes.c_int64),
        ("primitive_type", ctypes.c_int32), ("flags", ctypes.c_uint32),
        ("transform", ctypes.c_float * 16),
        ("attributes", ctypes.c_void_p), ("material", ctypes.c_void_p), ("texture", ctypes.c_void_p),
        ("tex_width", ctypes.c_int32), ("tex_height", ctypes.c_int32),
        ("scene_id", ctypes.c_uint64),        # ABI 4: non-zero = "same arrays as the last pass with this id": no upload
    ]


class Material(ctypes.Structure):
    """swr_material (include/swr.h): the extended fragment stage."""
    _fields_ = [("shader", ctypes.c_int32), ("shininess_log2", ctypes.c_int32),
                ("light_dir", ctypes.c_float * 4), ("half_dir", ctypes.c_float * 4),
                ("ambient", ctypes.c_float), ("diffuse", ctypes.c_float), ("specular", ctypes.c_float),
                ("reserved", ctypes.c_float)]

    @classmethod
    def from_shading(cls, sh):
        return cls(int(sh.shader), int(sh.shininess_log2),
                   (ctypes.c_float * 4)(*[float(x) for x in sh.light_dir], 0.0),
                   (ctypes.c_float * 4)(*[float(x) for x in sh.half_dir], 0.0),
                   float(sh.ambient), float(sh.diffuse), float(sh.specular), 0.0)


class Config(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("device_count", ctypes.c_uint32),
                ("wait_budget_ms", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class RenderTimes(ctypes.Structure):
    """swr_render_times: wall-clock phases of the last swr_render."""
    _fields_ = [("h2d_ms", ctypes.c_float), ("stream_build_ms", ctypes.c_float), ("draw_ms", ctypes.c_float),
                ("gather_ms", ctypes.c_float), ("total_ms", ctypes.c_float), ("scene_cached", ctypes.c_int32),
                ("frames", ctypes.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}

FAULT_NONE, FAULT_LOST_EVENT, FAULT_ENQUEUE = 0, 1, 2


class Timings(ctypes.Structure):
    _fields_ = [("setup_bin_ms", ctypes.c_float), ("scan_ms", ctypes.c_float),
                ("scatter_ms", ctypes.c_float), ("raster_ms", ctypes.c_float),
                ("total_ms", ctypes.c_float), ("tile_pairs", ctypes.c_int64),
                ("tiles", ctypes.c_int64), ("triangles", ctypes.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def library_path() -> str:
    return _LIB


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of the in-tree library (cross-compiles without a GPU)."""
    srcs = [os.path.join(_PKG, "csrc", f) for f in os.listdir(os.path.join(_PKG, "csrc"))]
    srcs.append(os.path.join(_PKG, "..", "include", "swr.h"))
    stale = force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _PKG, "-s", "lib/libswr_hip.so"])
    return _LIB


_lib = None


def load_library():
    """dlopen the product library; raises if it is not built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise SwrError(-4, f"{_LIB} is not built; run `make -C software-renderer_amd` (hipcc, gfx950)")
    L = ctypes.CDLL(_LIB)
    vp, i64, i32, u32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_uint32
    L.swr_abi_version.restype = ctypes.c_int
    L.swr_version.restype = ctypes.c_char_p
    L.swr_last_error.restype = ctypes.c_char_p
    L.swr_last_error.argtypes = [vp]
    L.swr_context_create.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(vp)]
    L.swr_context_destroy.argtypes = [vp]
    L.swr_context_destroy.restype = None
    L.swr_render.argtypes = [vp, ctypes.POINTER(RenderPass)]
    L.swr_scene_upload.argtypes = [vp, vp, i64, vp, i64]
    L.swr_target_set.argtypes = [vp, i64, i64, i64, i64]
    L.swr_draw.argtypes = [vp, vp, u32]
    L.swr_draw_primitives.argtypes = [vp, vp, u32, i32]
    L.swr_draw_primitives.restype = ctypes.c_int
    L.swr_sync.argtypes = [vp]
    L.swr_read_color.argtypes = [vp, vp]
    L.swr_read_depth.argtypes = [vp, vp]
    L.swr_timing_enable.argtypes = [vp, ctypes.c_int]
    L.swr_get_timings.argtypes = [vp, ctypes.POINTER(Timings)]
    L.swr_timing_totals.argtypes = [vp, ctypes.POINTER(Timings), ctypes.POINTER(i64)]
    L.swr_timing_reset.argtypes = [vp]
    L.swr_timing_sample.argtypes = [vp, ctypes.c_int]
    L.swr_timing_sample.restype = ctypes.c_int
    L.swr_pipeline_enable.argtypes = [vp, ctypes.c_int]
    L.swr_pipeline_enable.restype = ctypes.c_int
    L.swr_band_rows.argtypes = [i64, i32, i32, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    L.swr_scene_attributes.argtypes = [vp, vp, i64]
    L.swr_material_set.argtypes = [vp, ctypes.POINTER(Material)]
    L.swr_texture_upload.argtypes = [vp, vp, i32, i32]
    L.swr_context_bands.argtypes = [vp]
    L.swr_context_band_info.argtypes = [vp, i32, ctypes.POINTER(i32), ctypes.POINTER(i64), ctypes.POINTER(i64)]
    L.swr_host_alloc.argtypes = [ctypes.c_size_t]
    L.swr_host_alloc.restype = vp
    L.swr_host_free.argtypes = [vp]
    L.swr_host_free.restype = None
    L.swr_host_register.argtypes = [vp, ctypes.c_size_t]
    L.swr_host_unregister.argtypes = [vp]
    try:
        L.swr_render_timings.argtypes = [vp, ctypes.POINTER(RenderTimes)]
        L.swr_render_timings.restype = ctypes.c_int
        L.swr_debug_fault.argtypes = [vp, ctypes.c_int]
        L.swr_debug_fault.restype = ctypes.c_int
        L.swr_debug_set.argtypes = [vp, ctypes.c_int, ctypes.c_int64]
        L.swr_debug_set.restype = ctypes.c_int
    except AttributeError:
        if not os.environ.get("SWR_LIBRARY"):      # only an older A/B build loaded by tools/ may lack the ABI 4 entry points
            raise
    L.swr_present.argtypes = [vp, vp, vp]
    L.swr_present_wait.argtypes = [vp]
    for name in ("swr_context_bands", "swr_context_band_info", "swr_host_register", "swr_host_unregister", "swr_present",
                 "swr_present_wait"):
        getattr(L, name).restype = ctypes.c_int
    for name in ("swr_scene_attributes", "swr_material_set", "swr_texture_upload"):
        getattr(L, name).restype = ctypes.c_int
    for name in ("swr_context_create", "swr_render", "swr_scene_upload", "swr_target_set", "swr_draw",
                 "swr_sync", "swr_read_color", "swr_read_depth", "swr_timing_enable", "swr_get_timings", "swr_timing_totals", "swr_timing_reset",
                 "swr_band_rows", "swr_tile_rows", "swr_tile_cols"):
        getattr(L, name).restype = ctypes.c_int
    _lib = L
    return L


def device_count() -> int:
    """Visible HIP devices (0 without a GPU / driver)."""
    L = load_library()
    L.swr_device_count.restype = ctypes.c_int
    return int(L.swr_device_count())


def tile_shape():
    L = load_library()
    return L.swr_tile_cols(), L.swr_tile_rows()


def band_rows(height: int, parts: int, part: int):
    L = load_library()
    a, b = ctypes.c_int64(), ctypes.c_int64()
    rc = L.swr_band_rows(height, parts, part, ctypes.byref(a), ctypes.byref(b))
    if rc:
        raise SwrError(rc, "swr_band_rows: bad arguments")
    return a.value, b.value


# status codes of include/swr.h that the binding itself looks at
ERR_BAD_ARG, ERR_NO_SCENE = -1, -6


class HostImage:
    """A page-locked host image from swr_host_alloc (what a .storageModeShared MTLBuffer is to the reference,
    App.swift:59-60), viewed as a NumPy array; every GPU copies its band straight into it (swr_present)."""

    def __init__(self, shape, dtype):
        self._L = load_library()
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = self._L.swr_host_alloc(self.nbytes)
        if not self.ptr:
            raise SwrError(-7, f"swr_host_alloc({self.nbytes}) failed")
        buf = (ctypes.c_uint8 * self.nbytes).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=self.dtype).reshape(self.shape)
        self._busy = set()          # contexts with a present into this image in flight

    def free(self):
        if self.ptr and self._busy:
            raise SwrError(-1, "HostImage.free() while a swr_present into it is in flight: call present_wait() first")
        if self.ptr:
            self.array = None
            self._L.swr_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def host_register(a: np.ndarray):
    rc = load_library().swr_host_register(a.ctypes.data, a.nbytes)
    if rc:
        raise SwrError(rc, "swr_host_register failed")


def host_unregister(a: np.ndarray):
    load_library().swr_host_unregister(a.ctypes.data)


class Context:
    """swr_context: one per caller thread (GpuRenderer instance, App.swift:149); device_count > 1 = one context
    driving that many tile-row bands on as many GPUs as are visible."""

    def __init__(self, device: int = -1, device_count: int = 0, wait_budget_ms: int = 0):
        self._L = load_library()
        self._h = ctypes.c_void_p()
        self._present_refs = {}         # id -> destination of the presents in flight: kept alive until present_wait / close
        cfg = Config(device, device_count, wait_budget_ms, 0)
        rc = self._L.swr_context_create(ctypes.byref(cfg), ctypes.byref(self._h))
        if rc:
            raise SwrError(rc, (self._L.swr_last_error(None) or b"").decode())
        self.width = self.height = 0
        self.row_begin = self.row_end = 0

    def close(self):
        if self._h:
            self._L.swr_context_destroy(self._h)      # (waits for, or abandons, every copy in flight)
            self._h = ctypes.c_void_p()
        self._release_presents()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    _m_src = None

    def _check(self, rc: int):
        if rc:
            raise SwrError(rc, (self._L.swr_last_error(self._h) or b"").decode())

    # -- resident path -------------------------------------------------------------------
    def scene_upload(self, vertices: np.ndarray, indices: np.ndarray):
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 8)
        i = np.ascontiguousarray(indices, dtype=np.int64).reshape(-1)
        self._check(self._L.swr_scene_upload(self._h, v.ctypes.data, v.shape[0], i.ctypes.data, i.size))

    def scene_attributes(self, attrs: np.ndarray):
        a = np.ascontiguousarray(attrs, dtype=np.float32).reshape(-1, 8)
        self._check(self._L.swr_scene_attributes(self._h, a.ctypes.data, a.shape[0]))

    def material_set(self, material: "Material | None"):
        self._check(self._L.swr_material_set(self._h, ctypes.byref(material) if material is not None else None))

    def texture_upload(self, texture: np.ndarray):
        t = np.ascontiguousarray(texture, dtype=np.uint8)
        assert t.ndim == 3 and t.shape[2] == 4
        self._check(self._L.swr_texture_upload(self._h, t.ctypes.data, t.shape[1], t.shape[0]))

    def shading_set(self, shading):
        """attributes + material + texture of a scenes.Shading (None: back to the reference's stage)."""
        if shading is None:
            self.material_set(None)
            return
        self.scene_attributes(shading.attrs)
        if shading.texture is not None:
            self.texture_upload(shading.texture)
        self.material_set(Material.from_shading(shading))

    def target_set(self, width: int, height: int, row_begin: int = 0, row_end: int | None = None):
        if row_end is None:
            row_end = height
        self._check(self._L.swr_target_set(self._h, width, height, row_begin, row_end))
        self.width, self.height, self.row_begin, self.row_end = width, height, row_begin, row_end

    def draw(self, transform: np.ndarray, flags: int = 0, primitive_type: int = 0):
        # the frame loop calls this thousands of times a second: keep the host side of a draw to one ctypes call
        # (converting the matrix costs more than the call; the same array object is not converted twice)
        # (a float32 C-contiguous array is passed by its own buffer, so the pointer of the same array object can be
        # re-used — in-place updates of that array are seen; anything else is converted on every call)
        if transform is not self._m_src:
            if (isinstance(transform, np.ndarray) and transform.dtype == np.float32 and transform.size == 16
                    and transform.flags.c_contiguous):
                self._m_src, self._m = transform, transform
            else:
                self._m_src, self._m = None, np.ascontiguousarray(transform, dtype=np.float32).reshape(16)
            self._m_ptr = self._m.ctypes.data
        rc = self._L.swr_draw_primitives(self._h, self._m_ptr, flags, primitive_type)
        if rc:
            self._check(rc)

    def sync(self):
        self._check(self._L.swr_sync(self._h))

    def bands(self):
        """[(device, row_begin, row_end)] of every band of the context."""
        out = []
        for k in range(self._L.swr_context_bands(self._h)):
            d, a, b = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64()
            self._check(self._L.swr_context_band_info(self._h, k, ctypes.byref(d), ctypes.byref(a), ctypes.byref(b)))
            out.append((d.value, a.value, b.value))
        return out

    def present(self, color=None, depth=None):
        """Enqueue the async copy of the last drawn frame into the caller's full-size images (NumPy arrays or
        HostImages; page-locked ones make the call non-blocking).  present_wait() makes the pixels visible."""
        def ptr(x):
            if x is None:
                return None
            return x.ptr if isinstance(x, HostImage) else x.ctypes.data
        # the C side writes into these later (DMA, or the helper thread's staged memcpy): keep them alive until the copies
        # have landed (a HostImage refuses to be freed meanwhile).  Registered before the call — a copy may be in flight the
        # moment it returns — and taken back if the call enqueued nothing; one entry per image however many frames present it.
        added = [x for x in (color, depth) if x is not None and id(x) not in self._present_refs]
        for x in added:
            self._present_refs[id(x)] = x
            if isinstance(x, HostImage):
                x._busy.add(id(self))
        rc = self._L.swr_present(self._h, ptr(color), ptr(depth))
        if rc:
            if rc in (ERR_BAD_ARG, ERR_NO_SCENE):          # refused before anything was enqueued
                for x in added:
                    del self._present_refs[id(x)]
                    if isinstance(x, HostImage):
                        x._busy.discard(id(self))
            self._check(rc)

    def _release_presents(self):
        for x in self._present_refs.values():
            if isinstance(x, HostImage):
                x._busy.discard(id(self))
        self._present_refs = {}

    def present_wait(self):
        try:
            self._check(self._L.swr_present_wait(self._h))
        finally:
            self._release_presents()

    def render_timings(self) -> dict:
        t = RenderTimes()
        self._check(self._L.swr_render_timings(self._h, ctypes.byref(t)))
        return t.as_dict()

    def debug_set(self, key: int, value: int):
        """Test hooks (swr_debug_set): force a code path for this context; results never depend on them."""
        self._check(self._L.swr_debug_set(self._h, int(key), ctypes.c_int64(int(value))))

    def debug_fault(self, fault: int):
        """Fault injection for the failure-path tests (FAULT_LOST_EVENT / FAULT_ENQUEUE)."""
        self._check(self._L.swr_debug_fault(self._h, int(fault)))

    def read_color(self, out: np.ndarray | None = None) -> np.ndarray:
        if out is None:
            out = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        assert out.flags.c_contiguous and out.nbytes == self.width * self.height * 4
        self._check(self._L.swr_read_color(self._h, out.ctypes.data))
        return out

    def read_depth(self, out: np.ndarray | None = None) -> np.ndarray:
        if out is None:
            out = np.zeros((self.height, self.width), dtype=np.float32)
        assert out.flags.c_contiguous and out.nbytes == self.width * self.height * 4
        self._check(self._L.swr_read_depth(self._h, out.ctypes.data))
        return out

    def timing_enable(self, level=2):
        """0/False off, 1 = events around k_raster only, 2/True = around every stage."""
        level = 2 if level is True else (0 if level is False else int(level))
        self._check(self._L.swr_timing_enable(self._h, level))

    def timing_sample(self, every_nth: int):
        """Level-1 timing brackets only every n-th frame's k_raster."""
        self._check(self._L.swr_timing_sample(self._h, int(every_nth)))

    def timings(self) -> dict:
        t = Timings()
        self._check(self._L.swr_get_timings(self._h, ctypes.byref(t)))
        return t.as_dict()

    def timing_totals(self):
        """(sums dict, frames) over every frame since timing_reset()."""
        t, n = Timings(), ctypes.c_int64()
        self._check(self._L.swr_timing_totals(self._h, ctypes.byref(t), ctypes.byref(n)))
        return t.as_dict(), n.value

    def pipeline_enable(self, on: bool = True):
        """Overlap binning of the next frame with the raster of the previous one (default on)."""
        self._check(self._L.swr_pipeline_enable(self._h, 1 if on else 0))

    def timing_reset(self):
        self._check(self._L.swr_timing_reset(self._h))

    # -- one-shot path: Renderer.render(renderPass:) / GpuRenderer.render(renderPass:) -----
    def render(self, vertices, indices, transform, width, height, flags=0, primitive_type=0,
               color=None, depth=None, shading=None, scene_id=0):
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 8)
        i = np.ascontiguousarray(indices, dtype=np.int64).reshape(-1)
        m = np.ascontiguousarray(transform, dtype=np.float32).reshape(16)
        if color is None and not (flags & FLAG_NO_COLOR):
            color = np.full((height, width, 4), 0xCD, dtype=np.uint8)
        if depth is None:
            depth = np.full((height, width), -123.0, dtype=np.float32)
        rp = RenderPass()
        rp.color = color.ctypes.data if color is not None else None
        rp.depth = depth.ctypes.data
        rp.width, rp.height = width, height
        rp.color_bytes_per_row = width * 4
        rp.depth_bytes_per_row = width * 4
        rp.vertices, rp.vertex_count = v.ctypes.data, v.shape[0]
        rp.indices, rp.index_count = i.ctypes.data, i.size
        rp.primitive_type, rp.flags = primitive_type, flags
        rp.scene_id = int(scene_id)
        rp.transform = (ctypes.c_float * 16)(*m.tolist())
        if shading is not None:
            a = np.ascontiguousarray(shading.attrs, dtype=np.float32).reshape(-1, 8)
            mat = Material.from_shading(shading)
            rp.attributes = a.ctypes.data
            rp.material = ctypes.addressof(mat)
            if shading.texture is not None:
                t = np.ascontiguousarray(shading.texture, dtype=np.uint8)
                rp.texture, rp.tex_width, rp.tex_height = t.ctypes.data, t.shape[1], t.shape[0]
        self._check(self._L.swr_render(self._h, ctypes.byref(rp)))
        return color, depth


def render(scene, extra_flags: int = 0, device: int = -1):
    """Convenience: one-shot render of a scenes.Scene; returns (color, depth)."""
    with Context(device) as ctx:
        return ctx.render(scene.vertices, scene.indices, scene.transform, scene.width, scene.height,
                          scene.flags | extra_flags, shading=scene.shading)
