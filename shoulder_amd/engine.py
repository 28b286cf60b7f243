"""Engine: one HIP context (device + stream) behind the C-ABI, NumPy in / NumPy out.

Host-side glue only -- every numeric stage runs in libshoulder_hip.so.  A missing library or a
failing call raises; nothing falls back to the CPU.
"""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import LANDMARKS_DTYPE


class ShoulderHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libshoulder_hip error {code}: {msg}")
        self.code = code


_MODELS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")

UNET_ORDER = None


def unet_pack_order(depth):
    names = []
    for i in range(depth):
        names += [f"enc{i}a_w", f"enc{i}a_b", f"enc{i}b_w", f"enc{i}b_b"]
    names += ["bota_w", "bota_b", "botb_w", "botb_b"]
    for i in reversed(range(depth)):
        names += [f"up{i}_w", f"up{i}_b", f"dec{i}a_w", f"dec{i}a_b", f"dec{i}b_w", f"dec{i}b_b"]
    names += ["head_w", "head_b"]
    return names


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def _mat4(T):
    """a C-contiguous float64 4x4 (the C side reads 16 doubles); anything else is the reference's ValueError (base.py:57-58)"""
    T = np.ascontiguousarray(T, dtype=np.float64)
    if T.shape != (4, 4):
        raise ValueError("Invalid transformation matrix shape")
    return T


class Engine:
    def __init__(self, device=0, stream=None):
        self.L = _lib.load()
        h = ctypes.c_void_p()
        rc = self.L.sh_ctx_create(int(device), ctypes.c_void_p(stream) if stream else None, ctypes.byref(h))
        if rc != 0:
            raise ShoulderHipError(rc, f"sh_ctx_create(device={device}) failed (is a HIP device visible?)")
        self.h = h
        self.device = device
        self._inflight, self._pin_slot = [], 0

    def close(self):
        if getattr(self, "h", None):
            self._free_pinned()
            self.L.sh_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise ShoulderHipError(rc, self.L.sh_last_error(self.h).decode())

    # ---- parameters ----------------------------------------------------------------------------
    def load_rfc(self, npz_path=None):
        z = np.load(npz_path or os.path.join(_MODELS, "rfc_bg3.npz"))
        c = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
        feat, thr = c(z["feat"], np.int32), c(z["thr"], np.float32)
        ti, fi = c(z["true_idx"], np.int32), c(z["false_idx"], np.int32)
        lw, roots = c(z["leaf_weight"], np.float32), c(z["roots"], np.int32)
        self._chk(self.L.sh_load_rfc(self.h, _ptr(feat), _ptr(thr), _ptr(ti), _ptr(fi), _ptr(lw), len(feat), _ptr(roots), len(roots)))

    def load_unet(self, weights, base, depth):
        packed = np.concatenate([np.asarray(weights[k], dtype=np.float32).ravel() for k in unet_pack_order(depth)])
        packed = np.ascontiguousarray(packed)
        self._chk(self.L.sh_load_unet(self.h, int(base), int(depth), _ptr(packed), packed.size))

    def load_unet_onnx(self, path_or_bytes):
        """Parameters of the anatomic-neck network from an ONNX file of the supported UNet family (onnx_import.py; the
        reference opens `humerus/models/unetcrf_anp.onnx` at anatomic_neck.py:62-69).  Returns (base, depth)."""
        from .onnx_import import unet_from_onnx
        w, base, depth = unet_from_onnx(path_or_bytes)
        self.load_unet(w, base, depth)
        return base, depth

    def set_params(self, canal_cutoff=None, groove_cutoff=None, groove_deg_window=None, unet_dtype=None, bone_kind=None):
        """Read-modify-write of sh_params: fields left at None keep the value in force (an engine configured for a bf16
        UNet stays so when a facade object only sets its bone kind)."""
        p = _lib.Params()
        self._chk(self.L.sh_get_params(self.h, ctypes.byref(p)))
        if canal_cutoff is not None:
            p.canal_cutoff[0], p.canal_cutoff[1] = canal_cutoff
        if groove_cutoff is not None:
            p.groove_cutoff[0], p.groove_cutoff[1] = groove_cutoff
        if groove_deg_window is not None:
            p.groove_deg_window = groove_deg_window
        if unet_dtype is not None:
            p.unet_dtype = unet_dtype
        if bone_kind is not None:
            p.bone_kind = bone_kind
        self._chk(self.L.sh_set_params(self.h, ctypes.byref(p)))

    def get_params(self):
        p = _lib.Params()
        self._chk(self.L.sh_get_params(self.h, ctypes.byref(p)))
        return dict(canal_cutoff=tuple(p.canal_cutoff), groove_cutoff=tuple(p.groove_cutoff), groove_deg_window=p.groove_deg_window,
                    unet_dtype=p.unet_dtype, bone_kind=p.bone_kind)

    def reset_params(self):
        p = _lib.Params()
        self.L.sh_default_params(ctypes.byref(p))
        self._chk(self.L.sh_set_params(self.h, ctypes.byref(p)))

    def param_block(self):
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        self._chk(self.L.sh_param_block(self.h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def param_block_commit(self):
        """The device parameter block was overwritten from outside (a broadcast): refresh the host mirrors (sh_param_block_commit)."""
        self._chk(self.L.sh_param_block_commit(self.h))

    # ---- meshes ----------------------------------------------------------------------------------
    def upload(self, meshes):
        """meshes: list of (verts float32 (V,3), faces int32 (F,3))."""
        if len(meshes) == 0:
            raise ValueError("upload() needs at least one mesh")
        verts, faces, voff, foff = self.pack_meshes(meshes)
        self._chk(self.L.sh_upload_meshes(self.h, _ptr(verts), _ptr(faces), _ptr(voff), _ptr(foff), len(meshes)))
        self.voff, self.foff = voff, foff

    def upload_packed(self, packed):
        """sh_upload_meshes from the 4-tuple of pack_meshes()."""
        verts, faces, voff, foff = packed
        self._chk(self.L.sh_upload_meshes(self.h, _ptr(verts), _ptr(faces), _ptr(voff), _ptr(foff), len(voff) - 1))
        self.voff, self.foff = voff, foff

    def upload_stl(self, files):
        """files: list of paths or bytes objects, each one binary STL; parsed and merged on the device (sh_upload_stl)."""
        import pathlib
        blobs = [f if isinstance(f, (bytes, bytearray)) else pathlib.Path(f).read_bytes() for f in files]
        n = len(blobs)
        keep = [bytes(b) for b in blobs]      # the buffers must outlive the call
        ptrs = (ctypes.c_void_p * n)(*[ctypes.cast(ctypes.c_char_p(k), ctypes.c_void_p) for k in keep])
        sizes = (ctypes.c_size_t * n)(*[len(k) for k in keep])
        voff = np.zeros(n + 1, dtype=np.int64)
        foff = np.zeros(n + 1, dtype=np.int64)
        self._chk(self.L.sh_upload_stl(self.h, ptrs, sizes, n, _ptr(voff), _ptr(foff)))
        self.voff, self.foff = voff, foff

    # ---- the staging side: the NEXT batch is handed over while a run of the resident one executes -----------------------------
    @staticmethod
    def pack_meshes(meshes):
        """list of (verts, faces) -> (verts float32 (sumV,3), faces int32 (sumF,3), voff int64 (B+1), foff int64 (B+1)): the arrays
        sh_upload_meshes / sh_stage_meshes take.  A caller that streams batches packs each one once, outside its hot loop."""
        if len(meshes) == 0:
            raise ValueError("at least one mesh")
        verts = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.float32).reshape(-1, 3) for v, _ in meshes]))
        faces = np.ascontiguousarray(np.concatenate([np.asarray(f, dtype=np.int32).reshape(-1, 3) for _, f in meshes]))
        voff = np.zeros(len(meshes) + 1, dtype=np.int64)
        foff = np.zeros(len(meshes) + 1, dtype=np.int64)
        voff[1:] = np.cumsum([len(v) for v, _ in meshes])
        foff[1:] = np.cumsum([len(f) for _, f in meshes])
        return verts, faces, voff, foff

    def stage(self, meshes):
        """sh_stage_meshes: `meshes` = list of (verts, faces) or the 4-tuple of pack_meshes().  Returns at once; commit_staged()
        makes the batch the resident one."""
        packed = meshes if (isinstance(meshes, tuple) and len(meshes) == 4 and getattr(meshes[2], "dtype", None) == np.int64) else self.pack_meshes(meshes)
        verts, faces, voff, foff = packed
        self._chk(self.L.sh_stage_meshes(self.h, _ptr(verts), _ptr(faces), _ptr(voff), _ptr(foff), len(voff) - 1))
        self._staged_off = (voff, foff)
        self._staged_keep = packed      # the library reads the arrays from a background thread until the commit

    def stage_stl(self, files):
        """sh_stage_stl: list of paths or bytes objects (binary STL); parse + vertex merge on the device, beside the run in flight."""
        import pathlib
        keep = [bytes(f) if isinstance(f, (bytes, bytearray)) else pathlib.Path(f).read_bytes() for f in files]
        n = len(keep)
        ptrs = (ctypes.c_void_p * n)(*[ctypes.cast(ctypes.c_char_p(k), ctypes.c_void_p) for k in keep])
        sizes = (ctypes.c_size_t * n)(*[len(k) for k in keep])
        self._chk(self.L.sh_stage_stl(self.h, ptrs, sizes, n))
        self._staged_off = None
        self._staged_keep = keep        # the library reads the files from a background thread until the commit

    def commit_staged(self):
        """sh_commit_staged: the staged batch becomes the resident one (needs every submitted run collected)."""
        try:
            self._chk(self.L.sh_commit_staged(self.h, None, None))
        finally:
            # the arrays / file images stay referenced while the batch is STILL staged (a refused commit -- runs in flight -- leaves it
            # staged, and the library's background thread may still be copying from them)
            if not self.staged:
                self._staged_keep = None
        B = self.L.sh_batch_size(self.h)
        if getattr(self, "_staged_off", None) is not None:
            self.voff, self.foff = self._staged_off
        else:      # STL: the offsets were made on the device
            self.voff = self.fetch("voff", np.int64, (B + 1,)).copy()
            self.foff = self.fetch("foff", np.int64, (B + 1,)).copy()
        self._staged_off = None

    @property
    def staged(self):
        return self.L.sh_staged(self.h) == 1

    def synth_batch(self, T):
        T = np.ascontiguousarray(T, dtype=np.float64).reshape(-1, 16)
        V, F = int(self.voff[1] - self.voff[0]), int(self.foff[1] - self.foff[0])
        self._chk(self.L.sh_synth_batch(self.h, _ptr(T), len(T)))
        self.voff = np.arange(len(T) + 1, dtype=np.int64) * V
        self.foff = np.arange(len(T) + 1, dtype=np.int64) * F

    @property
    def B(self):
        return self.L.sh_batch_size(self.h)

    def set_keep_products(self, on=True):
        """Every proximal plane's resampled contour and polar rows are written (sh_set_keep_products): for fetch("prox.ixy") etc.;
        off (the default) a run writes the rows its later stages read."""
        self._chk(self.L.sh_set_keep_products(self.h, int(bool(on))))

    # ---- record format ------------------------------------------------------------------------------
    def set_record_rows(self, anp_rows=0):
        """sh_set_record_rows: 0 = full 104 KB records, R > 0 = packed records carrying R anatomic-neck rows (n_anp keeps the
        true count; anp_points(b) fetches every row).  run / submit / collect then return arrays of _lib.record_dtype(R)."""
        self._chk(self.L.sh_set_record_rows(self.h, int(anp_rows)))
        if int(anp_rows) != getattr(self, "_rec_rows", 0):
            self._free_pinned()
        self._rec_rows = int(anp_rows)

    @property
    def record_dtype(self):
        return _lib.record_dtype(getattr(self, "_rec_rows", 0))

    def anp_points(self, b):
        """Every anatomic-neck point (CT) of humerus b of the last run (sh_anp_points)."""
        n = ctypes.c_int()
        self._chk(self.L.sh_anp_points(self.h, int(b), None, 0, ctypes.byref(n)))
        out = np.empty((max(1, n.value), 3), dtype=np.float64)
        self._chk(self.L.sh_anp_points(self.h, int(b), _ptr(out), len(out), ctypes.byref(n)))
        return out[: n.value]

    # ---- run ---------------------------------------------------------------------------------------
    def run(self, stages=_lib.STAGE_ALL, fetch=True):
        """One sh_run over the resident batch.  fetch=True: a fresh record array; fetch="view": the engine's page-locked
        record buffer (no allocation, direct D2H), valid until the next run; fetch=False: records stay on the device."""
        if fetch == "view":
            out = self._pinned_records()
        else:
            out = np.zeros(self.B, dtype=self.record_dtype) if fetch else None
        self._chk(self.L.sh_run(self.h, int(stages), _ptr(out) if out is not None else None))
        return out

    def submit(self, stages=_lib.STAGE_ALL, fetch="view", out_ptr=None):
        """Enqueue one run and return at once (sh_submit); at most two may be in flight.  `collect()` hands back the records of
        the oldest one.  fetch="view": records go to one of the engine's two page-locked buffers; False: they stay on the
        device; out_ptr: raw address of B records of page-locked host or DEVICE memory (e.g. a gather's send buffer)."""
        if fetch not in ("view", False):
            raise ValueError('submit(fetch=...) takes "view" or False')
        out = None
        if out_ptr is not None:
            ptr = ctypes.c_void_p(int(out_ptr))
        elif fetch == "view":
            out = self._pinned_records(self._pin_slot)
            self._pin_slot ^= 1
            ptr = _ptr(out)
        else:
            ptr = None
        self._inflight.append(out)
        try:
            self._chk(self.L.sh_submit(self.h, int(stages), ptr))
        except Exception:
            self._inflight.pop()
            raise

    def collect(self):
        out = self._inflight.pop(0) if self._inflight else None      # (nothing in flight: sh_collect reports it)
        self._chk(self.L.sh_collect(self.h))
        return out

    def _pinned_records(self, slot=0):
        pins = self.__dict__.setdefault("_pins", {})
        ent = pins.get(slot)
        if ent is None or ent[1] < self.B:
            if ent is not None:
                self.L.sh_host_free(self.h, ent[0])
            p = ctypes.c_void_p()
            dt = self.record_dtype
            nbytes = self.B * dt.itemsize
            self._chk(self.L.sh_host_alloc(self.h, nbytes, ctypes.byref(p)))
            arr = np.frombuffer((ctypes.c_char * nbytes).from_address(p.value), dtype=dt)
            ent = pins[slot] = (p, self.B, arr)
        return ent[2][: self.B]

    def _free_pinned(self):
        for p, _, _ in self.__dict__.get("_pins", {}).values():
            self.L.sh_host_free(self.h, p)
        self.__dict__["_pins"] = {}

    def landmarks_device(self):
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        self._chk(self.L.sh_landmarks_device(self.h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def buffer_device(self, name):
        """(device address, bytes) of a named buffer (sh_buffer_device)."""
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        self._chk(self.L.sh_buffer_device(self.h, name.encode(), ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def affine_apply(self, T, dev_in, dev_out, off):
        """utils.transform_pts for B point sets resident on the device (sh_affine_apply): out[off[b]:off[b+1]] = T[b] * in[...];
        dev_in / dev_out are device addresses of float64 xyz arrays, T (B,4,4), off (B+1,) point offsets."""
        T = np.ascontiguousarray(T, dtype=np.float64)
        if T.ndim != 3 or T.shape[1:] != (4, 4):
            raise ValueError("Invalid transformation matrix shape")
        off = np.ascontiguousarray(off, dtype=np.int64)
        if off.shape != (len(T) + 1,):
            raise ValueError("off must hold B + 1 offsets")
        self._chk(self.L.sh_affine_apply(self.h, _ptr(T), ctypes.c_void_p(int(dev_in)), ctypes.c_void_p(int(dev_out)), _ptr(off), len(T)))

    def mesh_transformed(self, b, T):
        T = _mat4(T)
        if not 0 <= int(b) < len(self.voff) - 1:
            raise IndexError("mesh index out of range")
        out = np.empty((int(self.voff[b + 1] - self.voff[b]), 3), dtype=np.float64)
        self._chk(self.L.sh_mesh_transformed(self.h, int(b), _ptr(T), _ptr(out)))
        return out

    def transform_points(self, pts, T):
        """utils.transform_pts on the device: (n,3) float64 host points -> (n,3)."""
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
        T = _mat4(T)
        out = np.empty_like(pts)
        self._chk(self.L.sh_transform_points(self.h, _ptr(T), _ptr(pts), len(pts), _ptr(out)))
        return out

    def ring(self, set_name, b, k):
        """Closed largest loop of plane k of slice set `set_name` ("distal", "prox", "neckc") of humerus b: (n + 1, 2) float64 in
        the box frame (sh_ring; also for planes whose ring lives in the overflow pool)."""
        n = ctypes.c_int()
        self._chk(self.L.sh_ring(self.h, set_name.encode(), int(b), int(k), None, 0, ctypes.byref(n)))
        out = np.empty((max(1, n.value), 2), dtype=np.float64)
        self._chk(self.L.sh_ring(self.h, set_name.encode(), int(b), int(k), _ptr(out), len(out), ctypes.byref(n)))
        return out[: n.value]

    def section_plane(self, b, origin, normal, cap=8192):
        """Unique crossing points (CT) of mesh b with one plane: `mesh_ct.section(...).vertices`."""
        o = np.ascontiguousarray(origin, dtype=np.float64)
        n = np.ascontiguousarray(normal, dtype=np.float64)
        while True:      # (a dense mesh crosses a plane more often than the default capacity: ask again with room)
            out = np.empty((cap, 3), dtype=np.float64)
            k = ctypes.c_int()
            rc = self.L.sh_section_plane(self.h, int(b), _ptr(o), _ptr(n), _ptr(out), cap, ctypes.byref(k))
            if rc == -4 and cap < (1 << 24):
                cap *= 8
                continue
            self._chk(rc)
            return out[:k.value].copy()

    def slice_mesh_planes(self, verts, faces, origins, normals, edges=False):
        """`Trimesh.slice_plane(origin, normal)` of one mesh for P planes in one device pass (sh_slice_mesh_planes;
        arthroplasty.py:80-87).  -> list of (verts (n,3) f64, faces (m,3) i32[, cut edges (k,2) i32]) per plane."""
        v = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 3)
        f = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1, 3)
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        n = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)
        if len(o) != len(n):
            raise ValueError("one origin per normal")
        P = len(o)
        cnt = np.zeros((P, 3), dtype=np.int32)
        self._chk(self.L.sh_slice_mesh_planes(self.h, _ptr(v), len(v), _ptr(f), len(f), _ptr(o), _ptr(n), P, None, 0, None, 0, None, 0, _ptr(cnt)))
        cv, cf, ce = (max(1, int(x)) for x in cnt.max(axis=0))
        ov = np.empty((P, cv, 3), dtype=np.float64)
        of = np.empty((P, cf, 3), dtype=np.int32)
        oe = np.empty((P, ce, 2), dtype=np.int32) if edges else None
        self._chk(self.L.sh_slice_mesh_planes(self.h, _ptr(v), len(v), _ptr(f), len(f), _ptr(o), _ptr(n), P, _ptr(ov), cv, _ptr(of), cf,
                                              _ptr(oe) if edges else None, ce if edges else 0, _ptr(cnt)))
        out = []
        for p in range(P):
            nv_, nf_, ne_ = (int(x) for x in cnt[p])
            out.append((ov[p, :nv_].copy(), of[p, :nf_].copy()) + ((oe[p, :ne_].copy(),) if edges else ()))
        return out

    # ---- named buffers -----------------------------------------------------------------------------
    def fetch(self, name, dtype, shape=None):
        n, e = ctypes.c_size_t(), ctypes.c_int()
        self._chk(self.L.sh_buffer_info(self.h, name.encode(), ctypes.byref(n), ctypes.byref(e)))
        dtype = np.dtype(dtype)
        count = int(np.prod(shape)) if shape is not None else n.value // dtype.itemsize
        out = np.empty(count, dtype=dtype)
        self._chk(self.L.sh_fetch(self.h, name.encode(), _ptr(out), out.nbytes))
        return out.reshape(shape) if shape is not None else out

    def store(self, name, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self.L.sh_store(self.h, name.encode(), _ptr(arr), arr.nbytes))

    def unet_infer(self, images):
        """The anatomic-neck network alone: images (n, H, W) float32 -> logits (n, H, W) float32 (anatomic_neck.py:67-76)."""
        x = np.ascontiguousarray(images, dtype=np.float32)
        if x.ndim != 3:
            raise ValueError("images must have shape (n, H, W)")
        out = np.empty_like(x)
        self._chk(self.L.sh_unet_infer(self.h, _ptr(x), x.shape[0], x.shape[1], x.shape[2], _ptr(out)))
        return out

    # ---- timing --------------------------------------------------------------------------------------
    def enable_timing(self, on=True):
        self._chk(self.L.sh_enable_timing(self.h, int(on)))      # 0 off, 1 every launch, 2 UNet layers only

    def set_hull_mode(self, mode):
        """"host" | "device" | "auto": where the convex hull of the OBB stage is computed (sh_set_hull_mode); same results."""
        self._chk(self.L.sh_set_hull_mode(self.h, str(mode).encode()))

    @property
    def hull_mode(self):
        return "device" if self.L.sh_get_hull_mode(self.h) == 1 else "host"

    def set_overlap(self, on=True):
        """Streaming runs on the resident batch: compute the host hulls of the next run while the device works on this one."""
        self._chk(self.L.sh_set_overlap(self.h, 1 if on else 0))

    def set_unet_turns(self, on=True):
        """Several engines on one device: their UNet passes run one after another, geometry overlaps (sh_set_unet_turns)."""
        self._chk(self.L.sh_set_unet_turns(self.h, 1 if on else 0))

    def discard_prepared(self):
        self._chk(self.L.sh_discard_prepared(self.h))

    def reset_timers(self):
        self._chk(self.L.sh_kernel_time_ms(self.h, None, None, None))

    def kernel_time_ms(self, name):
        ms, n = ctypes.c_double(), ctypes.c_int()
        self._chk(self.L.sh_kernel_time_ms(self.h, name.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value
