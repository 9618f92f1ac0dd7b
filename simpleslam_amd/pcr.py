"""Host-side mirror of the reference's registration plugin over the C ABI.

Reference interface (compiled C++): PCR::PointCloudRegister with
``bool scan2Map(const PC_cPtr& src, const PC_cPtr& dst, pose_t& res)`` and
``scalar_t getFitnessScore()`` (reference PCR/include/PCR/PointCloudRegister.hpp:12-38),
implemented by LoamRegister / NdtRegister / VgicpRegister and selected by the config
key ``frontend.pcr`` (reference frontend/src/LidarOdometry.cpp:44-54).  The classes
below keep those names and meanings; every call goes through libpcr_hip.so
(include/pcr_hip.h) -- there is no CPU fallback: if the HIP library or a GPU is
missing, construction raises.

Clouds are float32 arrays of shape (n, 4) [x y z intensity] or (n, 8)
(pcl::PointXYZI layout), either numpy (host) or torch tensors resident in HBM.
Poses are 4x4 float64 numpy arrays (map <- lidar).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PCR_LIB (read HERE, by the Python loader -- the library itself reads no environment variable): another build of the library to load instead of the
# product, e.g. the development build ab/libdev.so of scripts/build_dev.sh.  The A/B and sweep scripts set it; nothing overwrites lib/libpcr_hip.so.
LIB_PATH = os.environ.get("PCR_LIB") or os.path.join(_HERE, "lib", "libpcr_hip.so")


class PcrParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32),
        ("loam_iters", C.c_int32), ("loam_early_exit", C.c_int32),
        ("loam_knn_max_sq", C.c_double), ("loam_plane_thresh", C.c_double), ("loam_point_thresh", C.c_double),
        ("loam_pos_conv", C.c_double), ("loam_rot_conv", C.c_double),
        ("ndt_resolution", C.c_double), ("ndt_step_size", C.c_double), ("ndt_outlier_ratio", C.c_double),
        ("ndt_trans_eps", C.c_double), ("ndt_max_iters", C.c_int32), ("ndt_min_points", C.c_int32),
        ("vgicp_resolution", C.c_double), ("vgicp_k_corr", C.c_int32), ("vgicp_max_iters", C.c_int32),
        ("vgicp_lm_inner", C.c_int32), ("vgicp_rot_eps", C.c_double), ("vgicp_trans_eps", C.c_double),
        ("vgicp_lm_init_scale", C.c_double),
        ("record_trace", C.c_int32),
        ("index_no_hints", C.c_int32), ("ndt_evaluate_repeats", C.c_int32), ("loam_disable_cache", C.c_int32), ("record_timeline", C.c_int32),
        ("loam_coresident", C.c_int32), ("loam_clamp_margin_mm", C.c_int32), ("full_target", C.c_int32), ("host_optimiser", C.c_int32),
        ("host_copy_xyz", C.c_int32),
    ]


class PcrStats(C.Structure):
    _fields_ = [
        ("total_ms", C.c_double), ("index_ms", C.c_double), ("solve_ms", C.c_double), ("kernel_ms", C.c_double),
        ("kernel_launches", C.c_int32), ("iterations", C.c_int32), ("n_src", C.c_int64), ("n_dst", C.c_int64),
        ("attempts", C.c_int32), ("target_builds", C.c_int32), ("region_repeats", C.c_int32), ("region_index", C.c_int32),
        ("aux_kernel_ms", C.c_double), ("region_points", C.c_int64), ("region_voxels", C.c_int64), ("pairs_grad", C.c_int64), ("pairs_hess", C.c_int64),
        ("index_box_hint", C.c_int32), ("index_layout_hint", C.c_int32),
    ]


# every symbol include/pcr_hip.h declares
ABI_SYMBOLS = [
    "pcr_default_params", "pcr_create", "pcr_destroy", "pcr_last_error", "pcr_scan2map", "pcr_scan2map_device", "pcr_host_pin", "pcr_host_unpin",
    "pcr_set_target", "pcr_align", "pcr_invalidate_target", "pcr_fitness", "pcr_loam_linearize", "pcr_get_trace", "pcr_get_trace_counts",
    "pcr_vgicp_covariances", "pcr_vgicp_neighbours", "pcr_vgicp_linearize", "pcr_voxel_filter", "pcr_voxel_filter_begin", "pcr_voxel_filter_end", "pcr_get_timeline", "pcr_ndt_derivatives", "pcr_get_stats", "pcr_set_profile", "pcr_set_stream", "pcr_set_query_tile", "pcr_comm_unique_id", "pcr_comm_init", "pcr_comm_info", "pcr_comm_peer_export", "pcr_comm_init_peer",
    "pcr_comm_init_host", "pcr_set_shard", "pcr_set_params", "pcr_get_params", "pcr_fitness_gated",
    "pcr_map_create", "pcr_map_destroy", "pcr_map_last_error", "pcr_map_add_keyframe", "pcr_map_keyframes", "pcr_map_clear", "pcr_map_update", "pcr_map_update_begin", "pcr_map_wait", "pcr_map_update_window", "pcr_map_submap",
    "pcr_map_submap_indices", "pcr_map_generation", "pcr_scan2map_submap",
    "pcr_ndt_opt_create", "pcr_ndt_opt_destroy", "pcr_ndt_opt_request", "pcr_ndt_opt_feed", "pcr_ndt_opt_result", "pcr_ndt_opt_counts",
    "pcr_vgicp_opt_create", "pcr_vgicp_opt_destroy", "pcr_vgicp_opt_request", "pcr_vgicp_opt_feed", "pcr_vgicp_opt_result",
    "pcr_sc_default_params", "pcr_sc_create", "pcr_sc_destroy", "pcr_sc_last_error", "pcr_sc_size", "pcr_sc_add", "pcr_sc_descriptor", "pcr_sc_distance",
    "pcr_sc_query",
]

# pcr_allreduce_fn (include/pcr_hip.h): int fn(double* inout, size_t count, int op, void* user)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_size_t, C.c_int, C.c_void_p)


class ScParams(C.Structure):
    """struct pcr_sc_params (include/pcr_hip.h)."""
    _fields_ = [("lidar_height", C.c_double), ("num_exclude_recent", C.c_int32), ("build_tree_gap", C.c_int32), ("num_candidates", C.c_int32),
                ("pad", C.c_int32), ("search_ratio", C.c_double), ("dist_thres", C.c_double)]


_lib = None


def load_library():
    """dlopen libpcr_hip.so and declare the prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C simpleslam_amd/csrc`). "
            "There is no CPU fallback.")
    try:
        # torch ships its own libamdhip64 under the same soname: whichever HIP runtime is mapped first serves the whole
        # process, and torch.cuda reports "No HIP GPUs" when that is not its own.  Map torch's first when it is installed.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
    L.pcr_default_params.argtypes = [C.POINTER(PcrParams)]
    L.pcr_default_params.restype = None
    L.pcr_create.argtypes = [C.c_char_p, C.POINTER(PcrParams)]
    L.pcr_create.restype = vp
    L.pcr_destroy.argtypes = [vp]
    L.pcr_destroy.restype = None
    L.pcr_last_error.argtypes = [vp]
    L.pcr_last_error.restype = C.c_char_p
    L.pcr_scan2map.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, dp, ip]
    L.pcr_scan2map_device.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, dp, ip]
    L.pcr_host_pin.argtypes = [vp, C.c_size_t]
    L.pcr_host_unpin.argtypes = [vp]
    L.pcr_set_target.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_int]
    L.pcr_align.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_int, dp, ip]
    L.pcr_invalidate_target.argtypes = [vp]
    L.pcr_fitness.argtypes = [vp]
    L.pcr_fitness.restype = C.c_double
    L.pcr_loam_linearize.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_int, dp, dp, dp, C.POINTER(C.c_int64), vp, vp, vp]
    L.pcr_get_trace.argtypes = [vp, C.POINTER(C.c_int32), vp, vp, vp, vp]
    L.pcr_get_trace_counts.argtypes = [vp, vp, vp]
    L.pcr_vgicp_covariances.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_int, vp]
    L.pcr_vgicp_neighbours.argtypes = [vp, C.c_size_t, vp, vp]
    L.pcr_vgicp_linearize.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_int, dp, dp, dp, dp, C.POINTER(C.c_int64)]
    L.pcr_ndt_derivatives.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_int, dp, dp, dp, dp, dp]
    L.pcr_get_timeline.argtypes = [vp, vp, C.c_size_t, ip, ip]
    L.pcr_voxel_filter.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_int, C.c_double, vp, C.c_size_t, C.c_int, C.POINTER(C.c_size_t)]
    L.pcr_voxel_filter_begin.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_double, vp, C.c_size_t]
    L.pcr_voxel_filter_end.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.pcr_get_stats.argtypes = [vp, C.POINTER(PcrStats)]
    L.pcr_scan2map_submap.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, dp, ip]
    L.pcr_ndt_opt_create.restype = vp
    L.pcr_ndt_opt_create.argtypes = [dp, C.c_double, C.c_double, C.c_int]
    L.pcr_ndt_opt_destroy.restype = None
    L.pcr_ndt_opt_destroy.argtypes = [vp]
    L.pcr_ndt_opt_request.argtypes = [vp, ip, dp, dp]
    L.pcr_ndt_opt_feed.argtypes = [vp, dp]
    L.pcr_ndt_opt_result.argtypes = [vp, dp, ip, ip, ip]
    L.pcr_ndt_opt_counts.argtypes = [vp, ip, ip, ip]
    L.pcr_vgicp_opt_create.restype = vp
    L.pcr_vgicp_opt_create.argtypes = [dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
    L.pcr_vgicp_opt_destroy.restype = None
    L.pcr_vgicp_opt_destroy.argtypes = [vp]
    L.pcr_vgicp_opt_request.argtypes = [vp, ip, dp, dp]
    L.pcr_vgicp_opt_feed.argtypes = [vp, dp]
    L.pcr_vgicp_opt_result.argtypes = [vp, dp, ip, ip, ip]
    L.pcr_map_generation.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.pcr_set_profile.argtypes = [vp, C.c_int]
    L.pcr_set_stream.argtypes = [vp, vp]
    L.pcr_set_query_tile.argtypes = [vp, dp, dp]
    L.pcr_comm_unique_id.argtypes = [vp]
    L.pcr_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    L.pcr_comm_peer_export.argtypes = [vp, vp]
    L.pcr_comm_init_peer.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    L.pcr_comm_init_host.argtypes = [vp, ALLREDUCE_FN, vp, C.c_int, C.c_int]
    L.pcr_comm_info.argtypes = [vp, ip, ip, ip]
    L.pcr_set_shard.argtypes = [vp, dp, dp, C.c_double]
    L.pcr_set_params.argtypes = [vp, C.POINTER(PcrParams)]
    L.pcr_get_params.argtypes = [vp, C.POINTER(PcrParams)]
    L.pcr_fitness_gated.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_int, dp, C.c_double, dp, C.POINTER(C.c_int64)]
    L.pcr_map_create.argtypes = [C.c_int]
    L.pcr_map_create.restype = vp
    L.pcr_map_destroy.argtypes = [vp]
    L.pcr_map_destroy.restype = None
    L.pcr_map_last_error.argtypes = [vp]
    L.pcr_map_last_error.restype = C.c_char_p
    L.pcr_map_add_keyframe.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_int, dp]
    L.pcr_map_keyframes.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.pcr_map_clear.argtypes = [vp]
    L.pcr_map_update.argtypes = [vp, dp, C.c_double, C.c_double, C.POINTER(C.c_size_t)]
    L.pcr_map_update_begin.argtypes = [vp, dp, C.c_double, C.c_double]
    L.pcr_map_wait.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.pcr_map_update_window.argtypes = [vp, C.c_longlong, C.c_int, C.c_double, C.POINTER(C.c_size_t)]
    L.pcr_map_submap.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.pcr_map_submap.restype = vp
    L.pcr_map_submap_indices.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.pcr_sc_default_params.argtypes = [C.POINTER(ScParams)]
    L.pcr_sc_default_params.restype = None
    L.pcr_sc_create.argtypes = [C.c_int, C.POINTER(ScParams)]
    L.pcr_sc_create.restype = vp
    L.pcr_sc_destroy.argtypes = [vp]
    L.pcr_sc_destroy.restype = None
    L.pcr_sc_last_error.argtypes = [vp]
    L.pcr_sc_last_error.restype = C.c_char_p
    L.pcr_sc_size.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.pcr_sc_add.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_int]
    L.pcr_sc_descriptor.argtypes = [vp, C.c_size_t, vp, vp, vp]
    L.pcr_sc_distance.argtypes = [vp, C.c_size_t, C.c_size_t, dp, ip]
    L.pcr_sc_query.argtypes = [vp, C.c_longlong, C.POINTER(C.c_longlong), C.POINTER(C.c_float), dp]
    _lib = L
    return L


def default_params(**overrides):
    p = PcrParams()
    load_library().pcr_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(f"pcr_params has no field {k!r}")
        setattr(p, k, v)
    return p


class PcrError(RuntimeError):
    pass


def _cloud(x):
    """-> (pointer, n, stride_bytes, on_device, keepalive)"""
    if hasattr(x, "data_ptr"):  # torch tensor
        import torch
        if x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] < 3 or not x.is_contiguous():
            raise ValueError("cloud tensors must be contiguous float32 of shape (n, >=3)")
        return C.c_void_p(x.data_ptr()), x.shape[0], x.shape[1] * 4, 1 if x.is_cuda else 0, x
    a = np.ascontiguousarray(x, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] < 3:
        raise ValueError("clouds must have shape (n, >=3)")
    return a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1] * 4, 0, a


def _pose_in(T):
    T = np.asarray(T, dtype=np.float64)
    if T.shape != (4, 4):
        raise ValueError("pose must be 4x4")
    return np.ascontiguousarray(T.T).reshape(16).copy()  # column-major


def _pose_out(buf):
    return buf.reshape(4, 4).T.copy()


class PointCloudRegister:
    """PCR::PointCloudRegister (reference PointCloudRegister.hpp:12-38)."""

    method = None

    def __init__(self, params=None, **overrides):
        self._lib = load_library()
        if params is None:
            params = default_params(**overrides)
        elif overrides:
            for k, v in overrides.items():
                setattr(params, k, v)
        self.params = params
        self._h = self._lib.pcr_create(self.method.encode(), C.byref(params))
        if not self._h:
            raise PcrError(self._lib.pcr_last_error(None).decode())
        self.isConverge = False
        # per-call marshalling kept off the hot path (a C++ caller has none): one persistent pose buffer and flag, and the
        # (pointer, size, stride) triple of the most recent device tensors remembered by identity
        self._pose_buf = (C.c_double * 16)()
        self._pose_cm = np.frombuffer(self._pose_buf, dtype=np.float64).reshape(4, 4)     # column-major 4x4 = transposed view
        self._conv = C.c_int(0)
        self._conv_ref = C.byref(self._conv)
        self._seen = {}

    def _cloud_cached(self, x):
        """(pointer, n, stride, on_device, keepalive) of a device tensor, remembered by identity through a WEAK reference (no tensor
        is kept alive by this object) and revalidated on every hit: same storage pointer, shape, strides and dtype."""
        if hasattr(x, "data_ptr"):
            import weakref
            hit = self._seen.get(id(x))
            if hit is not None and hit[0]() is x and hit[2] == (x.data_ptr(), tuple(x.shape), x.stride(), x.dtype):
                return hit[1]
            c = _cloud(x)
            if len(self._seen) > 64:
                self._seen = {k: v for k, v in self._seen.items() if v[0]() is not None}
                if len(self._seen) > 64:
                    self._seen.clear()
            self._seen[id(x)] = (weakref.ref(x), c[:4] + (None,), (x.data_ptr(), tuple(x.shape), x.stride(), x.dtype))
            return c
        return _cloud(x)

    # -- reference interface ------------------------------------------------
    def scan2Map(self, src, dst, res):
        """Refine `res` (4x4, map<-lidar) in place; returns isConverge.
        The target index is rebuilt on every call, like the reference (LoamRegister.cpp:110)."""
        sp, sn, ss, sdev, _k1 = self._cloud_cached(src)
        dp_, dn, ds, ddev, _k2 = self._cloud_cached(dst)
        if ss != ds:
            raise ValueError("src and dst must share a point stride")
        if sdev != ddev:
            raise ValueError("src and dst must both be host arrays or both be device tensors")
        res_np = np.asarray(res)
        if res_np.shape != (4, 4):
            raise ValueError("pose must be 4x4")
        self._pose_cm[...] = res_np.T
        fn = self._lib.pcr_scan2map_device if sdev else self._lib.pcr_scan2map
        self._check(fn(self._h, sp, sn, dp_, dn, ss, self._pose_buf, self._conv_ref))
        res_np[...] = self._pose_cm.T
        self.isConverge = bool(self._conv.value)
        return self.isConverge

    def getFitnessScore(self):
        return float(self._lib.pcr_fitness(self._h))

    # -- static-map localisation (reference test/loc.cpp) ----------------------
    def setTarget(self, dst):
        p, n, s, dev, _k = _cloud(dst)
        self._check(self._lib.pcr_set_target(self._h, p, n, s, dev))

    def align(self, src, res):
        p, n, s, dev, _k = self._cloud_cached(src)
        res_np = np.asarray(res)
        if res_np.shape != (4, 4):
            raise ValueError("pose must be 4x4")
        self._pose_cm[...] = res_np.T
        self._check(self._lib.pcr_align(self._h, p, n, s, dev, self._pose_buf, self._conv_ref))
        res_np[...] = self._pose_cm.T
        self.isConverge = bool(self._conv.value)
        return self.isConverge

    def invalidateTarget(self):
        self._check(self._lib.pcr_invalidate_target(self._h))

    def scan2MapSubmap(self, src, submap, pose, rebuild=False):
        """scan2Map with the device-resident sub-map of a SubMap as `dst` (no host copy of the map).  The handle keeps the target
        structures it builds for as long as the SubMap stays at the same generation (pcr_scan2map_submap); rebuild=True is the
        reference's behaviour of rebuilding on every call (pcr_scan2map_device on the same memory) -- same result either way."""
        p, n, s, dev, _k = _cloud(src)
        dp, dn, ds = submap.pointer()
        if dn and ds != s:
            raise ValueError("scan and sub-map must share one point layout")
        if not dev:
            import torch
            t = torch.from_numpy(np.ascontiguousarray(src, np.float32)).cuda()
            p, _k = C.c_void_p(t.data_ptr()), t
        pc = _pose_in(pose)
        conv = C.c_int(0)
        if rebuild:
            self._check(self._lib.pcr_scan2map_device(self._h, p, n, C.c_void_p(dp), dn, s, pc.ctypes.data_as(C.POINTER(C.c_double)), C.byref(conv)))
        else:
            self._check(self._lib.pcr_scan2map_submap(self._h, p, n, 1, submap._m, pc.ctypes.data_as(C.POINTER(C.c_double)), C.byref(conv)))
        np.asarray(pose)[...] = _pose_out(pc)
        self.isConverge = bool(conv.value)
        return self.isConverge

    # -- the step before the path ------------------------------------------------
    def voxelDownSample(self, cloud, grid_size):
        """pcp::voxelDownSample / pcl::VoxelGrid (reference common/pcp/pcp.hpp:14-28): centroid per occupied voxel, ascending
        voxel index.  Host array in -> numpy array out; CUDA tensor in -> CUDA tensor out (nothing leaves HBM)."""
        p, n, s, dev, keep = _cloud(cloud)
        cnt = C.c_size_t(0)
        if dev:
            import torch
            out = torch.empty((max(n, 1), s // 4), dtype=torch.float32, device=keep.device)
            self._check(self._lib.pcr_voxel_filter(self._h, p, n, s, 1, float(grid_size), C.c_void_p(out.data_ptr()), n, 1, C.byref(cnt)))
            return out[:cnt.value]
        out = np.zeros((max(n, 1), s // 4), np.float32)
        self._check(self._lib.pcr_voxel_filter(self._h, p, n, s, 0, float(grid_size), out.ctypes.data_as(C.c_void_p), n, 0, C.byref(cnt)))
        return out[:cnt.value].copy()

    def voxelDownSampleBegin(self, cloud, grid_size):
        """pcr_voxel_filter_begin for a CUDA tensor: queue the filter, return a token for voxelDownSampleEnd (the input must stay alive until then)."""
        import torch
        p, n, s, dev, keep = _cloud(cloud)
        if not dev:
            raise PcrError("voxelDownSampleBegin takes a CUDA tensor")
        out = torch.empty((max(n, 1), s // 4), dtype=torch.float32, device=keep.device)
        self._check(self._lib.pcr_voxel_filter_begin(self._h, p, n, s, float(grid_size), C.c_void_p(out.data_ptr()), n))
        return (out, keep)

    def voxelDownSampleEnd(self, token):
        cnt = C.c_size_t(0)
        self._check(self._lib.pcr_voxel_filter_end(self._h, C.byref(cnt)))
        return token[0][:cnt.value]

    # -- introspection ---------------------------------------------------------
    def stats(self):
        st = PcrStats()
        self._check(self._lib.pcr_get_stats(self._h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in PcrStats._fields_}

    def set_profile(self, level):
        self._check(self._lib.pcr_set_profile(self._h, int(level)))

    def set_stream(self, stream_ptr):
        self._check(self._lib.pcr_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_query_tile(self, lo, hi):
        lo = np.ascontiguousarray(lo, np.float64)
        hi = np.ascontiguousarray(hi, np.float64)
        dp = C.POINTER(C.c_double)
        self._check(self._lib.pcr_set_query_tile(self._h, lo.ctypes.data_as(dp), hi.ctypes.data_as(dp)))

    def comm_init(self, unique_id, rank, nranks):
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        self._check(self._lib.pcr_comm_init(self._h, C.cast(buf, C.c_void_p), rank, nranks))

    def comm_info(self):
        """pcr_comm_info -> dict(rank, nranks, transport): who takes part in this handle's exchange, as the communicator reports it."""
        r, n, t = C.c_int(0), C.c_int(1), C.c_int(0)
        self._check(self._lib.pcr_comm_info(self._h, C.byref(r), C.byref(n), C.byref(t)))
        return dict(rank=r.value, nranks=n.value, transport={0: "none", 1: "rccl", 2: "host", 3: "peer"}[t.value])

    def comm_peer_export(self):
        """64 bytes naming this rank's receive buffer of the peer exchange (pcr_comm_peer_export): share them in rank order, then comm_init_peer"""
        buf = C.create_string_buffer(64)
        self._check(self._lib.pcr_comm_peer_export(self._h, buf))
        return bytes(buf.raw)

    def comm_init_peer(self, handles, rank, nranks):
        """pcr_comm_init_peer: `handles` = every rank's comm_peer_export(), in rank order"""
        blob = b"".join(handles)
        assert len(blob) == 64 * nranks
        self._check(self._lib.pcr_comm_init_peer(self._h, blob, rank, nranks))

    def comm_init_host(self, fn, rank, nranks):
        """The exchange of a sharded call through the caller's collective (pcr_comm_init_host): fn(ptr, count, op, user) -> 0,
        combining `count` doubles in place over all ranks (op 0 = sum, 1 = max).  shard.ThreadCollective / shard.gloo_collective
        provide one.  fn = None clears it."""
        self._ar = ALLREDUCE_FN(fn) if fn is not None else ALLREDUCE_FN()      # kept alive with the handle
        self._check(self._lib.pcr_comm_init_host(self._h, self._ar, None, rank, nranks))

    def set_shard(self, lo, hi, halo):
        """pcr_set_shard: this rank's tile [lo, hi) and the halo its target cloud carries (shard.tile_for_method)."""
        lo = np.ascontiguousarray(lo, np.float64)
        hi = np.ascontiguousarray(hi, np.float64)
        dp = C.POINTER(C.c_double)
        self._check(self._lib.pcr_set_shard(self._h, lo.ctypes.data_as(dp), hi.ctypes.data_as(dp), float(halo)))

    def set_params(self, **overrides):
        """pcr_set_params on the live handle: change optimiser settings between calls (e.g. VgicpRegister.initForLC)."""
        p = PcrParams()
        self._check(self._lib.pcr_get_params(self._h, C.byref(p)))
        for k, v in overrides.items():
            if not hasattr(p, k):
                raise AttributeError(f"pcr_params has no field {k!r}")
            setattr(p, k, v)
        self._check(self._lib.pcr_set_params(self._h, C.byref(p)))
        self.params = p

    def fitnessGated(self, src, pose, max_sq=1.0):
        """getFitnessScore of the reference's test/align.cpp:29-61 against the handle's current target: mean of the squared
        1-NN distances <= max_sq of the source transformed by `pose` (-1 when none) and the number of points counted."""
        p, n, s, dev, _k = _cloud(src)
        pc = _pose_in(pose)
        score, cnt = C.c_double(0), C.c_int64(0)
        self._check(self._lib.pcr_fitness_gated(self._h, p, n, s, dev, pc.ctypes.data_as(C.POINTER(C.c_double)), float(max_sq), C.byref(score), C.byref(cnt)))
        return score.value, int(cnt.value)

    def _check(self, rc):
        if rc != 0:
            raise PcrError(self._lib.pcr_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pcr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LoamRegister(PointCloudRegister):
    """PCR::LoamRegister (reference PCR/src/LoamRegister.cpp)."""
    method = "loam"

    def linearize(self, src, pose, per_point=False):
        """One linearisation against the current target (setTarget): dict(JtJ, JtE, n[, status, rows, nn])."""
        p, n, s, dev, _k = _cloud(src)
        pc = _pose_in(pose)
        JtJ = np.zeros(36)
        JtE = np.zeros(6)
        cnt = C.c_int64(0)
        status = np.zeros(n, np.int8) if per_point else None
        rows = np.zeros((n, 7)) if per_point else None
        nn = np.zeros((n, 5), np.int32) if per_point else None
        dp = C.POINTER(C.c_double)
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        self._check(self._lib.pcr_loam_linearize(self._h, p, n, s, dev, pc.ctypes.data_as(dp), JtJ.ctypes.data_as(dp),
                                                 JtE.ctypes.data_as(dp), C.byref(cnt), vp(status), vp(rows), vp(nn)))
        out = dict(JtJ=JtJ.reshape(6, 6), JtE=JtE, n=int(cnt.value))
        if per_point:
            out.update(status=status, rows=rows, nn=nn)
        return out

    def trace(self):
        """Per-iteration normal equations of the last call (needs record_trace=1)."""
        it = max(1, self.params.loam_iters)
        JtJ, JtE, n, x = np.zeros((it, 36)), np.zeros((it, 6)), np.zeros(it, np.int64), np.zeros((it, 6))
        k = C.c_int32(0)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        self._check(self._lib.pcr_get_trace(self._h, C.byref(k), vp(JtJ), vp(JtE), vp(n), vp(x)))
        k = k.value
        hits, srch = np.zeros(it, np.int64), np.zeros(it, np.int64)
        self._check(self._lib.pcr_get_trace_counts(self._h, vp(hits), vp(srch)))
        return dict(iters_run=k, JtJ=JtJ[:k].reshape(-1, 6, 6), JtE=JtE[:k], n=n[:k], x=x[:k], cache_hits=hits[:k], searches=srch[:k])

    def timeline(self):
        """[launch][block][8] stamps in microseconds relative to each launch's earliest block entry
        (needs pcr_params.record_timeline = 1).  Columns 0-7: entry, prologue, posted, searched, plane, accumulated, stored, partial sums folded (inside the
        prologue); 8-10 (dense search only): row ranges in LDS, first candidate chunk arrived, candidate stream done; 11: normal
        equations solved (inside the prologue)."""
        nl, nb = C.c_int(0), C.c_int(0)
        self._check(self._lib.pcr_get_timeline(self._h, None, 0, C.byref(nl), C.byref(nb)))
        raw = np.zeros((nl.value, nb.value, 16), np.uint64)
        self._check(self._lib.pcr_get_timeline(self._h, raw.ctypes.data_as(C.c_void_p), raw.size, C.byref(nl), C.byref(nb)))
        t = raw.astype(np.int64)
        return (t - t[:, :, :1].min(axis=1, keepdims=True)) / 100.0


class NdtRegister(PointCloudRegister):
    """PCR::NdtRegister (reference PCR/src/NdtRegister.cpp)."""
    method = "ndt"

    def derivatives(self, src, p6, double_hessian=False):
        """One computeDerivatives pass at p = [t; roll pitch yaw] against the current target."""
        p, n, s, dev, _k = _cloud(src)
        p6 = np.ascontiguousarray(p6, np.float64).reshape(6)
        g, H = np.zeros(6), np.zeros(36)
        Hd = np.zeros(36) if double_hessian else None
        sc = C.c_double(0)
        dp = C.POINTER(C.c_double)
        self._check(self._lib.pcr_ndt_derivatives(self._h, p, n, s, dev, p6.ctypes.data_as(dp), C.byref(sc), g.ctypes.data_as(dp),
                                                  H.ctypes.data_as(dp), Hd.ctypes.data_as(dp) if Hd is not None else None))
        out = dict(score=sc.value, grad=g, hess=H.reshape(6, 6))
        if double_hessian:
            out["hess_d"] = Hd.reshape(6, 6)
        return out


class VgicpRegister(PointCloudRegister):
    """PCR::VgicpRegister (reference PCR/src/VgicpRegister.cpp)."""
    method = "vgicp"

    def initForLC(self):
        """VgicpRegister::initForLC (VgicpRegister.cpp:21-28) on the live object, as LoopClosureManager's constructor calls it
        (backend/src/LoopClosureManager.cpp:21-22): 100 iterations, transformation epsilon 1e-6.  (Its
        setMaxCorrespondenceDistance(150) only acts in the non-voxel GICP, fast_gicp_impl.hpp:18.)"""
        self.set_params(vgicp_max_iters=100, vgicp_trans_eps=1e-6)

    def covariances(self, pts):
        """(n,3,3) per-point covariances (fast_gicp_impl.hpp:241-297)."""
        p, n, s, dev, _k = _cloud(pts)
        c6 = np.zeros((n, 6))
        self._check(self._lib.pcr_vgicp_covariances(self._h, p, n, s, dev, c6.ctypes.data_as(C.c_void_p)))
        out = np.zeros((n, 3, 3))
        out[:, 0, 0], out[:, 0, 1], out[:, 0, 2], out[:, 1, 1], out[:, 1, 2], out[:, 2, 2] = c6.T
        out[:, 1, 0], out[:, 2, 0], out[:, 2, 1] = out[:, 0, 1], out[:, 0, 2], out[:, 1, 2]
        return out

    def neighbours(self, n):
        """(n,20) original indices of every point's 20 nearest neighbours as the last covariances() call on a scan-sized cloud found them
        (0xffffffff: none), and the number of queries that went to the wave-per-query search."""
        out = np.zeros((n, 20), np.uint32)
        q = C.c_uint32(0)
        self._check(self._lib.pcr_vgicp_neighbours(self._h, n, out.ctypes.data_as(C.c_void_p), C.byref(q)))
        return out, int(q.value)

    def linearize(self, src, pose):
        p, n, s, dev, _k = _cloud(src)
        pc = _pose_in(pose)
        H, b = np.zeros(36), np.zeros(6)
        err, nc = C.c_double(0), C.c_int64(0)
        dp = C.POINTER(C.c_double)
        self._check(self._lib.pcr_vgicp_linearize(self._h, p, n, s, dev, pc.ctypes.data_as(dp), H.ctypes.data_as(dp),
                                                  b.ctypes.data_as(dp), C.byref(err), C.byref(nc)))
        return dict(H=H.reshape(6, 6), b=b, err=err.value, n=int(nc.value))


def make_register(pcr_type, **overrides):
    """The reference's factory on cfg["frontend"]["pcr"] (LidarOdometry.cpp:32,44-54)."""
    table = {"loam": LoamRegister, "ndt": NdtRegister, "vgicp": VgicpRegister}
    if pcr_type not in table:
        raise RuntimeError(f"such pcr type({pcr_type}) is not exist, please implemented your self!")
    return table[pcr_type](**overrides)


def host_pin(array):
    """pcr_host_pin on a numpy array's buffer (kept pinned until host_unpin(array); the array must stay alive and unresized)."""
    a = np.asarray(array)
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("only contiguous arrays can be pinned")
    if load_library().pcr_host_pin(a.ctypes.data_as(C.c_void_p), a.nbytes) != 0:
        raise PcrError(load_library().pcr_last_error(None).decode())


def host_unpin(array):
    a = np.asarray(array)
    if load_library().pcr_host_unpin(a.ctypes.data_as(C.c_void_p)) != 0:
        raise PcrError(load_library().pcr_last_error(None).decode())


def comm_unique_id():
    buf = (C.c_char * 128)()
    if load_library().pcr_comm_unique_id(C.cast(buf, C.c_void_p)) != 0:
        raise PcrError(load_library().pcr_last_error(None).decode())
    return bytes(buf)


class SubMap:
    """The key-frame store and sub-map of the reference's MapManager (frontend/src/MapManager.cpp:151-201), kept in HBM.

    addKeyFrame(points, pose); updateMap(position) selects the key frames within `radius` (8 m, MapManager.hpp:68), transforms
    and concatenates them and voxel-filters the result; `pointer()` is the device-resident sub-map that
    PointCloudRegister.scan2MapSubmap registers against without a host copy."""

    def __init__(self, device=-1):
        self._lib = load_library()
        self._m = self._lib.pcr_map_create(int(device))
        if not self._m:
            raise PcrError(self._lib.pcr_map_last_error(None).decode())

    def __del__(self):
        try:
            if self._m:
                self._lib.pcr_map_destroy(self._m)
                self._m = None
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise PcrError(self._lib.pcr_map_last_error(self._m).decode())

    def addKeyFrame(self, cloud, pose):
        p, n, s, dev, _keep = _cloud(cloud)
        pc = _pose_in(pose)
        self._check(self._lib.pcr_map_add_keyframe(self._m, p, n, s, dev, pc.ctypes.data_as(C.POINTER(C.c_double))))

    def generation(self):
        """(store id, generation of the sub-map it holds): every updateMap / updateWindow starts a new generation."""
        i, g = C.c_uint64(0), C.c_uint64(0)
        self._check(self._lib.pcr_map_generation(self._m, C.byref(i), C.byref(g)))
        return int(i.value), int(g.value)

    def keyframes(self):
        n = C.c_size_t(0)
        self._check(self._lib.pcr_map_keyframes(self._m, C.byref(n)))
        return n.value

    def updateMap(self, position, radius=8.0, grid_size=0.4):
        pos = np.ascontiguousarray(position, np.float64).reshape(3)
        n = C.c_size_t(0)
        self._check(self._lib.pcr_map_update(self._m, pos.ctypes.data_as(C.POINTER(C.c_double)), float(radius), float(grid_size), C.byref(n)))
        return n.value

    def clear(self):
        """pcr_map_clear: forget the key frames and the sub-map, keep the device memory (a new session on a store that has grown)."""
        self._check(self._lib.pcr_map_clear(self._m))

    def updateMapBegin(self, position, radius=8.0, grid_size=0.4):
        """pcr_map_update_begin: select + queue the assembly (the reference's map thread works beside the front end); wait() -- or whatever asks for
        the sub-map first -- collects it."""
        pos = np.ascontiguousarray(position, np.float64).reshape(3)
        self._check(self._lib.pcr_map_update_begin(self._m, pos.ctypes.data_as(C.POINTER(C.c_double)), float(radius), float(grid_size)))

    def wait(self):
        """pcr_map_wait -> points of the sub-map."""
        n = C.c_size_t(0)
        self._check(self._lib.pcr_map_wait(self._m, C.byref(n)))
        return n.value

    def loopFindNearKeyframes(self, key, search_num, grid_size=0.4):
        """LoopClosureManager::loopFindNearKeyframes (backend/src/LoopClosureManager.cpp:40-60): the key frames key +- search_num,
        transformed, concatenated and voxel-filtered, as the device-resident target of the loop-closure registration."""
        n = C.c_size_t(0)
        self._check(self._lib.pcr_map_update_window(self._m, int(key), int(search_num), float(grid_size), C.byref(n)))
        return n.value

    def pointer(self):
        """-> (device pointer, number of points, stride in bytes) of the assembled sub-map."""
        n, s = C.c_size_t(0), C.c_size_t(0)
        p = self._lib.pcr_map_submap(self._m, C.byref(n), C.byref(s))
        return p, n.value, s.value

    def submapIdx(self):
        n = C.c_size_t(0)
        self._check(self._lib.pcr_map_submap_indices(self._m, None, 0, C.byref(n)))
        idx = np.zeros(n.value, np.int64)
        if n.value:
            self._check(self._lib.pcr_map_submap_indices(self._m, idx.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return idx

    def download(self):
        """The sub-map as a host array (tests, visualisation)."""
        import torch
        p, n, s = self.pointer()
        out = torch.empty((n, s // 4), dtype=torch.float32, device="cuda")
        if n:
            hip = C.CDLL("libamdhip64.so")
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            rc = hip.hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(p), n * s, 3)      # hipMemcpyDeviceToDevice
            if rc:
                raise PcrError(f"hipMemcpy failed: {rc}")
        return out.cpu().numpy()


class ScanContext:
    """context::ScanContext (backend/include/backend/ScanContext.hpp, backend/src/ScanContext.cpp): addContext(scan) builds the
    20 x 60 polar descriptor on the device; query(id) -> (match id or -1, yaw in rad), QueryResult of the reference."""

    def __init__(self, device=-1, **params):
        self._lib = load_library()
        p = ScParams()
        self._lib.pcr_sc_default_params(C.byref(p))
        for k, v in params.items():
            if not hasattr(p, k):
                raise ValueError(f"unknown ScanContext parameter {k}")
            setattr(p, k, v)
        self._height = float(p.lidar_height)
        self._s = self._lib.pcr_sc_create(int(device), C.byref(p))
        if not self._s:
            raise PcrError(self._lib.pcr_sc_last_error(None).decode())

    def __del__(self):
        try:
            if self._s:
                self._lib.pcr_sc_destroy(self._s)
                self._s = None
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise PcrError(self._lib.pcr_sc_last_error(self._s).decode())

    def addContext(self, cloud):
        p, n, s, dev, _keep = _cloud(cloud)
        self._check(self._lib.pcr_sc_add(self._s, p, n, s, dev))

    def __len__(self):
        n = C.c_size_t(0)
        self._check(self._lib.pcr_sc_size(self._s, C.byref(n)))
        return n.value

    def descriptor(self, i):
        """-> (20 x 60 descriptor, ring key, sector key)"""
        d, rk, sk = np.zeros((20, 60)), np.zeros(20), np.zeros(60)
        self._check(self._lib.pcr_sc_descriptor(self._s, int(i), d.ctypes.data_as(C.c_void_p), rk.ctypes.data_as(C.c_void_p), sk.ctypes.data_as(C.c_void_p)))
        return d, rk, sk

    def distance(self, i, j):
        d, sh = C.c_double(0), C.c_int(0)
        self._check(self._lib.pcr_sc_distance(self._s, int(i), int(j), C.byref(d), C.byref(sh)))
        return d.value, sh.value

    def query(self, i):
        """-> (match or -1, yaw as float32, best candidate distance or None when the search did not run)"""
        m, yaw, d = C.c_longlong(-1), C.c_float(0), C.c_double(0)
        self._check(self._lib.pcr_sc_query(self._s, int(i), C.byref(m), C.byref(yaw), C.byref(d)))
        return m.value, np.float32(yaw.value), (None if d.value == np.finfo(np.float64).max else d.value)
