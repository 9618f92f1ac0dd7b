"""oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes bindings for the CPU restatement of the reference's registration path
(oracle/*_oracle.c -> oracle/liboracle.so) and, when present, for the
reference's own vendored nanoflann compiled into oracle/_ref.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; nothing under simpleslam_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_DIR, "liboracle.so")
_REF = os.path.join(_DIR, "_ref", "libref_nanoflann.so")


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_LIB) or any(
        os.path.getmtime(os.path.join(_DIR, f)) > os.path.getmtime(_LIB)
        for f in os.listdir(_DIR) if f.endswith("_oracle.c") or f == "Makefile"
    ):
        subprocess.check_call(["make", "-C", _DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    if not os.path.exists(_REF) or force:
        subprocess.check_call(["make", "-C", _DIR, "ref"], stdout=subprocess.DEVNULL)


class LoamParams(C.Structure):
    _fields_ = [
        ("plane_pts", C.c_int), ("knn_max_sq", C.c_double), ("plane_thresh", C.c_double),
        ("point_thresh", C.c_double), ("pos_conv", C.c_double), ("rot_conv", C.c_double),
        ("iters", C.c_int), ("early_exit", C.c_int), ("threads", C.c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.oracle_kd_build.restype = C.c_void_p
        L.oracle_kd_build.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
        L.oracle_kd_free.argtypes = [C.c_void_p]
        L.oracle_kd_knn.restype = C.c_int
        L.oracle_kd_knn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_knn_brute.restype = C.c_int
        L.oracle_knn_brute.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_colpiv_qr_solve_5x3.restype = C.c_int
        L.oracle_colpiv_qr_solve_5x3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_plane_fit5.restype = C.c_int
        L.oracle_plane_fit5.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
        L.oracle_ldlt6_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_se3_exp.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_t2se3.argtypes = [C.c_void_p]
        L.oracle_pose_update.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_loam_default_params.argtypes = [C.POINTER(LoamParams)]
        L.oracle_loam_linearize.restype = C.c_long
        L.oracle_loam_linearize.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                            C.POINTER(LoamParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]
        L.oracle_loam_scan2map.restype = C.c_int
        L.oracle_loam_scan2map.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                           C.POINTER(LoamParams), C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        _lib = L
    return _lib


def set_variant(which, value):
    """Switch one of the two UNPINNED readings of the reference (see oracle_set_variant in loam_oracle.c) to its alternative:
    which 0 = LOAM weight roots in double, 1 = NDT rotation() as Eigen's polar factor.  Measurement aid; default 0 = the reading the
    HIP path implements."""
    L = lib()
    L.oracle_set_variant.argtypes = [C.c_int, C.c_int]
    L.oracle_set_variant.restype = None
    L.oracle_set_variant(int(which), int(value))
    try:
        _nd().oracle_set_variant  # same shared object
    except Exception:
        pass


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] >= 3
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def loam_params(**kw):
    p = LoamParams()
    lib().oracle_loam_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


class KdTree:
    """Exact k-NN index over float xyz rows (own restatement)."""

    def __init__(self, pts):
        self.pts = _f32(pts)
        self.h = lib().oracle_kd_build(_p(self.pts), self.pts.shape[0], self.pts.shape[1])

    def knn(self, queries, k=5):
        q = np.ascontiguousarray(queries, dtype=np.float64).reshape(-1, 3)
        idx = np.full((q.shape[0], k), -1, np.int32)
        d2 = np.full((q.shape[0], k), np.inf, np.float64)
        L = lib()
        for i in range(q.shape[0]):
            L.oracle_kd_knn(self.h, _p(q[i]), k, _p(idx[i]), _p(d2[i]))
        return idx, d2

    def __del__(self):
        try:
            lib().oracle_kd_free(self.h)
        except Exception:
            pass


def knn_brute(pts, queries, k=5):
    pts = _f32(pts)
    q = np.ascontiguousarray(queries, dtype=np.float64).reshape(-1, 3)
    idx = np.full((q.shape[0], k), -1, np.int32)
    d2 = np.full((q.shape[0], k), np.inf, np.float64)
    L = lib()
    for i in range(q.shape[0]):
        L.oracle_knn_brute(_p(pts), pts.shape[0], pts.shape[1], _p(q[i]), k, _p(idx[i]), _p(d2[i]))
    return idx, d2


def colpiv_qr_solve_5x3(A, b):
    A = np.ascontiguousarray(A, np.float64).reshape(5, 3)
    b = np.ascontiguousarray(b, np.float64).reshape(5)
    x = np.zeros(3)
    r = lib().oracle_colpiv_qr_solve_5x3(_p(A), _p(b), _p(x))
    return x, r


def plane_fit5(A, thresh=float(np.float32(0.2))):
    A = np.ascontiguousarray(A, np.float64).reshape(5, 3)
    x = np.zeros(3)
    ok = lib().oracle_plane_fit5(_p(A), _p(x), thresh)
    return x, bool(ok)


def ldlt6_solve(M, rhs):
    M = np.ascontiguousarray(M, np.float64).reshape(6, 6)
    rhs = np.ascontiguousarray(rhs, np.float64).reshape(6)
    x = np.zeros(6)
    lib().oracle_ldlt6_solve(_p(M), _p(rhs), _p(x))
    return x


def se3_exp(k):
    """4x4 (numpy row/col indexable) of exp([rho; omega])."""
    k = np.ascontiguousarray(k, np.float64).reshape(6)
    T = np.zeros(16)
    lib().oracle_se3_exp(_p(k), _p(T))
    return T.reshape(4, 4).T.copy()


def t2se3(T):
    Tc = np.ascontiguousarray(np.asarray(T, np.float64).T).reshape(16).copy()
    lib().oracle_t2se3(_p(Tc))
    return Tc.reshape(4, 4).T.copy()


def loam_linearize(tree, src, pose, params=None, per_point=False):
    """One linearisation. pose: 4x4. Returns dict(JtJ 6x6, JtE 6, n[, status, rows, nn])."""
    src = _f32(src)
    params = params or loam_params()
    pc = np.ascontiguousarray(np.asarray(pose, np.float64).T).reshape(16)
    JtJ = np.zeros(36)
    JtE = np.zeros(6)
    n_src = src.shape[0]
    status = np.zeros(n_src, np.int8) if per_point else None
    rows = np.zeros((n_src, 7)) if per_point else None
    nn = np.zeros((n_src, 5), np.int32) if per_point else None
    n = lib().oracle_loam_linearize(tree.h, _p(src), n_src, src.shape[1], _p(pc), C.byref(params), _p(JtJ), _p(JtE),
                                    _p(status), _p(rows), _p(nn))
    out = dict(JtJ=JtJ.reshape(6, 6), JtE=JtE, n=int(n))
    if per_point:
        out.update(status=status, rows=rows, nn=nn)
    return out


def loam_scan2map(src, dst, pose, params=None, trace=False):
    """Full LOAM scan2Map. Returns (pose 4x4, converged, info)."""
    src = _f32(src)
    dst = _f32(dst)
    assert src.shape[1] == dst.shape[1], "src/dst must share a row stride"
    params = params or loam_params()
    pc = np.ascontiguousarray(np.asarray(pose, np.float64).T).reshape(16).copy()
    tr = np.zeros((params.iters, 43)) if trace else None
    tx = np.zeros((params.iters, 6)) if trace else None
    nrun = C.c_int(0)
    conv = lib().oracle_loam_scan2map(_p(src), src.shape[0], _p(dst), dst.shape[0], src.shape[1], _p(pc),
                                      C.byref(params), _p(tr), _p(tx), C.byref(nrun))
    info = dict(iters_run=nrun.value)
    if trace:
        info.update(JtJ=tr[:, :36].reshape(-1, 6, 6), JtE=tr[:, 36:42], n=tr[:, 42].astype(np.int64), x=tx)
    return pc.reshape(4, 4).T.copy(), bool(conv), info


# ---------------------------------------------------------------------------
# The reference's own nanoflann (oracle/_ref), when it was built
# ---------------------------------------------------------------------------
_ref = None


def ref_available():
    return os.path.exists(_REF)


def ref_lib():
    global _ref
    if _ref is None:
        if not ref_available():
            build()
        R = C.CDLL(_REF)
        R.ref_kd_build.restype = C.c_void_p
        R.ref_kd_build.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
        R.ref_kd_free.argtypes = [C.c_void_p]
        R.ref_kd_knn_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        for f in ("ref_kd_radius", "ref_kfs_radius", "ref_kfs_nearest", "ref_vov_knn"):
            getattr(R, f).restype = C.c_size_t
        R.ref_kd_radius.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        R.ref_kfs_radius.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_size_t]
        R.ref_kfs_nearest.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        R.ref_vov_knn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _ref = R
    return _ref


def use_reference_nanoflann(on=True):
    """Let the LOAM restatement search through the REFERENCE's vendored nanoflann (oracle/_ref) instead of its own kd-tree: same
    neighbours (up to the order of exact ties), the reference's own index cost.  Used by bench.py's cpu_baseline leg."""
    L = lib()
    L.oracle_set_knn_backend.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.oracle_set_knn_backend.restype = None
    if not on:
        L.oracle_set_knn_backend(None, None, None)
        return
    R = ref_lib()
    addr = lambda f: C.cast(f, C.c_void_p)
    L.oracle_set_knn_backend(addr(R.ref_kd_build), addr(R.ref_kd_knn), addr(R.ref_kd_free))


def ref_radius(pts, query, radius, sorted_=False):
    """PointCloudKdtree::radiusSearch (pcl_adaptor.hpp:60-78) through the reference's nanoflann: indices and squared distances as returned."""
    pts = _f32(pts)
    R = ref_lib()
    h = R.ref_kd_build(_p(pts), pts.shape[0], pts.shape[1])
    q = np.ascontiguousarray(query, np.float64).reshape(3)
    idx, d2 = np.zeros(pts.shape[0], np.uint64), np.zeros(pts.shape[0])
    n = R.ref_kd_radius(h, _p(q), float(radius), int(sorted_), _p(idx), _p(d2), pts.shape[0])
    R.ref_kd_free(h)
    return idx[:n].astype(np.int64), d2[:n].copy()


def ref_keyframes_radius(positions, query, radius):
    """KeyFramesKdtree::radiusSearch (kfs_adaptor.hpp:57-75, compiled as it lies) as MapManager::updateMap calls it
    (frontend/src/MapManager.cpp:176-177): key-frame indices in the order the tree returns them, squared distances."""
    pos = np.ascontiguousarray(positions, np.float64).reshape(-1, 3)
    q = np.ascontiguousarray(query, np.float64).reshape(3)
    idx, d2 = np.zeros(len(pos), np.uint64), np.zeros(len(pos))
    n = ref_lib().ref_kfs_radius(_p(pos), len(pos), _p(q), float(radius), _p(idx), _p(d2), len(pos))
    return idx[:n].astype(np.int64), d2[:n].copy()


def ref_ring_key_knn(keys, query, k=10):
    """VectorOfVectorsKdTree<., double, 20>::nearestKSearch (vov_adaptor.h, compiled as it lies; metric_L2) as
    ScanContext::query calls it (backend/src/ScanContext.cpp:250)."""
    keys = np.ascontiguousarray(keys, np.float64).reshape(-1, 20)
    q = np.ascontiguousarray(query, np.float64).reshape(20)
    idx, d2 = np.zeros(k, np.uint64), np.zeros(k)
    n = ref_lib().ref_vov_knn(_p(keys), len(keys), _p(q), int(k), _p(idx), _p(d2))
    return idx[:n].astype(np.int64), d2[:n].copy()


def ref_knn(pts, queries_f32, k=5):
    """k-NN through the reference's vendored nanoflann (f64 distance on f32 coords)."""
    pts = _f32(pts)
    q = _f32(queries_f32)
    R = ref_lib()
    h = R.ref_kd_build(_p(pts), pts.shape[0], pts.shape[1])
    idx = np.zeros((q.shape[0], k), np.int64)
    d2 = np.zeros((q.shape[0], k), np.float64)
    R.ref_kd_knn_batch(h, _p(q), q.shape[0], q.shape[1], k, _p(idx), _p(d2))
    R.ref_kd_free(h)
    return idx, d2


# ---------------------------------------------------------------------------
# VGICP restatement (oracle/vgicp_oracle.c)
# ---------------------------------------------------------------------------
class VgicpParams(C.Structure):
    _fields_ = [("resolution", C.c_double), ("k_corr", C.c_int), ("max_iters", C.c_int), ("lm_inner", C.c_int),
                ("rot_eps", C.c_double), ("trans_eps", C.c_double), ("lm_init", C.c_double), ("threads", C.c_int)]


def _vg():
    L = lib()
    if not getattr(L, "_vg_ready", False):
        L.oracle_vgicp_default_params.argtypes = [C.POINTER(VgicpParams)]
        L.oracle_sym3_eig.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_vgicp_covariances.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_int]
        L.oracle_vgicp_scan2map.restype = C.c_int
        L.oracle_vgicp_scan2map.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                            C.POINTER(VgicpParams), C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_vgicp_error.restype = C.c_double
        L.oracle_vgicp_error.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_vgicp_linearize.restype = C.c_long
        L.oracle_vgicp_linearize.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                             C.POINTER(VgicpParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_vgicp_voxel_at.restype = C.c_int
        L.oracle_vgicp_voxel_at.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_fitness_score.restype = C.c_double
        L.oracle_fitness_score.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_double]
        L.oracle_kd_knn_f32.restype = C.c_int
        L.oracle_kd_knn_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L._vg_ready = True
    return L


def vgicp_params(**kw):
    p = VgicpParams()
    _vg().oracle_vgicp_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def sym3_eig(A):
    A = np.ascontiguousarray(A, np.float64).reshape(3, 3)
    w, V = np.zeros(3), np.zeros(9)
    _vg().oracle_sym3_eig(_p(A), _p(w), _p(V))
    return w, V.reshape(3, 3)


def vgicp_covariances(pts, k=20, threads=1):
    pts = _f32(pts)
    covs = np.zeros((pts.shape[0], 3, 3))
    _vg().oracle_vgicp_covariances(_p(pts), pts.shape[0], pts.shape[1], k, _p(covs), threads)
    return covs


def knn_f32(pts, queries_f32, k):
    """Exact k-NN with float squared distances (PCL/FLANN semantics), ties on the lower index."""
    pts = _f32(pts)
    q = np.ascontiguousarray(queries_f32, np.float32).reshape(-1, 3)
    L = _vg()
    h = L.oracle_kd_build(_p(pts), pts.shape[0], pts.shape[1])
    idx = np.full((q.shape[0], k), -1, np.int32)
    d2 = np.full((q.shape[0], k), np.inf, np.float32)
    for i in range(q.shape[0]):
        L.oracle_kd_knn_f32(h, _p(q[i]), k, _p(idx[i]), _p(d2[i]))
    L.oracle_kd_free(h)
    return idx, d2


def vgicp_scan2map(src, dst, pose, params=None, src_covs=None, dst_covs=None):
    src, dst = _f32(src), _f32(dst)
    assert src.shape[1] == dst.shape[1]
    params = params or vgicp_params()
    pc = np.ascontiguousarray(np.asarray(pose, np.float64).T).reshape(16).copy()
    info = np.zeros(4, np.int64)
    sc = np.ascontiguousarray(src_covs, np.float64) if src_covs is not None else None
    dc = np.ascontiguousarray(dst_covs, np.float64) if dst_covs is not None else None
    conv = _vg().oracle_vgicp_scan2map(_p(src), src.shape[0], _p(dst), dst.shape[0], src.shape[1], _p(pc), C.byref(params),
                                       _p(sc), _p(dc), _p(info))
    return pc.reshape(4, 4).T.copy(), bool(conv), dict(outer=int(info[0]), linearizations=int(info[1]),
                                                       error_evals=int(info[2]), correspondences=int(info[3]))


def vgicp_linearize(src, dst, pose, src_covs, dst_covs, params=None):
    src, dst = _f32(src), _f32(dst)
    params = params or vgicp_params()
    pc = np.ascontiguousarray(np.asarray(pose, np.float64).T).reshape(16).copy()
    sc = np.ascontiguousarray(src_covs, np.float64)
    dc = np.ascontiguousarray(dst_covs, np.float64)
    H, b, err = np.zeros(36), np.zeros(6), C.c_double(0)
    nc = _vg().oracle_vgicp_linearize(_p(src), src.shape[0], _p(dst), dst.shape[0], src.shape[1], _p(pc), C.byref(params),
                                      _p(sc), _p(dc), _p(H), _p(b), C.byref(err))
    return dict(H=H.reshape(6, 6), b=b, err=err.value, n=int(nc))


def vgicp_error(src, dst, pose_lin, pose_eval, src_covs, dst_covs, params=None):
    """compute_error at pose_eval on the correspondences of a linearisation at pose_lin (an LM trial, lsq_registration_impl.hpp:141)."""
    src, dst = _f32(src), _f32(dst)
    params = params or vgicp_params()
    pl = np.ascontiguousarray(np.asarray(pose_lin, np.float64).T).reshape(16).copy()
    pe = np.ascontiguousarray(np.asarray(pose_eval, np.float64).T).reshape(16).copy()
    sc = np.ascontiguousarray(src_covs, np.float64)
    dc = np.ascontiguousarray(dst_covs, np.float64)
    return float(_vg().oracle_vgicp_error(_p(src), src.shape[0], _p(dst), dst.shape[0], src.shape[1], _p(pl), _p(pe), C.byref(params),
                                          _p(sc), _p(dc)))


def vgicp_voxel_at(dst, dst_covs, res, p):
    dst = _f32(dst)
    dc = np.ascontiguousarray(dst_covs, np.float64)
    pp = np.ascontiguousarray(p, np.float64)
    mean, cov = np.zeros(3), np.zeros(9)
    n = _vg().oracle_vgicp_voxel_at(_p(dst), dst.shape[0], dst.shape[1], _p(dc), float(res), _p(pp), _p(mean), _p(cov))
    return n, mean, cov.reshape(3, 3)


def fitness_score(src, dst, pose, max_range=1.7976931348623157e308):
    src, dst = _f32(src), _f32(dst)
    pc = np.ascontiguousarray(np.asarray(pose, np.float64).T).reshape(16).copy()
    return float(_vg().oracle_fitness_score(_p(src), src.shape[0], _p(dst), dst.shape[0], src.shape[1], _p(pc), float(max_range)))


# ---------------------------------------------------------------------------
# NDT restatement (oracle/ndt_oracle.c)
# ---------------------------------------------------------------------------
class NdtParams(C.Structure):
    _fields_ = [("resolution", C.c_double), ("step_size", C.c_double), ("outlier_ratio", C.c_double), ("trans_eps", C.c_double),
                ("max_iters", C.c_int), ("min_points", C.c_int), ("eig_mult", C.c_double), ("threads", C.c_int), ("pad_", C.c_int)]


def _nd():
    L = lib()
    if not getattr(L, "_nd_ready", False):
        L.oracle_ndt_default_params.argtypes = [C.POINTER(NdtParams)]
        L.oracle_ndt_scan2map.restype = C.c_int
        L.oracle_ndt_scan2map.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.POINTER(NdtParams),
                                          C.c_void_p, C.POINTER(C.c_double)]
        L.oracle_ndt_derivatives.restype = C.c_double
        L.oracle_ndt_derivatives.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.POINTER(NdtParams),
                                             C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_ndt_leaf_at.restype = C.c_int
        L.oracle_ndt_leaf_at.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(NdtParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_svd6_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_ndt_trial_value.restype = C.c_double
        L.oracle_ndt_trial_value.argtypes = [C.c_double] * 9
        L._nd_ready = True
    return L


def ndt_params(**kw):
    p = NdtParams()
    _nd().oracle_ndt_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def ndt_trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, f_t, g_t):
    """trialValueSelectionMT (ndt_omp_impl.hpp:690-769)."""
    return float(_nd().oracle_ndt_trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, f_t, g_t))


def svd6_solve(A, b):
    A = np.ascontiguousarray(A, np.float64).reshape(6, 6)
    b = np.ascontiguousarray(b, np.float64).reshape(6)
    x = np.zeros(6)
    _nd().oracle_svd6_solve(_p(A), _p(b), _p(x))
    return x


def ndt_scan2map(src, dst, pose, params=None):
    src, dst = _f32(src), _f32(dst)
    assert src.shape[1] == dst.shape[1]
    params = params or ndt_params()
    pc = np.ascontiguousarray(np.asarray(pose, np.float64).T).reshape(16).copy()
    info = np.zeros(3, np.int64)
    sc = C.c_double(0)
    conv = _nd().oracle_ndt_scan2map(_p(src), src.shape[0], _p(dst), dst.shape[0], src.shape[1], _p(pc), C.byref(params), _p(info), C.byref(sc))
    return pc.reshape(4, 4).T.copy(), bool(conv), dict(iterations=int(info[0]), derivative_passes=int(info[1]),
                                                       hessian_passes=int(info[2]), score=sc.value)


def ndt_derivatives(src, dst, p6, params=None, double_hessian=False):
    """score, gradient(6), Hessian(6x6) of computeDerivatives at p = [t; euler xyz]; optionally computeHessian's f64 Hessian."""
    src, dst = _f32(src), _f32(dst)
    params = params or ndt_params()
    p6 = np.ascontiguousarray(p6, np.float64).reshape(6)
    g, H = np.zeros(6), np.zeros(36)
    Hd = np.zeros(36) if double_hessian else None
    s = _nd().oracle_ndt_derivatives(_p(src), src.shape[0], _p(dst), dst.shape[0], src.shape[1], _p(p6), C.byref(params), _p(g), _p(H), _p(Hd))
    out = dict(score=float(s), grad=g, hess=H.reshape(6, 6))
    if double_hessian:
        out["hess_d"] = Hd.reshape(6, 6)
    return out


def ndt_leaf_at(dst, p, params=None):
    dst = _f32(dst)
    params = params or ndt_params()
    pp = np.ascontiguousarray(p, np.float32).reshape(3)
    mean, cov, icov = np.zeros(3), np.zeros(9), np.zeros(9)
    n = _nd().oracle_ndt_leaf_at(_p(dst), dst.shape[0], dst.shape[1], C.byref(params), _p(pp), _p(mean), _p(cov), _p(icov))
    return n, mean, cov.reshape(3, 3), icov.reshape(3, 3)


# ---------------------------------------------------------------------------
# pcl::VoxelGrid restatement (oracle/voxel_oracle.c)
# ---------------------------------------------------------------------------
def voxel_filter(pts, leaf):
    """-> (centroids in ascending voxel index, unfiltered flag)."""
    pts = _f32(pts)
    L = lib()
    L.oracle_voxel_filter.restype = C.c_int
    L.oracle_voxel_filter.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_float, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    out = np.zeros((max(pts.shape[0], 1), pts.shape[1]), np.float32)
    cnt = C.c_size_t(0)
    rc = L.oracle_voxel_filter(_p(pts), pts.shape[0], pts.shape[1], float(leaf), _p(out), pts.shape[0], C.byref(cnt))
    assert rc in (0, 1), rc
    return out[:cnt.value].copy(), rc == 1


# ---------------------------------------------------------------------------
# MapManager::updateMap sub-map assembly (oracle/submap_oracle.c)
# ---------------------------------------------------------------------------
def submap_assemble(clouds, poses, position, radius, grid):
    """clouds: list of (n_k, stride) arrays; poses: list of 4x4 -> (sub-map, indices of the key frames used)."""
    L = lib()
    L.oracle_submap_assemble.restype = C.c_int
    L.oracle_submap_assemble.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_double, C.c_float,
                                         C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    stride = clouds[0].shape[1]
    cat = np.ascontiguousarray(np.concatenate([_f32(c) for c in clouds], 0), np.float32) if clouds else np.zeros((0, 4), np.float32)
    counts = np.array([c.shape[0] for c in clouds], np.uint64)
    P = np.ascontiguousarray(np.stack([np.asarray(T, np.float64).T.reshape(16) for T in poses]), np.float64)
    pos = np.ascontiguousarray(position, np.float64).reshape(3)
    sel = np.zeros(len(clouds), np.int64)
    nsel, nout = C.c_size_t(0), C.c_size_t(0)
    out = np.zeros((max(cat.shape[0], 1), stride), np.float32)
    rc = L.oracle_submap_assemble(_p(cat), _p(counts), _p(P), len(clouds), stride, _p(pos), float(radius), float(grid),
                                  _p(sel), C.byref(nsel), _p(out), cat.shape[0], C.byref(nout))
    assert rc in (0, 1), rc
    return out[:nout.value].copy(), sel[:nsel.value].copy()


class SequenceFront:
    """The CPU side of simpleslam_amd.sequence.drive(): the four steps of the front end's loop by the restatements in this package
    (voxel_filter, submap_assemble, {loam,ndt,vgicp}_scan2map) -- the checker of the GPU front, and bench.py's cpu_baseline for it."""

    def __init__(self, method, params=None):
        self.method, self.params = method, params
        self.clouds, self.kf_poses, self.submap = [], [], np.zeros((0, 4), np.float32)

    def voxel(self, scan, grid):
        return voxel_filter(scan, grid)[0]

    def scan2map(self, ds, pose):
        if self.method == "loam":
            out, conv, info = loam_scan2map(ds, self.submap, pose, self.params)
            it = info["iters_run"]
        elif self.method == "ndt":
            out, conv, info = ndt_scan2map(ds, self.submap, pose, self.params)
            it = info["iterations"]
        else:
            out, conv, info = vgicp_scan2map(ds, self.submap, pose, self.params)
            it = info["outer"]
        pose[...] = out
        return conv, it

    def add_keyframe(self, scan, pose):
        self.clouds.append(_f32(scan)); self.kf_poses.append(np.array(pose, float))

    def update_map(self, position, radius, grid):
        self.submap = submap_assemble(self.clouds, self.kf_poses, position, radius, grid)[0]

    def submap_points(self):
        return self.submap.shape[0]

    def finish(self):
        pass


# ---------------------------------------------------------------------------
# ScanContext loop-closure descriptor (oracle/scancontext_oracle.c); query's state machine restated here
# ---------------------------------------------------------------------------
class ScanContextOracle:
    """backend/src/ScanContext.cpp: addContext :56-66, query :231-279 (ring-key candidates by exact k-NN)."""

    def __init__(self, lidar_height=2.0, num_exclude_recent=40, build_tree_gap=10, num_candidates=10, search_ratio=0.1, dist_thres=0.4):
        self.lidar_height, self.excl, self.gap, self.ncand = lidar_height, num_exclude_recent, build_tree_gap, num_candidates
        self.search_ratio, self.dist_thres = search_ratio, np.float32(dist_thres)
        self.polar, self.ring, self.sector = [], [], []
        self.tree_size = 0
        L = lib()
        L.oracle_sc_make.restype = None
        L.oracle_sc_make.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_double, C.c_void_p]
        L.oracle_sc_keys.restype = None
        L.oracle_sc_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_sc_distance.restype = None
        L.oracle_sc_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        self.L = L

    def add(self, pts):
        pts = _f32(pts)
        desc = np.zeros(20 * 60, np.float64)                      # column-major 20 x 60
        self.L.oracle_sc_make(_p(pts), pts.shape[0], pts.shape[1], float(self.lidar_height), _p(desc))
        rk, sk = np.zeros(20), np.zeros(60)
        self.L.oracle_sc_keys(_p(desc), _p(rk), _p(sk))
        self.polar.append(desc); self.ring.append(rk); self.sector.append(sk)

    def descriptor(self, i):
        return self.polar[i].reshape(60, 20).T.copy()             # as a [ring][sector] array

    def distance(self, i, j):
        d, s = C.c_double(0), C.c_int(0)
        self.L.oracle_sc_distance(_p(self.polar[i]), _p(self.polar[j]), float(self.search_ratio), C.byref(d), C.byref(s))
        return d.value, s.value

    def query(self, i):
        """-> (match or -1, yaw as float32, best distance or None)"""
        if i <= self.excl + self.ncand:
            return -1, np.float32(0), None
        if self.tree_size == 0 or i - self.tree_size > self.excl + self.gap:
            self.tree_size = i - self.excl
        keys = np.stack(self.ring[:self.tree_size])
        d2 = ((keys - self.ring[i][None, :]) ** 2).sum(1)
        cand = np.argsort(d2, kind="stable")[:self.ncand]
        best, align, idx = np.finfo(np.float64).max, 0, 0
        for c in cand:
            d, s = self.distance(i, int(c))
            if d < best:
                best, align, idx = d, s, int(c)
        if best > float(self.dist_thres):
            return -1, np.float32(0), best
        yaw = np.float32(float(np.float32(np.float32(360.0) / np.float32(60.0)) * np.float32(align)) * np.pi / 180.0)     # trans::deg2rad<float>
        return idx, yaw, best
