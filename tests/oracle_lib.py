"""ctypes binding of oracle/libppf_oracle.so — the CPU checker.  TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product
package.  Builds the library with oracle/Makefile when it is missing (g++ only, no GPU needed).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libppf_oracle.so")

FLAG_PRESAMPLED = 1
FLAG_DIST_FROM_DISTANCE_STEP = 4
FLAG_FEATURE_DARBOUX = 8
MODE_DET, MODE_LIBM = 0, 1


class OraclePose(C.Structure):
    _fields_ = [
        ("pose", C.c_double * 16),
        ("q", C.c_double * 4),
        ("t", C.c_double * 3),
        ("angle", C.c_double),
        ("alpha", C.c_double),
        ("residual", C.c_double),
        ("model_index", C.c_uint32),
        ("num_votes", C.c_uint32),
    ]


def build(force: bool = False) -> str:
    srcs = [os.path.join(ORACLE_DIR, "ppf_oracle.cpp"), os.path.join(ORACLE_DIR, "ppf_icp_oracle.cpp"),
            os.path.join(ORACLE_DIR, "ppf_prep_oracle.cpp"),
            os.path.join(ROOT, "include", "ppf_detmath.h")]
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(LIB_PATH) for p in srcs
    )
    if force or stale:
        subprocess.run(["make", "-C", ORACLE_DIR, "-B" if force else "-s"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None
_lib_path = LIB_PATH


def _host_tag() -> str:
    """Short tag of this machine's CPU (model name + flags): a native build is only ever loaded where it was made."""
    import hashlib
    try:
        txt = open("/proc/cpuinfo").read()
        keep = [ln for ln in txt.splitlines() if ln.startswith(("model name", "flags"))][:2]
        return hashlib.sha1("\n".join(keep).encode()).hexdigest()[:10]
    except Exception:
        return "unknown"


def use_native_build() -> bool:
    """Switch this process to the -O3 -march=native build of the oracle (oracle/Makefile target `native`), compiling
    it here if needed.  Used by bench.py's cpu_baseline leg only.  Returns False (and keeps the portable build) when
    the compile fails.  Results are bit-identical to the portable build (tests/test_oracle_native.py)."""
    global _lib, _lib_path
    tag = _host_tag()
    path = os.path.join(ORACLE_DIR, f"libppf_oracle_native_{tag}.so")
    try:
        subprocess.run(["make", "-s", "-C", ORACLE_DIR, "native", f"NATIVE_TAG={tag}"], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
    except Exception:
        return False
    if not os.path.exists(path):
        return False
    if _lib_path != path:
        _lib_path = path
        _lib = None
    return True


def lib():
    global _lib
    if _lib is None:
        if _lib_path == LIB_PATH:
            build()
        L = C.CDLL(_lib_path)
        fp = C.POINTER(C.c_float)
        L.oracle_train.restype = C.c_void_p
        L.oracle_train.argtypes = [fp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_set_search_params.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int]
        L.oracle_set_policy.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int]
        L.oracle_model_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(C.c_double),
                                        C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.oracle_model_sampled.argtypes = [C.c_void_p, fp]
        L.oracle_model_pairs.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), fp]
        L.oracle_model_bucket_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                                C.POINTER(C.c_double)]
        L.oracle_sample.restype = C.c_int
        L.oracle_sample.argtypes = [fp, C.c_int, C.c_int, C.c_double, fp, C.c_int]
        L.oracle_match.restype = C.c_int
        L.oracle_match.argtypes = [C.c_void_p, fp, C.c_int, C.c_int, fp, C.c_int, C.c_int, C.c_double, C.c_double,
                                   C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(OraclePose), C.c_int,
                                   C.POINTER(OraclePose), C.c_int, C.POINTER(C.c_int), fp, C.c_int, C.POINTER(C.c_int)]
        L.oracle_accumulator.restype = C.c_int
        L.oracle_accumulator.argtypes = [C.c_void_p, fp, C.c_int, fp, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
        L.oracle_cluster.restype = C.c_int
        L.oracle_cluster.argtypes = [C.c_void_p, C.POINTER(OraclePose), C.c_int, C.c_int, C.POINTER(OraclePose), C.c_int]
        L.oracle_murmur3_x64_128.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_uint64)]
        L.oracle_pair_feature_darboux.restype = C.c_int
        L.oracle_pair_feature_darboux.argtypes = [fp, fp, fp, fp, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double),
                                                  C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
        L.oracle_pair_feature.restype = C.c_uint32
        L.oracle_pair_feature.argtypes = [fp, fp, fp, fp, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double),
                                          C.POINTER(C.c_int32)]
        L.oracle_transform_rt.argtypes = [fp, fp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.oracle_alpha.restype = C.c_double
        L.oracle_alpha.argtypes = [fp, fp, fp, C.c_int]
        L.oracle_math_eval.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double), C.c_int]
        L.oracle_dcm_to_quat.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.oracle_quat_to_dcm.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.oracle_max_threads.restype = C.c_int
        L.oracle_icp_refine.restype = C.c_int
        L.oracle_icp_refine.argtypes = [fp, C.c_int, fp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                        C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
        ip = C.POINTER(C.c_int)
        L.oracle_prep_crop.argtypes = [fp, C.c_int, C.c_int, ip, fp, C.c_int, C.c_int, C.POINTER(C.c_double), ip, ip,
                                       C.POINTER(C.c_double)]
        L.oracle_prep_voxel.argtypes = [fp, C.c_int, C.c_int, C.c_float, fp, ip]
        L.oracle_prep_knn.argtypes = [fp, C.c_int, C.c_int, C.c_int, ip, fp]
        L.oracle_prep_sor.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_ubyte), fp,
                                      C.POINTER(C.c_double)]
        L.oracle_prep_normals.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp, fp]
        L.oracle_prep_to_mat.argtypes = [fp, fp, C.c_int, fp]
        _lib = L
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _f64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def pose_to_dict(p: OraclePose) -> dict:
    return {
        "pose": np.array(p.pose, dtype=np.float64).reshape(4, 4),
        "q": np.array(p.q, dtype=np.float64),
        "t": np.array(p.t, dtype=np.float64),
        "angle": float(p.angle),
        "alpha": float(p.alpha),
        "residual": float(p.residual),
        "model_index": int(p.model_index),
        "num_votes": int(p.num_votes),
    }


class OracleDetector:
    """CPU restatement of cv::ppf_match_3d::PPF3DDetector as the reference uses it
    (/root/reference/include/CloudProcessing.h:205-236,442,495)."""

    def __init__(self, relative_sampling_step=0.05, relative_distance_step=0.05, num_angles=30.0, mode=MODE_DET,
                 dist_from_distance_step=False):
        self.rel_sampling = float(relative_sampling_step)
        self.rel_distance = float(relative_distance_step)
        self.num_angles_arg = float(num_angles)
        self.mode = int(mode)
        self.dist_flag = FLAG_DIST_FROM_DISTANCE_STEP if dist_from_distance_step else 0
        self.h = None

    def __del__(self):
        try:
            if self.h:
                lib().oracle_free(self.h)
                self.h = None
        except Exception:
            pass

    def train_model(self, pc: np.ndarray, presampled: bool = False, darboux: bool = False):
        pc, p = _f32(pc)
        assert pc.ndim == 2 and pc.shape[1] >= 6
        flags = (FLAG_PRESAMPLED if presampled else 0) | self.dist_flag | (FLAG_FEATURE_DARBOUX if darboux else 0)
        if self.h:
            lib().oracle_free(self.h)
        self.h = lib().oracle_train(p, pc.shape[0], pc.shape[1], self.rel_sampling, self.rel_distance,
                                    self.num_angles_arg, flags, self.mode)
        if not self.h:
            raise RuntimeError("oracle_train failed")
        return self

    def set_search_params(self, position_threshold=-1.0, rotation_threshold=-1.0, use_weighted=False):
        lib().oracle_set_search_params(self.h, position_threshold, rotation_threshold, int(use_weighted))

    def set_policy(self, key_exact=False, pair_radius=0.0, rot_relative=False, alpha_2pi=False):
        """PCL-semantics switches (oracle/ppf_oracle.cpp: Model::key_exact / pair_radius / rot_relative / alpha_2pi)."""
        lib().oracle_set_policy(self.h, int(key_exact), float(pair_radius), int(rot_relative), int(alpha_2pi))
        return self

    def info(self) -> dict:
        n = C.c_int(); s = C.c_uint32(); a = C.c_double(); d = C.c_double(); na = C.c_int()
        lib().oracle_model_info(self.h, C.byref(n), C.byref(s), C.byref(a), C.byref(d), C.byref(na))
        return {"n_ref": n.value, "slots": s.value, "angle_step": a.value, "distance_step": d.value,
                "num_angles": na.value}

    def sampled_model(self) -> np.ndarray:
        n = self.info()["n_ref"]
        out = np.empty((n, 6), dtype=np.float32)
        lib().oracle_model_sampled(self.h, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def pairs(self):
        n = self.info()["n_ref"]
        hsh = np.empty(n * n, dtype=np.uint32)
        alp = np.empty(n * n, dtype=np.float32)
        lib().oracle_model_pairs(self.h, hsh.ctypes.data_as(C.POINTER(C.c_uint32)),
                                 alp.ctypes.data_as(C.POINTER(C.c_float)))
        return hsh.reshape(n, n), alp.reshape(n, n)

    def bucket_stats(self) -> dict:
        a = C.c_uint64(); b = C.c_uint64(); s = C.c_double()
        lib().oracle_model_bucket_stats(self.h, C.byref(a), C.byref(b), C.byref(s))
        return {"non_empty": a.value, "max_len": b.value, "sum_sq": s.value}

    def match(self, scene, edge=None, relative_scene_sample_step=1.0 / 40.0, relative_scene_distance=0.05,
              presampled=False, ref_list=None, threads=0, cluster=True, max_final=None) -> dict:
        scene, sp = _f32(scene)
        ns = scene.shape[0]
        if edge is not None:
            edge, ep = _f32(edge)
            ne, es = edge.shape[0], edge.shape[1]
        else:
            ep, ne, es = None, 0, 6
        flags = FLAG_PRESAMPLED if presampled else 0
        step = int(1.0 / relative_scene_sample_step)
        cap_rows = ns if presampled else max(ns, 1)
        if ref_list is not None:
            ref_arr = np.ascontiguousarray(ref_list, dtype=np.int32)
            rp, nrl = ref_arr.ctypes.data_as(C.POINTER(C.c_int)), ref_arr.shape[0]
            cap_ref = nrl
        else:
            rp, nrl = None, 0
            cap_ref = cap_rows // max(step, 1) + 4
        triples = np.zeros((cap_ref, 3), dtype=np.uint32)
        votes = np.zeros(cap_ref, dtype=np.uint64)
        pairs = np.zeros(cap_ref, dtype=np.uint64)
        raw = (OraclePose * cap_ref)()
        fin_cap = cap_ref if max_final is None else max_final
        fin = (OraclePose * max(fin_cap, 1))()
        nfin = C.c_int(0)
        sampled = np.zeros((cap_rows, 6), dtype=np.float32)
        nsamp = C.c_int(0)
        nref = lib().oracle_match(
            self.h, sp, ns, scene.shape[1], ep, ne, es, float(relative_scene_sample_step),
            float(relative_scene_distance), flags, rp, nrl, int(threads),
            triples.ctypes.data_as(C.POINTER(C.c_uint32)), votes.ctypes.data_as(C.POINTER(C.c_uint64)),
            pairs.ctypes.data_as(C.POINTER(C.c_uint64)), raw, cap_ref,
            fin if cluster else None, fin_cap, C.byref(nfin) if cluster else None,
            sampled.ctypes.data_as(C.POINTER(C.c_float)), cap_rows, C.byref(nsamp))
        if nref < 0:
            raise RuntimeError("oracle_match failed")
        return {
            "n_ref": nref,
            "triples": triples[:nref].copy(),
            "votes_per_ref": votes[:nref].copy(),
            "pairs_per_ref": pairs[:nref].copy(),
            "raw_poses": [pose_to_dict(raw[i]) for i in range(nref)],
            "poses": [pose_to_dict(fin[i]) for i in range(min(nfin.value, fin_cap))] if cluster else [],
            "n_final": nfin.value,
            "sampled_scene": sampled[: nsamp.value].copy(),
        }

    def accumulator(self, sampled_scene, i, sampled_paired=None) -> np.ndarray:
        info = self.info()
        sc, sp = _f32(sampled_scene)
        acc = np.zeros(info["num_angles"] * info["n_ref"], dtype=np.uint32)
        if sampled_paired is not None:
            pr, pp = _f32(sampled_paired)
            npair = pr.shape[0]
        else:
            pp, npair = None, 0
        lib().oracle_accumulator(self.h, sp, sc.shape[0], pp, npair, int(i), acc.ctypes.data_as(C.POINTER(C.c_uint32)))
        return acc.reshape(info["n_ref"], info["num_angles"])


def sample(pc, rel_step) -> np.ndarray:
    pc, p = _f32(pc)
    out = np.zeros((pc.shape[0], 6), dtype=np.float32)
    rows = lib().oracle_sample(p, pc.shape[0], pc.shape[1], float(rel_step), out.ctypes.data_as(C.POINTER(C.c_float)),
                               pc.shape[0])
    return out[:rows].copy()


def murmur3_x64_128(data: bytes, seed: int):
    out = (C.c_uint64 * 2)()
    buf = C.create_string_buffer(data, len(data))
    lib().oracle_murmur3_x64_128(buf, len(data), seed, out)
    return int(out[0]), int(out[1])


def pair_feature(p1, n1, p2, n2, angle_step, dist_step, mode=MODE_DET):
    a = [_f32(v) for v in (p1, n1, p2, n2)]
    f = np.zeros(4, dtype=np.float64)
    key = np.zeros(4, dtype=np.int32)
    h = lib().oracle_pair_feature(a[0][1], a[1][1], a[2][1], a[3][1], angle_step, dist_step, mode, _f64p(f),
                                  key.ctypes.data_as(C.POINTER(C.c_int32)))
    return f, key, int(h)


def pair_feature_darboux(p1, n1, p2, n2, angle_step, dist_step, mode=MODE_DET):
    """(f, key, hash) of PCL's pair feature, or None for a degenerate pair."""
    a = [_f32(v) for v in (p1, n1, p2, n2)]
    f = np.zeros(4, dtype=np.float64)
    key = np.zeros(4, dtype=np.int32)
    h = C.c_uint32()
    ok = lib().oracle_pair_feature_darboux(a[0][1], a[1][1], a[2][1], a[3][1], angle_step, dist_step, mode, _f64p(f),
                                           key.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(h))
    return (f, key, int(h.value)) if ok else None


def transform_rt(p, n, mode=MODE_DET):
    pa, pp = _f32(p)
    na, np_ = _f32(n)
    R = np.zeros(9, dtype=np.float64)
    t = np.zeros(3, dtype=np.float64)
    lib().oracle_transform_rt(pp, np_, mode, _f64p(R), _f64p(t))
    return R.reshape(3, 3), t


def alpha(p1, n1, p2, mode=MODE_DET) -> float:
    a = [_f32(v) for v in (p1, n1, p2)]
    return float(lib().oracle_alpha(a[0][1], a[1][1], a[2][1], mode))


def math_eval(fn: str, x, x2=None, mode=MODE_DET) -> np.ndarray:
    code = {"acos": 0, "sin": 1, "cos": 2, "atan2": 3}[fn]
    x = np.ascontiguousarray(x, dtype=np.float64)
    x2a = np.ascontiguousarray(x2 if x2 is not None else np.zeros_like(x), dtype=np.float64)
    out = np.empty_like(x)
    lib().oracle_math_eval(code, mode, _f64p(x), _f64p(x2a), _f64p(out), x.size)
    return out


def dcm_to_quat(R) -> np.ndarray:
    R = np.ascontiguousarray(R, dtype=np.float64).reshape(9)
    q = np.zeros(4, dtype=np.float64)
    lib().oracle_dcm_to_quat(_f64p(R), _f64p(q))
    return q


def quat_to_dcm(q) -> np.ndarray:
    q = np.ascontiguousarray(q, dtype=np.float64)
    R = np.zeros(9, dtype=np.float64)
    lib().oracle_quat_to_dcm(_f64p(q), _f64p(R))
    return R.reshape(3, 3)


def max_threads() -> int:
    return int(lib().oracle_max_threads())


def icp_refine(model, scene, poses, iterations=100, tolerance=0.005, rejection_scale=2.5, num_levels=8):
    """ICP(iterations, tolerance, rejection_scale, num_levels).registerModelToScene(model, scene, poses):
    returns (refined poses (k,4,4), residuals (k,), iterations used (k,))."""
    m, mp = _f32(np.ascontiguousarray(np.asarray(model)[:, :6]))
    s, sp = _f32(np.ascontiguousarray(np.asarray(scene)[:, :6]))
    P = np.ascontiguousarray(np.asarray(poses, dtype=np.float64).reshape(-1, 16)).copy()
    res = np.zeros(P.shape[0], dtype=np.float64)
    its = np.zeros(P.shape[0], dtype=np.int32)
    lib().oracle_icp_refine(mp, m.shape[0], sp, s.shape[0], int(iterations), float(tolerance), float(rejection_scale),
                            int(num_levels), _f64p(P), _f64p(res), P.shape[0], its.ctypes.data_as(C.POINTER(C.c_int)))
    return P.reshape(-1, 4, 4), res, its


def icp_refine_traced(model, scene, poses, **kw):
    """icp_refine plus the oracle's per-pass trace: rows {level, iteration, source rows, scene rows, accepted by the
    rejection threshold, kept by picky ICP, exit code (0 iterated, 1 six or fewer correspondences, 2 solve failed, 3 NaN)}."""
    buf = np.zeros((4096, 7), dtype=np.int32)
    L = lib()
    L.oracle_icp_set_trace.restype = C.c_int
    L.oracle_icp_set_trace(buf.ctypes.data_as(C.POINTER(C.c_int)), buf.shape[0])
    try:
        out = icp_refine(model, scene, poses, **kw)
    finally:
        n = L.oracle_icp_set_trace(None, 0)
    return out + (buf[:n].copy(),)


# ---- pre-processing stages (oracle/ppf_prep_oracle.cpp) ------------------------------------------------------------
def _xyz(cloud):
    a = np.ascontiguousarray(np.asarray(cloud, dtype=np.float32))
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def prep_crop(cloud, box, depth, intr):
    """SceneCropping for one box = (x, y, w, h); intr = (fx, fy, ppx, ppy).  Returns (kept indices, planes(13,))."""
    a, ap = _xyz(cloud)
    d = np.ascontiguousarray(depth, dtype=np.float32)
    bx = (C.c_int * 4)(*[int(v) for v in box])
    it = (C.c_double * 4)(*[float(v) for v in intr])
    keep = np.zeros(a.shape[0], dtype=np.int32)
    n = C.c_int(0)
    planes = (C.c_double * 13)()
    lib().oracle_prep_crop(ap, a.shape[0], a.shape[1], bx, d.ctypes.data_as(C.POINTER(C.c_float)), d.shape[0], d.shape[1], it,
                           keep.ctypes.data_as(C.POINTER(C.c_int)), C.byref(n), planes)
    return keep[: n.value].copy(), np.array(planes)


def prep_voxel(cloud, leaf):
    a, ap = _xyz(cloud)
    out = np.zeros((a.shape[0], 3), dtype=np.float32)
    n = C.c_int(0)
    rc = lib().oracle_prep_voxel(ap, a.shape[0], a.shape[1], float(leaf), out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n))
    if rc:
        raise ValueError("leaf size too small")
    return out[: n.value].copy()


def prep_knn(cloud, k):
    a, ap = _xyz(cloud)
    idx = np.zeros((a.shape[0], k), dtype=np.int32)
    d2 = np.zeros((a.shape[0], k), dtype=np.float32)
    lib().oracle_prep_knn(ap, a.shape[0], a.shape[1], int(k), idx.ctypes.data_as(C.POINTER(C.c_int)),
                          d2.ctypes.data_as(C.POINTER(C.c_float)))
    return idx, d2


def prep_sor(cloud, mean_k=50, std_mul=1.5):
    """StatisticalOutlierRemoval: returns (keep mask, mean neighbour distances, threshold)."""
    a, ap = _xyz(cloud)
    keep = np.zeros(a.shape[0], dtype=np.uint8)
    dist = np.zeros(a.shape[0], dtype=np.float32)
    thr = C.c_double(0)
    lib().oracle_prep_sor(ap, a.shape[0], a.shape[1], int(mean_k), float(std_mul), keep.ctypes.data_as(C.POINTER(C.c_ubyte)),
                          dist.ctypes.data_as(C.POINTER(C.c_float)), C.byref(thr))
    return keep.astype(bool), dist, thr.value


def prep_normals(cloud, k=30):
    a, ap = _xyz(cloud)
    nrm = np.zeros((a.shape[0], 3), dtype=np.float32)
    curv = np.zeros(a.shape[0], dtype=np.float32)
    lib().oracle_prep_normals(ap, a.shape[0], a.shape[1], int(k), nrm.ctypes.data_as(C.POINTER(C.c_float)),
                              curv.ctypes.data_as(C.POINTER(C.c_float)))
    return nrm, curv


def prep_to_mat(xyz, normals):
    a, ap = _xyz(np.asarray(xyz)[:, :3])
    b, bp = _xyz(normals)
    out = np.zeros((a.shape[0], 6), dtype=np.float32)
    lib().oracle_prep_to_mat(ap, bp, a.shape[0], out.ctypes.data_as(C.POINTER(C.c_float)))
    return out
