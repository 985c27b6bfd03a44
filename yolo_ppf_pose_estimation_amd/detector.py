"""Host-side mirror of the operator interface the reference uses for the PPF path.

Names, argument meaning and error behaviour follow ``cv::ppf_match_3d::PPF3DDetector`` exactly as
/root/reference/include/CloudProcessing.h calls it:

    PPF3DDetector(relativeSamplingStep, relativeDistanceStep)            :205,217,234
    detector.trainModel(model N x 6 f32)                                 :236
    detector.match(scene, results, relativeSceneSampleStep, relSceneDist):442
    detector.match_S2B(scene, edge, results, step, dist)                 :495
    detector.read(FileNode) / detector.write(FileStorage)                :112 / :250
    Pose3D.pose, Pose3D.printPose()                 src/YOLO_cropping_ppf_test.cpp:124-125

All compute goes through the C-ABI of libppf_hip.so (HIP kernels on gfx950).  Nothing here falls
back to numpy/CPU: without the built extension the import of the library raises.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np

from . import _capi
from ._capi import IcpParams, MatchParams, MatchStats, ModelInfo, Pose, PPFError, TrainParams, Vote, check, lib


class Pose3D:
    """cv::ppf_match_3d::Pose3D: 4x4 ``pose`` (model -> scene), quaternion ``q`` [w x y z], ``t``,
    rotation ``angle``, ``alpha``, ``modelIndex``, ``numVotes``, ``residual``."""

    __slots__ = ("pose", "q", "t", "angle", "alpha", "residual", "modelIndex", "numVotes")

    def __init__(self, rec: Optional[Pose] = None):
        if rec is None:
            self.pose = np.eye(4)
            self.q = np.array([1.0, 0, 0, 0])
            self.t = np.zeros(3)
            self.angle = self.alpha = self.residual = 0.0
            self.modelIndex = self.numVotes = 0
        else:
            self.pose = np.array(rec.pose, dtype=np.float64).reshape(4, 4)
            self.q = np.array(rec.q, dtype=np.float64)
            self.t = np.array(rec.t, dtype=np.float64)
            self.angle, self.alpha, self.residual = float(rec.angle), float(rec.alpha), float(rec.residual)
            self.modelIndex, self.numVotes = int(rec.model_index), int(rec.num_votes)

    def to_record(self) -> Pose:
        rec = Pose()
        rec.pose[:] = self.pose.reshape(16).tolist()
        rec.q[:] = self.q.tolist()
        rec.t[:] = self.t.tolist()
        rec.angle, rec.alpha, rec.residual = self.angle, self.alpha, self.residual
        rec.model_index, rec.num_votes = self.modelIndex, self.numVotes
        return rec

    def printPose(self):
        print(f"\n-- Pose to Model Index {self.modelIndex}: NumVotes = {self.numVotes}, Residual = {self.residual}")
        print(self.pose)

    def clone(self) -> "Pose3D":
        return Pose3D(self.to_record())


def _cloud(a, name, normal_offset: int = 3):
    """A cloud argument: float32 rows with x y z first and the normal `normal_offset` floats into the row (3 for the
    reference's N x 6 Mat, 4 for rows laid out like pcl::PointNormal); the row pitch is the array's width."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] < 6 or a.shape[0] < 1 or normal_offset < 3 or normal_offset + 3 > a.shape[1]:
        raise PPFError(_capi.PPF_ERR_INVALID, f"{name} must be an N x 6 float32 cloud (x y z nx ny nz)")
    return a


class _ModelHandle:
    """Ref-counted owner of a ppf_model*: detector copies share it (the reference copies its
    detector by value before every match, CloudProcessing.h:432,485)."""

    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr:
                lib().ppf_model_release(self.ptr)
                self.ptr = None
        except Exception:
            pass


class PPF3DDetector:
    def __init__(self, relativeSamplingStep: float = 0.05, relativeDistanceStep: float = 0.05, numAngles: float = 30,
                 *, distance_from_distance_step: bool = False, max_tile_refs: int = 0, key_equality: int = 0,
                 feature: int = 0):
        self.sampling_step_relative = float(relativeSamplingStep)
        self.distance_step_relative = float(relativeDistanceStep)
        self.angle_step_relative = float(numAngles)
        self._dist_flag = bool(distance_from_distance_step)
        self._max_tile_refs = int(max_tile_refs)
        self._key_equality = int(key_equality)  # 0: whole hash bucket votes (OpenCV); 1: exact quantised key (PCL)
        self._feature = int(feature)            # 0: three acos angles (OpenCV); 1: Darboux-frame feature, floor keys (PCL)
        self._pair_radius = 0.0                 # > 0: scene pairs within this distance only (PCL)
        self._rot_metric_relative = False       # cluster on the relative rotation angle (PCL)
        self._alpha_range_2pi = False           # alpha differences wrapped into [-pi, pi], binned over 2 pi (PCL)
        self._position_threshold = -1.0
        self._rotation_threshold = -1.0
        self._use_weighted_avg = False
        self._model: Optional[_ModelHandle] = None
        self.last_stats: Optional[dict] = None

    # -- copy semantics: share the trained table ------------------------------------------------
    def __copy__(self):
        d = PPF3DDetector(self.sampling_step_relative, self.distance_step_relative, self.angle_step_relative,
                          distance_from_distance_step=self._dist_flag, max_tile_refs=self._max_tile_refs,
                          key_equality=self._key_equality, feature=self._feature)
        d._pair_radius, d._rot_metric_relative = self._pair_radius, self._rot_metric_relative
        d._alpha_range_2pi = self._alpha_range_2pi
        d._position_threshold, d._rotation_threshold = self._position_threshold, self._rotation_threshold
        d._use_weighted_avg = self._use_weighted_avg
        d._model = self._model
        return d

    @property
    def trained(self) -> bool:
        return self._model is not None

    def setSearchParams(self, positionThreshold: float = -1, rotationThreshold: float = -1,
                        useWeightedClustering: bool = False):
        self._position_threshold = float(positionThreshold)
        self._rotation_threshold = float(rotationThreshold)
        self._use_weighted_avg = bool(useWeightedClustering)

    def setPolicy(self, pair_radius: float = 0.0, rot_metric_relative: bool = False, alpha_range_2pi: bool = False):
        """PCL-semantics switches of the match (ppf_match_params.pair_radius / rot_metric_relative); the key policy is a
        constructor argument because it shapes the trained table."""
        self._pair_radius = float(pair_radius)
        self._rot_metric_relative = bool(rot_metric_relative)
        self._alpha_range_2pi = bool(alpha_range_2pi)
        return self

    # -- training ---------------------------------------------------------------------------------
    def trainModel(self, model: np.ndarray, presampled: bool = False, *, normal_offset: int = 3):
        pc = _cloud(model, "model", normal_offset)
        tp = TrainParams()
        lib().ppf_default_train_params(C.byref(tp))
        tp.relative_sampling_step = self.sampling_step_relative
        tp.relative_distance_step = self.distance_step_relative
        tp.num_angles = self.angle_step_relative
        tp.presampled = int(presampled)
        tp.distance_from_distance_step = int(self._dist_flag)
        tp.max_tile_refs = self._max_tile_refs
        tp.key_equality = self._key_equality
        tp.feature = self._feature
        out = C.c_void_p()
        check(lib().ppf_model_train(pc.ctypes.data, pc.shape[0], pc.shape[1], normal_offset, C.byref(tp), C.byref(out)))
        self._model = _ModelHandle(out.value)
        return self

    def device(self) -> int:
        """HIP device the trained table lives on."""
        self._require_trained()
        d = C.c_int(-1)
        check(lib().ppf_model_get_device(self._model.ptr, C.byref(d)))
        return d.value

    def nearest_pairs(self, f4) -> np.ndarray:
        """pcl::PPFHashMapSearch::nearestNeighborSearch: the (i, j) pairs of the sampled model whose quantised feature equals
        the quantised ``f4`` (four floats, the model's feature kind), ascending, as an (n, 2) uint32 array."""
        self._require_trained()
        f = (C.c_float * 4)(*[float(v) for v in f4])
        n = C.c_int(0)
        check(lib().ppf_model_nearest_pairs(self._model.ptr, f, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 2), dtype=np.uint32)
        check(lib().ppf_model_nearest_pairs(self._model.ptr, f, out.ctypes.data_as(C.POINTER(C.c_uint32)), out.shape[0], C.byref(n)))
        return out[:n.value]

    def trim_contexts(self, keep: int = 0) -> int:
        """Release the idle warm contexts of the host-buffer entries beyond ``keep`` (each holds the scratch of its last
        call); returns how many were released."""
        self._require_trained()
        n = C.c_int(0)
        check(lib().ppf_model_trim_contexts(self._model.ptr, int(keep), C.byref(n)))
        return n.value

    def info(self) -> dict:
        self._require_trained()
        mi = ModelInfo()
        check(lib().ppf_model_get_info(self._model.ptr, C.byref(mi)))
        return {k: getattr(mi, k) for k, _ in ModelInfo._fields_}

    def sampled_model(self) -> np.ndarray:
        n = self.info()["n_ref"]
        out = np.empty((n, 6), dtype=np.float32)
        check(lib().ppf_model_get_sampled(self._model.ptr, out.ctypes.data, n))
        return out

    def table(self) -> dict:
        """CSR dump of the device table (inspection / tests)."""
        mi = self.info()
        nb, ne, T = mi["n_buckets"], mi["n_entries"], mi["n_tiles"]
        slot = np.empty(nb, dtype=np.uint32)
        off = np.empty(T * (nb + 1), dtype=np.uint32)
        cell = np.empty(ne, dtype=np.int32)
        alpha = np.empty(ne, dtype=np.float32)
        check(lib().ppf_model_get_table(self._model.ptr, slot.ctypes.data, off.ctypes.data, cell.ctypes.data,
                                        alpha.ctypes.data))
        return {"bucket_slot": slot, "bucket_off": off.reshape(T, nb + 1), "entry_cell": cell, "entry_alpha": alpha}

    # -- (de)serialisation: detector.write / detector.read -----------------------------------------
    def write(self, path: str):
        self._require_trained()
        check(lib().ppf_model_save(self._model.ptr, str(path).encode()))

    def read(self, path: str):
        out = C.c_void_p()
        check(lib().ppf_model_load(str(path).encode(), C.byref(out)))
        self._model = _ModelHandle(out.value)
        mi = self.info()
        return self

    def to_bytes(self) -> bytes:
        """The model file's byte stream in memory (ppf_model_save_mem)."""
        self._require_trained()
        n = C.c_size_t(0)
        check(lib().ppf_model_save_mem(self._model.ptr, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        check(lib().ppf_model_save_mem(self._model.ptr, buf, n.value, C.byref(n)))
        return buf.raw[: n.value]

    def from_bytes(self, data: bytes):
        """ppf_model_load_mem: the same validation as read()."""
        out = C.c_void_p()
        check(lib().ppf_model_load_mem(data, len(data), C.byref(out)))
        self._model = _ModelHandle(out.value)
        return self

    # -- matching -----------------------------------------------------------------------------------
    def _require_trained(self):
        if self._model is None:
            raise PPFError(_capi.PPF_ERR_NOT_TRAINED, "The model is not trained. Cannot match without training")

    def _params(self, step, dist, presampled, ref_offset=0, ref_stride=1, skip_clustering=False,
                vote_mode: int = 0) -> MatchParams:
        mp = MatchParams()
        lib().ppf_default_match_params(C.byref(mp))
        mp.relative_scene_sample_step = float(step)
        mp.relative_scene_distance = float(dist)
        mp.position_threshold = self._position_threshold
        mp.rotation_threshold = self._rotation_threshold
        mp.use_weighted_avg = int(self._use_weighted_avg)
        mp.presampled = int(presampled)
        mp.ref_offset, mp.ref_stride = int(ref_offset), int(ref_stride)
        mp.skip_clustering = int(skip_clustering)
        mp.vote_mode = int(vote_mode)  # 0: count tables for runs of many hits, 1: one atomic per (entry, hit)
        mp.pair_radius = self._pair_radius
        mp.rot_metric_relative = int(self._rot_metric_relative)
        mp.alpha_range_2pi = int(self._alpha_range_2pi)
        return mp

    def match(self, scene: np.ndarray, relativeSceneSampleStep: float = 1.0 / 5.0,
              relativeSceneDistance: float = 0.03, *, presampled: bool = False, edge: Optional[np.ndarray] = None,
              max_results: Optional[int] = None, normal_offset: int = 3) -> List[Pose3D]:
        """detector.match(): clustered poses sorted by votes (descending)."""
        self._require_trained()
        sc = _cloud(scene, "scene", normal_offset)
        ed = _cloud(edge, "edge", normal_offset) if edge is not None else None
        mp = self._params(relativeSceneSampleStep, relativeSceneDistance, presampled)
        cap = max(sc.shape[0] // max(int(1.0 / relativeSceneSampleStep), 1) + 8, 8)
        out = (Pose * cap)()
        n = C.c_int(0)
        check(lib().ppf_match(self._model.ptr, sc.ctypes.data, sc.shape[0], sc.shape[1], normal_offset,
                              ed.ctypes.data if ed is not None else None, ed.shape[0] if ed is not None else 0,
                              ed.shape[1] if ed is not None else 6, normal_offset if ed is not None else 3,
                              C.byref(mp), out, cap, C.byref(n)))
        res = [Pose3D(out[i]) for i in range(n.value)]
        return res if max_results is None else res[:max_results]

    def match_S2B(self, scene: np.ndarray, edge: np.ndarray, relativeSceneSampleStep: float = 0.05,
                  relativeSceneDistance: float = 0.05, *, presampled: bool = False, normal_offset: int = 3) -> List[Pose3D]:
        """detector.match_S2B(): reference points from the surface cloud, paired points from the
        edge cloud (the reference's own definition is not recoverable)."""
        return self.match(scene, relativeSceneSampleStep, relativeSceneDistance, presampled=presampled, edge=edge,
                          normal_offset=normal_offset)

    def raw_votes(self, scene: np.ndarray, relativeSceneSampleStep: float = 1.0 / 5.0,
                  relativeSceneDistance: float = 0.03, *, presampled: bool = False,
                  edge: Optional[np.ndarray] = None, ref_offset: int = 0, ref_stride: int = 1,
                  vote_mode: int = 0, normal_offset: int = 3) -> dict:
        """Per-reference-point argmax triples {refIndMax, alphaIndMax, maxVotes} + raw poses +
        exact counters: the bit-exact parity surface."""
        self._require_trained()
        sc = _cloud(scene, "scene", normal_offset)
        ed = _cloud(edge, "edge", normal_offset) if edge is not None else None
        mp = self._params(relativeSceneSampleStep, relativeSceneDistance, presampled, ref_offset, ref_stride, True,
                          vote_mode)
        cap = sc.shape[0] + 8
        votes = (Vote * cap)()
        poses = (Pose * cap)()
        n = C.c_int(0)
        st = MatchStats()
        check(lib().ppf_raw_votes(self._model.ptr, sc.ctypes.data, sc.shape[0], sc.shape[1], normal_offset,
                                  ed.ctypes.data if ed is not None else None, ed.shape[0] if ed is not None else 0,
                                  ed.shape[1] if ed is not None else 6, normal_offset if ed is not None else 3,
                                  C.byref(mp), votes, poses, cap, C.byref(n), C.byref(st)))
        nr = n.value
        triples = np.array([[votes[i].ref_ind_max, votes[i].alpha_ind_max, votes[i].max_votes] for i in range(nr)],
                           dtype=np.uint32).reshape(nr, 3)
        self.last_stats = _capi.stats_dict(st)
        return {"n_ref": nr, "triples": triples, "raw_poses": [Pose3D(poses[i]) for i in range(nr)],
                "stats": self.last_stats}

    def accumulators(self, scene: np.ndarray, relativeSceneSampleStep: float, *, edge: Optional[np.ndarray] = None,
                     ref_offset: int = 0, ref_stride: int = 1, vote_mode: int = 0) -> np.ndarray:
        """Full Hough accumulators (n_ref, N_m, numAngles) of presampled clouds (debug / parity)."""
        self._require_trained()
        sc = _cloud(scene, "scene")
        ed = _cloud(edge, "edge") if edge is not None else None
        mp = self._params(relativeSceneSampleStep, 0.05, True, ref_offset, ref_stride, True, vote_mode)
        mi = self.info()
        step = int(1.0 / relativeSceneSampleStep)
        n_tot = (sc.shape[0] + step - 1) // step
        nr = max((n_tot - ref_offset + ref_stride - 1) // ref_stride, 0)
        acc = np.zeros((max(nr, 1), mi["n_ref"], mi["num_angles"]), dtype=np.uint32)
        n = C.c_int(0)
        check(lib().ppf_debug_accumulators(self._model.ptr, sc.ctypes.data, sc.shape[0], sc.shape[1], 3,
                                           ed.ctypes.data if ed is not None else None,
                                           ed.shape[0] if ed is not None else 0, ed.shape[1] if ed is not None else 6, 3,
                                           C.byref(mp), acc.ctypes.data, acc.size, C.byref(n)))
        return acc[: n.value]

    def cluster(self, poses: List[Pose3D], num_poses: Optional[int] = None) -> List[Pose3D]:
        """clusterPoses() on a caller-supplied list (e.g. gathered from several ranks)."""
        self._require_trained()
        n = len(poses)
        arr = (Pose * max(n, 1))()
        for i, p in enumerate(poses):
            arr[i] = p.to_record()
        out = (Pose * max(n, 1))()
        nout = C.c_int(0)
        mp = self._params(1.0, 0.05, True)
        check(lib().ppf_cluster_poses(self._model.ptr, arr, n, n if num_poses is None else int(num_poses),
                                      C.byref(mp), out, max(n, 1), C.byref(nout)))
        return [Pose3D(out[i]) for i in range(nout.value)]


def match_batch(detectors: List[PPF3DDetector], scenes: List[np.ndarray], relativeSceneSampleStep: float = 1.0 / 5.0,
                relativeSceneDistance: float = 0.03, *, presampled: bool = False, top_k: int = 5) -> List[List[List[Pose3D]]]:
    """ppf_match_batch: result[c][k] = best `top_k` poses of crop c against detectors[k] (BASELINE config C5)."""
    for d in detectors:
        d._require_trained()
    clouds = [_cloud(sc, "scene") for sc in scenes]
    stride = clouds[0].shape[1]
    if any(c.shape[1] != stride for c in clouds):
        raise PPFError(_capi.PPF_ERR_INVALID, "all scenes must share one row pitch")
    nm, nc = len(detectors), len(clouds)
    models = (C.c_void_p * nm)(*[d._model.ptr for d in detectors])
    ptrs = (C.c_void_p * nc)(*[c.ctypes.data for c in clouds])
    ns = (C.c_int * nc)(*[c.shape[0] for c in clouds])
    mp = detectors[0]._params(relativeSceneSampleStep, relativeSceneDistance, presampled)
    out = (Pose * (nc * nm * top_k))()
    n_out = (C.c_int * (nc * nm))()
    check(lib().ppf_match_batch(models, nm, ptrs, ns, stride, 3, nc, C.byref(mp), out, top_k, n_out))
    return [[[Pose3D(out[(c * nm + k) * top_k + i]) for i in range(n_out[c * nm + k])] for k in range(nm)]
            for c in range(nc)]


def samplePCByQuantization(pc: np.ndarray, relative_step: float, *, normal_offset: int = 3) -> np.ndarray:
    a = _cloud(pc, "cloud", normal_offset)
    n = C.c_int(0)
    out = np.empty((a.shape[0], 6), dtype=np.float32)
    check(lib().ppf_sample_cloud(a.ctypes.data, a.shape[0], a.shape[1], normal_offset, float(relative_step), out.ctypes.data,
                                 a.shape[0], C.byref(n)))
    return out[: n.value].copy()


def pairFeatures(pc: np.ndarray, feature: int = _capi.PPF_FEATURE_DARBOUX, *, normal_offset: int = 3) -> np.ndarray:
    """pcl::PPFEstimation::compute (ppf_pair_features): (N, N, 5) float32 = f1..f4, alpha_m of every ordered pair; NaN rows for
    i == j and degenerate pairs."""
    a = _cloud(pc, "cloud", normal_offset)
    n = a.shape[0]
    out = np.empty((n, n, 5), dtype=np.float32)
    check(lib().ppf_pair_features(a.ctypes.data, n, a.shape[1], normal_offset, int(feature), out.ctypes.data, n * n))
    return out


def transformPCPose(pc: np.ndarray, pose: np.ndarray, *, normal_offset: int = 3) -> np.ndarray:
    a = _cloud(pc, "cloud", normal_offset)
    T = np.ascontiguousarray(pose, dtype=np.float64).reshape(16)
    out = np.empty((a.shape[0], 6), dtype=np.float32)
    check(lib().ppf_transform_pc_pose(a.ctypes.data, a.shape[0], a.shape[1], normal_offset,
                                      T.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data))
    return out


class ICP:
    """cv::ppf_match_3d::ICP as the reference uses it right after the match
    (``ICP icp(100, 0.005f, 2.5f, 8); icp.registerModelToScene(models[id], scene, resultsSub);``
    /root/reference/include/CloudProcessing.h:465-470, :518-523): multi-level point-to-plane ICP with picky
    correspondences and median+MAD rejection, on the device (ppf_icp_refine / ppf_icp_register)."""

    def __init__(self, iterations: int = 100, tolerance: float = 0.005, rejectionScale: float = 2.5, numLevels: int = 8,
                 *, flags: int = 0):
        self._prm = IcpParams()
        lib().ppf_default_icp_params(C.byref(self._prm))
        self._prm.flags = int(flags)  # _capi.PPF_ICP_NO_SMALL_LEVELS | PPF_ICP_ONE_STREAM: other schedules, same results
        self._prm.iterations, self._prm.tolerance = int(iterations), float(tolerance)
        self._prm.rejection_scale, self._prm.num_levels = float(rejectionScale), int(numLevels)
        self.last_iterations: List[int] = []

    def registerModelToScene(self, srcPC: np.ndarray, dstPC: np.ndarray, poses: Optional[List[Pose3D]] = None):
        """With ``poses``: refines every pose in place (pose <- poseICP * pose, residual set) and returns the list.
        Without: registers src to dst from the identity and returns ``(residual, pose4x4)``."""
        src, dst = _cloud(srcPC, "srcPC"), _cloud(dstPC, "dstPC")
        if poses is None:
            pose = (C.c_double * 16)()
            res, it = C.c_double(0), C.c_int(0)
            check(lib().ppf_icp_register(src.ctypes.data, src.shape[0], src.shape[1], 3, dst.ctypes.data, dst.shape[0],
                                         dst.shape[1], 3, C.byref(self._prm), pose, C.byref(res), C.byref(it)))
            self.last_iterations = [it.value]
            return res.value, np.array(pose, dtype=np.float64).reshape(4, 4)
        n = len(poses)
        recs = (Pose * max(n, 1))()
        for i, p in enumerate(poses):
            recs[i] = p.to_record()
        iters = (C.c_int * max(n, 1))()
        check(lib().ppf_icp_refine(src.ctypes.data, src.shape[0], src.shape[1], 3, dst.ctypes.data, dst.shape[0], dst.shape[1], 3,
                                   C.byref(self._prm), recs, n, iters))
        self.last_iterations = [iters[i] for i in range(n)]
        for i, p in enumerate(poses):
            r = Pose3D(recs[i])
            p.pose, p.q, p.t, p.angle, p.residual = r.pose, r.q, r.t, r.angle, r.residual
        return poses
