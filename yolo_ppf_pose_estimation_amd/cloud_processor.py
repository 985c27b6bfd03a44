"""Host-side mirror of the reference's ``ppf::CloudProcessor`` (/root/reference/include/CloudProcessing.h) over the
C-ABI: the PCL stages that produce the matcher's N x 6 input (SceneCropping :263, Subsampling :361, OutlierProcessing
:341, NormalEstimation :381, EdgeExtraction :406, PointCloudXYZNormalToMat :163) followed by the PPF calls
(LoadSingleModel :209, TrainDetector :222, Matching :428, Matching_S2B :481, with the ICP step).  Clouds stay on the
device between stages (``DeviceCloud`` wraps a ``ppf_cloud*``).  Method names, argument order and defaults are the
reference's; the YOLO detector that supplies ``boxes`` is out of scope."""
from __future__ import annotations

import ctypes as C
import time
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _capi
from ._capi import IcpParams, Pose, PPFError, check, lib
from .detector import ICP, PPF3DDetector, Pose3D


class DeviceCloud:
    """A device-resident cloud (rows ``x y z nx ny nz`` + curvature)."""

    def __init__(self, ptr):
        self._ptr = ptr

    @classmethod
    def upload(cls, rows: np.ndarray, normal_offset: int = 3) -> "DeviceCloud":
        a = np.ascontiguousarray(rows, dtype=np.float32)
        if a.ndim != 2 or a.shape[1] < 3:
            raise PPFError(_capi.PPF_ERR_INVALID, "cloud must be N x 3 (xyz) or N x 6 (xyz + normal) float32")
        cols = 6 if a.shape[1] >= 6 else 3
        out = C.c_void_p()
        check(lib().ppf_cloud_upload(a.ctypes.data, a.shape[0], a.shape[1], normal_offset, cols, C.byref(out)))
        return cls(out)

    def __del__(self):
        try:
            if self._ptr:
                lib().ppf_cloud_release(self._ptr)
                self._ptr = None
        except Exception:
            pass

    def __len__(self) -> int:
        n = C.c_int(0)
        check(lib().ppf_cloud_size(self._ptr, C.byref(n)))
        return n.value

    size = __len__

    def download(self):
        """(rows (n, 6) float32, curvature (n,) float32)"""
        n = len(self)
        rows = np.zeros((n, 6), dtype=np.float32)
        curv = np.zeros(n, dtype=np.float32)
        check(lib().ppf_cloud_download(self._ptr, rows.ctypes.data, curv.ctypes.data, n))
        return rows, curv

    def rows(self) -> np.ndarray:
        return self.download()[0]

    def xyz(self) -> np.ndarray:
        return self.download()[0][:, :3].copy()

    def device_rows(self):
        """(device pointer, n): packed n x 6 rows for ppf_match_device / ppf_icp_refine_device"""
        p, n = C.c_void_p(), C.c_int(0)
        check(lib().ppf_cloud_device_rows(self._ptr, C.byref(p), C.byref(n)))
        return p.value, n.value

    def _stage(self, fn, *args) -> "DeviceCloud":
        out = C.c_void_p()
        check(fn(self._ptr, *args, C.byref(out)))
        return DeviceCloud(out)

    # the stages, one call each
    def crop(self, box, depth: np.ndarray, intr) -> "DeviceCloud":
        d = np.ascontiguousarray(depth, dtype=np.float32)
        bx = (C.c_int * 4)(*[int(v) for v in box])
        it = (C.c_double * 4)(*[float(v) for v in intr])
        return self._stage(lib().ppf_prep_crop, bx, d.ctypes.data, d.shape[0], d.shape[1], it)

    def voxel_grid(self, leaf: float) -> "DeviceCloud":
        return self._stage(lib().ppf_prep_voxel_grid, float(leaf))

    def outlier_removal(self, mean_k: int = 50, stddev_mul: float = 1.5) -> "DeviceCloud":
        return self._stage(lib().ppf_prep_outlier_removal, int(mean_k), float(stddev_mul))

    def normals(self, k: int = 30) -> "DeviceCloud":
        return self._stage(lib().ppf_prep_normals, int(k))

    def edges(self, curvature_threshold: float) -> "DeviceCloud":
        return self._stage(lib().ppf_prep_edges, C.c_float(curvature_threshold))

    def to_mat(self) -> "DeviceCloud":
        return self._stage(lib().ppf_prep_to_mat)

    def knn(self, k: int):
        n = len(self)
        idx = np.zeros((n, k), dtype=np.int32)
        d2 = np.zeros((n, k), dtype=np.float32)
        check(lib().ppf_prep_knn(self._ptr, int(k), idx.ctypes.data, d2.ctypes.data))
        return idx, d2


class CloudProcessor:
    """``ppf::CloudProcessor``: holds the scene cloud, the depth image, the detector's boxes, the per-object clouds
    and the PPF detectors; every method is the reference's, in the order its driver calls them
    (src/YOLO_cropping_ppf_test.cpp:84-123)."""

    def __init__(self, scene: Optional[np.ndarray] = None, depth: Optional[np.ndarray] = None,
                 boxes: Sequence[Sequence[int]] = (), classIds: Sequence[int] = (), indices: Sequence[int] = (),
                 relativeSamplingStep: float = 0.025, relativeDistanceStep: float = 0.05):
        self.scene = DeviceCloud.upload(scene) if scene is not None else None
        self.depth = None if depth is None else np.ascontiguousarray(depth, dtype=np.float32)
        self.boxes, self.classIds, self.indices = [tuple(b) for b in boxes], list(classIds), list(indices)
        self.relativeSamplingStep, self.relativeDistanceStep = relativeSamplingStep, relativeDistanceStep
        self.objects: List[DeviceCloud] = []
        self.objects_with_normals: List[DeviceCloud] = []
        self.objects_edges: List[DeviceCloud] = []
        self.models: List[np.ndarray] = []
        self.detectors: List[PPF3DDetector] = []
        self.if_trained: List[bool] = []
        self.label_to_id: Dict[str, int] = {}
        self.id_to_label: Dict[int, str] = {}
        self._model_clouds: Dict[int, DeviceCloud] = {}
        self.timings: Dict[str, float] = {}  # seconds spent in the last resident match / ICP call

    # ---- the PCL half -------------------------------------------------------------------------------------
    def SceneCropping(self, CameraIntr) -> List[DeviceCloud]:
        """CameraIntr: 3x3 matrix (fx, fy on the diagonal, ppx, ppy in the last column), as the reference passes it"""
        K = np.asarray(CameraIntr, dtype=np.float64)
        intr = (K[0, 0], K[1, 1], K[0, 2], K[1, 2])
        for box in self.boxes:
            self.objects.append(self.scene.crop(box, self.depth, intr))
        return self.objects

    def Subsampling(self, leafsize: float) -> List[DeviceCloud]:
        self.objects = [o.voxel_grid(leafsize) for o in self.objects]
        return self.objects

    def OutlierProcessing(self, meanK: int = 50, Thresh: float = 1.5) -> List[DeviceCloud]:
        self.objects = [o.outlier_removal(meanK, Thresh) for o in self.objects]
        return self.objects

    def NormalEstimation(self, k: int = 30) -> List[DeviceCloud]:
        self.objects_with_normals += [o.normals(k) for o in self.objects]
        return self.objects_with_normals

    def EdgeExtraction(self, curvThreshold: float) -> List[DeviceCloud]:
        self.objects_edges += [o.edges(curvThreshold) for o in self.objects_with_normals]
        return self.objects_edges

    @staticmethod
    def PointCloudXYZNormalToMat(pcl_cloud: DeviceCloud, resident: bool = False):
        """the N x 6 Mat of the reference (numpy array); ``resident=True`` keeps it on the device (a DeviceCloud that
        Matching / Matching_S2B accept directly, so no cloud crosses PCIe between the crop and the pose)"""
        mat = pcl_cloud.to_mat()
        return mat if resident else mat.rows()

    # ---- the PPF half -------------------------------------------------------------------------------------
    def LoadSingleModel(self, model_input: np.ndarray, label: str):
        self.models.append(np.ascontiguousarray(model_input, dtype=np.float32))
        idx = len(self.models) - 1
        self.if_trained.append(False)
        self.label_to_id[label], self.id_to_label[idx] = idx, label
        self.detectors.append(PPF3DDetector(self.relativeSamplingStep, self.relativeDistanceStep))

    def TrainDetector(self, relativeSamplingStep_train: float = 0.025, relativeDistanceStep_train: float = 0.5):
        for i, model in enumerate(self.models):
            self.detectors[i] = PPF3DDetector(relativeSamplingStep_train, relativeDistanceStep_train).trainModel(model)
            self.if_trained[i] = True

    def _match(self, name, scene, edge, step, dist) -> Optional[Pose3D]:
        idx = self.label_to_id[name]
        if not self.if_trained[idx]:
            raise PPFError(_capi.PPF_ERR_NOT_TRAINED, f"Model [{name}] not trained yet.")
        det = self.detectors[idx]
        if isinstance(scene, DeviceCloud):
            return self._match_resident(idx, det, scene, edge, step, dist)
        results = det.match(scene, step, dist) if edge is None else det.match_S2B(scene, edge, step, dist)
        if not results:
            return None  # the reference prints "No matching Poses found" and exits (:450-454)
        sub = results[:5]
        ICP(100, 0.005, 2.5, 8).registerModelToScene(self.models[idx], scene, sub)
        return sub[0]

    def _match_resident(self, idx, det, scene: DeviceCloud, edge: Optional[DeviceCloud], step, dist) -> Optional[Pose3D]:
        mp = det._params(step, dist, False)
        cap = len(scene) + 8
        out = (Pose * cap)()
        n = C.c_int(0)
        t0 = time.perf_counter()
        check(lib().ppf_match_clouds(det._model.ptr, scene._ptr, edge._ptr if edge is not None else None, C.byref(mp), out, cap,
                                     C.byref(n)))
        self.timings["match"] = time.perf_counter() - t0
        if n.value == 0:
            return None
        top = min(5, n.value)
        if idx not in self._model_clouds:
            self._model_clouds[idx] = DeviceCloud.upload(self.models[idx])
        prm = IcpParams()
        lib().ppf_default_icp_params(C.byref(prm))
        t0 = time.perf_counter()
        check(lib().ppf_icp_refine_clouds(self._model_clouds[idx]._ptr, scene._ptr, C.byref(prm), out, top, None))
        self.timings["icp"] = time.perf_counter() - t0
        return Pose3D(out[0])

    def Matching(self, name: str, scene: np.ndarray, relativeSceneSampleStep: float = 0.0714,
                 relativeSceneDistance: float = 0.05) -> Optional[Pose3D]:
        return self._match(name, scene, None, relativeSceneSampleStep, relativeSceneDistance)

    def Matching_S2B(self, name: str, scene: np.ndarray, edge: np.ndarray, relativeSceneSampleStep: float = 0.05,
                     relativeSceneDistance: float = 0.05) -> Optional[Pose3D]:
        return self._match(name, scene, edge, relativeSceneSampleStep, relativeSceneDistance)
