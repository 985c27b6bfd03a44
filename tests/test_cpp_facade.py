"""The header-only C++ facade (include/ppf_match_3d.hpp) compiles without OpenCV/PCL with plain g++ and
drives the reference's call sequence (examples/cloud_processor_demo.cpp)."""
import os
import subprocess

import numpy as np
import pytest

from yolo_ppf_pose_estimation_amd import _capi, ply, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "yolo_ppf_pose_estimation_amd", "csrc")


def _build(tmp_path, name="cloud_processor_demo", std="c++17", with_opencv_stand_in=False):
    exe = str(tmp_path / name)
    inc = ["-I", os.path.join(ROOT, "include")] + (["-I", os.path.join(ROOT, "tests", "mock_opencv")] if with_opencv_stand_in else [])
    subprocess.run(["g++", f"-std={std}", "-Wall", "-Wextra", "-Werror"] + inc +
                   [os.path.join(ROOT, "examples", name + ".cpp"), "-L", CSRC, "-lppf_hip",
                    f"-Wl,-rpath,{CSRC}", "-o", exe], check=True)
    return exe


def _build_call_shapes(tmp_path, recipe, compiler="g++"):
    """tests/cpp/reference_call_shapes.cpp: the reference's call shapes with file-scope using directives and unqualified
    names only, as C++11 (the reference's standard, CMakeLists.txt:60) with warnings as errors."""
    exe = str(tmp_path / f"reference_call_shapes_{recipe}")
    subprocess.run([compiler, "-std=c++11", "-Wall", "-Wextra", "-Werror", f"-DSHAPES_RECIPE={recipe}",
                    "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "mock_opencv"),
                    os.path.join(ROOT, "tests", "cpp", "reference_call_shapes.cpp"), "-L", CSRC, "-lppf_hip",
                    f"-Wl,-rpath,{CSRC}", "-o", exe], check=True)
    return exe


def _inputs(tmp_path, bottle):
    m, s = str(tmp_path / "model.ply"), str(tmp_path / "scene.ply")
    ply.write_ply(bottle[::4], m)
    scene, _ = synth.make_scene(bottle, 3000, seed=3)
    ply.write_ply(scene, s)
    return m, s


def test_facade_compiles_and_fails_loudly_without_gpu(tmp_path, bottle):
    exe = _build(tmp_path)
    if _capi.lib().ppf_device_count() > 0:
        pytest.skip("a GPU is present")
    m, s = _inputs(tmp_path, bottle)
    r = subprocess.run([exe, m, s], capture_output=True, text=True)
    assert r.returncode == 10 + _capi.PPF_ERR_HIP
    assert "no HIP device" in r.stderr


@pytest.mark.gpu
def test_facade_matches_python_binding(tmp_path, bottle):
    from yolo_ppf_pose_estimation_amd.detector import ICP, PPF3DDetector
    exe = _build(tmp_path)
    m, s = _inputs(tmp_path, bottle)
    out = str(tmp_path / "moved.ply")
    r = subprocess.run([exe, m, s, "-", out, "0.05"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    votes = int(r.stdout.split("RESULT votes=")[1].split()[0])
    model = ply.load_ply_simple(m)
    scene = ply.load_ply_simple(s)
    poses = PPF3DDetector(0.05, 0.05).trainModel(model).match(scene, 0.05, 0.05)[:5]
    ICP(100, 0.005, 2.5, 8).registerModelToScene(model, scene, poses)  # the demo refines its top 5 the same way
    assert votes == poses[0].numVotes
    assert float(r.stdout.split("residual=")[1].split()[0]) == poses[0].residual
    moved = ply.load_ply_simple(out)
    want = ply.transform_pc_pose(model, poses[0].pose)
    np.testing.assert_allclose(moved[:, :3], want[:, :3], atol=2e-5)
    # S2B through the facade: edge == scene gives the same top pose
    r2 = subprocess.run([exe, m, s, s, out, "0.05"], capture_output=True, text=True)
    assert r2.returncode == 0 and int(r2.stdout.split("RESULT votes=")[1].split()[0]) == votes


@pytest.mark.parametrize("recipe", [1, 2])
def test_reference_call_shapes_compile_unqualified_under_both_recipes(tmp_path, bottle, recipe):
    """`using namespace cv;` next to the facade's namespace, unqualified Mat / Pose3D / PPF3DDetector / ICP / FileStorage:
    recipe 1 = include swap only (PPF_MATCH_3D_AS_CV), recipe 2 = include swap + two using directives.
    Without a GPU the program must stop at the first compute call with PPF_ERR_HIP."""
    exe = _build_call_shapes(tmp_path, recipe)
    if _capi.lib().ppf_device_count() > 0:
        pytest.skip("a GPU is present")
    m, s = _inputs(tmp_path, bottle)
    r = subprocess.run([exe, m, s, "-", str(tmp_path), str(tmp_path / "moved.ply")], capture_output=True, text=True)
    assert r.returncode == 10 + _capi.PPF_ERR_HIP, r.stdout + r.stderr
    assert "no HIP device" in r.stderr


def test_reference_call_shapes_compile_with_clang_too(tmp_path):
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("no clang++ in this image")
    for recipe in (1, 2):
        _build_call_shapes(tmp_path, recipe, compiler=clang)


def test_examples_compile_as_cxx11_with_and_without_the_opencv_stand_in(tmp_path):
    """the reference is C++11 (CMakeLists.txt:60); with OpenCV present the facade's Mat IS cv::Mat"""
    for name in ("cloud_processor_demo", "cloud_stages_demo", "pcl_pipeline_demo"):
        _build(tmp_path, name, std="c++11")
        _build(tmp_path, name, std="c++11", with_opencv_stand_in=True)


@pytest.mark.gpu
@pytest.mark.parametrize("recipe,edge", [(1, False), (2, True)])
def test_reference_call_shapes_run_and_equal_the_python_binding(tmp_path, bottle, recipe, edge):
    """the call-shape program (train -> FileStorage write -> read -> by-value copy -> match / match_S2B -> ICP top 5 ->
    transformPCPose -> writePLY) gives the votes, residual and moved cloud of the Python binding"""
    from yolo_ppf_pose_estimation_amd.detector import ICP, PPF3DDetector
    exe = _build_call_shapes(tmp_path, recipe)
    m, s = _inputs(tmp_path, bottle)
    out = str(tmp_path / "moved.ply")
    r = subprocess.run([exe, m, s, s if edge else "-", str(tmp_path), out, "0.05"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    res = dict(kv.split("=") for kv in r.stdout.split("RESULT ")[1].split())
    assert res["loaded_equals_trained"] == "1" and res["untrained_refused"] == "1" and float(res["depth"]) == float(np.float32(0.635))
    model, scene = ply.load_ply_simple(m), ply.load_ply_simple(s)
    det = PPF3DDetector(0.05, 0.05).trainModel(model)
    poses = (det.match_S2B(scene, scene, 0.05, 0.05) if edge else det.match(scene, 0.05, 0.05))[:5]
    ICP(100, 0.005, 2.5, 8).registerModelToScene(model, scene, poses)
    assert int(res["votes"]) == poses[0].numVotes
    assert float(res["residual"]) == poses[0].residual
    assert float(res["p03"]) == poses[0].pose[0, 3] and float(res["t0"]) == poses[0].t[0] and float(res["q0"]) == poses[0].q[0]
    assert int(res["moved_rows"]) == model.shape[0]
    moved = ply.load_ply_simple(out)
    np.testing.assert_allclose(moved[:, :3], ply.transform_pc_pose(model, poses[0].pose)[:, :3], atol=2e-5)
    assert os.path.getsize(tmp_path / "detector_bottle.xml") > 1000


def test_pcl_shaped_facade_compiles_and_fails_loudly_without_gpu(tmp_path, bottle):
    exe = _build(tmp_path, "pcl_pipeline_demo")
    if _capi.lib().ppf_device_count() > 0:
        pytest.skip("a GPU is present")
    m, s = _inputs(tmp_path, bottle)
    r = subprocess.run([exe, m, s], capture_output=True, text=True)
    assert r.returncode == 10 + _capi.PPF_ERR_HIP and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_pcl_shaped_facade_runs(tmp_path, bottle):
    """PPFEstimation -> PPFHashMapSearch -> PPFRegistration: the same engine call as the Python binding with the
    equivalent parameters (model rows as given, absolute distance step, sampling rate 20) and the PCL policy switches the
    PCL-named classes turn on (Darboux-frame pair feature, exact key equality, pair radius model_diameter / 2, relative rotation metric, 2 pi alpha range)."""
    from yolo_ppf_pose_estimation_amd.detector import PPF3DDetector
    exe = _build(tmp_path, "pcl_pipeline_demo")
    m, s = _inputs(tmp_path, bottle)
    r = subprocess.run([exe, m, s], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "converged=1" in r.stdout
    search = dict(kv.split("=") for kv in r.stdout.split("SEARCH ")[1].splitlines()[0].split())
    assert int(search["pairs"]) >= 1 and search["has01"] == "1" and search["sorted"] == "1"
    outp = dict(kv.split("=") for kv in r.stdout.split("OUTPUT ")[1].splitlines()[0].split())
    votes = int(r.stdout.split("votes=")[1].split()[0])
    model, scene = ply.load_ply_simple(m), ply.load_ply_simple(s)
    diameter = float(np.linalg.norm((model[:, :3].max(0) - model[:, :3].min(0)).astype(np.float32)))
    rel = float(np.float32(0.012)) / float(np.float32(diameter))
    det = PPF3DDetector(rel, rel, 2.0 * np.pi / float(np.float32(12.0 / 180.0 * 3.14159265)), distance_from_distance_step=True,
                        key_equality=1, feature=1)
    det.trainModel(model, presampled=True)
    det.setSearchParams(float(np.float32(0.05)), float(np.float32(30.0 / 180.0 * 3.14159265)))
    det.setPolicy(pair_radius=0.5 * float(np.float32(diameter)), rot_metric_relative=True, alpha_range_2pi=True)
    poses = det.match(scene, 1.0 / 20.0, 0.05, presampled=True)
    assert votes == poses[0].numVotes
    assert int(outp["rows"]) == model.shape[0] and outp["moved_ok"] == "1"   # align(output): the source under the final transformation
    # nearestNeighborSearch against the oracle's per-pair keys: every model pair with the key of pair (0, 1), nothing else
    import oracle_lib as O
    astep, dstep = det.info()["angle_step"], det.info()["distance_step"]
    f01, key01, _ = O.pair_feature_darboux(model[0, :3], model[0, 3:], model[1, :3], model[1, 3:], astep, dstep)
    got = det.nearest_pairs(f01.astype(np.float32))
    assert int(search["pairs"]) == len(got)
    rng = np.random.default_rng(0)
    gs = {(int(i), int(j)) for i, j in got}
    assert (0, 1) in gs
    for i, j in list(gs)[:40]:                               # what it returns has the key
        r_ = O.pair_feature_darboux(model[i, :3], model[i, 3:], model[j, :3], model[j, 3:], astep, dstep)
        assert r_ is not None and list(r_[1]) == list(key01)
    for _ in range(300):                                     # a sample of what it does not return has another key
        i, j = (int(v) for v in rng.integers(0, model.shape[0], size=2))
        if i == j or (i, j) in gs:
            continue
        r_ = O.pair_feature_darboux(model[i, :3], model[i, 3:], model[j, :3], model[j, 3:], astep, dstep)
        assert r_ is None or list(r_[1]) != list(key01)


def test_stage_wrapper_compiles_and_fails_loudly_without_gpu(tmp_path, bottle):
    exe = _build(tmp_path, "cloud_stages_demo")
    if _capi.lib().ppf_device_count() > 0:
        pytest.skip("a GPU is present")
    m = str(tmp_path / "model.ply")
    ply.write_ply(bottle[::40], m)
    s = str(tmp_path / "scene.ply")
    ply.write_ply(bottle[::40, :3], s)
    d = str(tmp_path / "depth.f32")
    np.zeros((8, 8), np.float32).tofile(d)
    r = subprocess.run([exe, m, s, d, "8", "8", "1", "1", "2", "2", "10", "10", "4", "4", "0.01", "1.0"],
                       capture_output=True, text=True)
    assert r.returncode == 10 + _capi.PPF_ERR_HIP
    assert "no HIP device" in r.stderr


@pytest.mark.gpu
def test_stage_wrapper_runs_the_drivers_sequence_on_the_real_frame(tmp_path, bottle):
    """examples/cloud_stages_demo.cpp (C++ wrapper of the stages + detector + ICP) on the reference's depth frame
    gives the counts and the refined residual of the Python mirror of the same sequence"""
    import prep_data as D
    from yolo_ppf_pose_estimation_amd.cloud_processor import DeviceCloud
    from yolo_ppf_pose_estimation_amd.detector import ICP, PPF3DDetector
    exe = _build(tmp_path, "cloud_stages_demo")
    xyz, depth, box, intr = D.c1_frame()
    xyz = xyz[::3].copy()                      # keep the ASCII PLY small
    m, s, d = str(tmp_path / "model.ply"), str(tmp_path / "scene.ply"), str(tmp_path / "depth.f32")
    ply.write_ply(bottle, m)
    ply.write_ply(xyz, s)
    depth.tofile(d)
    args = [exe, m, s, d, str(depth.shape[0]), str(depth.shape[1])] + [str(v) for v in box] + [repr(v) for v in intr] + ["0.004", "1.0"]
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    res = dict(kv.split("=") for kv in r.stdout.split("RESULT ")[1].split())
    obj = DeviceCloud.upload(xyz).crop(box, depth, intr).voxel_grid(0.004).outlier_removal(50, 1.0).normals(30)
    obj_mat, edge_mat = obj.to_mat().rows(), obj.edges(0.03).to_mat().rows()
    assert (int(res["object"]), int(res["edges"])) == (obj_mat.shape[0], edge_mat.shape[0])
    poses = PPF3DDetector(0.05, 0.05).trainModel(ply.load_ply_simple(m)).match_S2B(obj_mat, edge_mat, 0.05, 0.05)[:5]
    ICP(100, 0.005, 2.5, 8).registerModelToScene(ply.load_ply_simple(m), obj_mat, poses)
    assert int(res["votes"]) == poses[0].numVotes
    assert float(res["residual"]) == poses[0].residual


OPENCV_OVERLOADS = r'''
// The reference's own calls on cv::FileStorage / cv::FileNode (CloudProcessing.h:111-113, 249-251) and a cv::Mat view
// whose row step is larger than its column count, through the OpenCV overloads of the facade.
#include <cstdio>
#include <type_traits>
#include "ppf_match_3d.hpp"
using namespace ppfhip::ppf_match_3d;
int main(int argc, char** argv) {
  if (argc < 4) return 2;
  try {
    Mat model = loadPLYSimple(argv[1], 1), scene = loadPLYSimple(argv[2], 1); /* cv::Mat: the facade adds no second Mat */
    static_assert(std::is_same<Mat, cv::Mat>::value && std::is_same<Matx44d, cv::Matx44d>::value, "facade types are the cv types");
    // a 9-column CV_32F buffer whose first 6 columns are the cloud: step1() == 9, cols == 6
    cv::Mat wide(scene.rows, 9, CV_32F);
    for (int i = 0; i < scene.rows; i++) {
      for (int k = 0; k < 6; k++) wide.ptr<float>(i)[k] = scene.ptr<float>(i)[k];
      for (int k = 6; k < 9; k++) wide.ptr<float>(i)[k] = 1e9f;
    }
    cv::Mat view = wide.colRange(0, 6);
    cv::Mat cvmodel(model.rows, 6, CV_32F, model.ptr<float>(0));
    PPF3DDetector detector(0.05, 0.05);
    detector.trainModel(cvmodel);
    std::vector<Pose3DPtr> a, b, c;
    detector.match(scene, a, 0.05, 0.05);
    detector.match(view, b, 0.05, 0.05);
    {
      cv::FileStorage fsOut(argv[3], cv::FileStorage::WRITE);
      detector.write(fsOut);
      fsOut.release();
    }
    PPF3DDetector loaded(0.05, 0.05);
    cv::FileStorage fsLoad(argv[3], cv::FileStorage::READ);
    cv::FileNode fnLoad = fsLoad.root();
    loaded.read(fnLoad);
    loaded.match(view, c, 0.05, 0.05);
    bool same = a.size() == b.size() && a.size() == c.size() && !a.empty();
    for (size_t i = 0; same && i < a.size(); i++)
      same = a[i]->numVotes == b[i]->numVotes && a[i]->numVotes == c[i]->numVotes && a[i]->pose == b[i]->pose && a[i]->pose == c[i]->pose;
    bool refused = false;
    try { cv::Mat dbl(10, 6, CV_64F); detector.match(dbl, b, 0.05, 0.05); } catch (const Error& e) { refused = e.status == PPF_ERR_INVALID; }
    std::printf("RESULT poses=%zu same=%d refused_double=%d\n", a.size(), (int)same, (int)refused);
    return same && refused ? 0 : 1;
  } catch (const Error& e) {
    std::fprintf(stderr, "ppf error %d: %s\n", (int)e.status, e.what());
    return 10 + (int)e.status;
  }
}
'''


def _build_opencv_overloads(tmp_path):
    src = tmp_path / "opencv_overloads.cpp"
    src.write_text(OPENCV_OVERLOADS)
    exe = str(tmp_path / "opencv_overloads")
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-I",
                    os.path.join(ROOT, "tests", "mock_opencv"), str(src), "-L", CSRC, "-lppf_hip", f"-Wl,-rpath,{CSRC}",
                    "-o", exe], check=True)
    return exe


def test_opencv_overloads_compile_against_a_stand_in_header(tmp_path):
    """cv::FileStorage / cv::FileNode read/write and the step1()-based stride compile (a minimal stand-in for
    <opencv2/core.hpp> lives under tests/mock_opencv; OpenCV itself is not installed here)."""
    _build_opencv_overloads(tmp_path)


@pytest.mark.gpu
def test_opencv_overloads_round_trip_and_strided_views(tmp_path, bottle):
    exe = _build_opencv_overloads(tmp_path)
    m, s = _inputs(tmp_path, bottle)
    r = subprocess.run([exe, m, s, str(tmp_path / "detector_bottle.xml")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "same=1 refused_double=1" in r.stdout
