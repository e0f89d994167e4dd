"""The header-only C++ adaptors (hyslam_amd/host/: FeatureExtractor / Stereomatcher / FeatureMatcher / FeatureFactory call surface over the
C ABI) compiled against host/cv_compat.h and run like hySLAM's call sites use them (ImageProcessing::ProcessStereoImage, TrackLocalMap,
TrackMotionModel, TrackReferenceKeyFrame, LandMarkFuser)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "cpp", "_build")
EXE = os.path.join(BUILD, "test_adaptor")
EXE_M = os.path.join(BUILD, "test_matcher_adaptor")
EXE_B = os.path.join(BUILD, "bench_adaptor")
EXE_R = os.path.join(BUILD, "test_matcher_replacement")


def build_replacement():
    """integration (b) of INTEGRATION.md §3: tests/cpp/test_matcher_adaptor.cpp compiled against the UNPATCHED declarations of the reference's FeatureMatcher /
    FeatureFactory (cv_compat.h with HYSLAM_AMD_COMPAT_UNPATCHED: no virtual anywhere) together with hyslam_amd/host/replace/FeatureMatcher.cc, the
    translation unit that takes the place of src/features/FeatureMatcher.cc"""
    os.makedirs(BUILD, exist_ok=True)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-pthread", "-DHYSLAM_AMD_COMPAT_UNPATCHED", "-DHYSLAM_AMD_UNPATCHED_MATCHER",
                           os.path.join(ROOT, "tests", "cpp", "test_matcher_adaptor.cpp"), os.path.join(ROOT, "hyslam_amd", "host", "replace", "FeatureMatcher.cc"), "-o", EXE_R,
                           "-L" + os.path.join(ROOT, "hyslam_amd"), "-lhyslam_amd", "-Wl,-rpath," + os.path.join(ROOT, "hyslam_amd"),
                           "-L" + os.path.join(ROOT, "oracle", "_build"), "-lhs_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle", "_build")])


def build(src="test_adaptor.cpp", exe=EXE):
    """the test programs check their results against the oracle and link it; bench_adaptor only times the adaptors and must NOT (bench.py runs it
    for its `call_site` block: nothing the bench measures may touch oracle/)"""
    os.makedirs(BUILD, exist_ok=True)
    with_oracle = "bench" not in src
    if with_oracle:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["g++", "-O2" if "bench" in src else "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-pthread", os.path.join(ROOT, "tests", "cpp", src), "-o", exe,
                           "-L" + os.path.join(ROOT, "hyslam_amd"), "-lhyslam_amd", "-Wl,-rpath," + os.path.join(ROOT, "hyslam_amd")]
                          + (["-L" + os.path.join(ROOT, "oracle", "_build"), "-lhs_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle", "_build")] if with_oracle else []))


def run():
    from hyslam_amd.synth import synth_stereo_pair
    L, R = synth_stereo_pair(7, 640, 480)
    return subprocess.run([EXE, "640", "480"], input=L.tobytes() + R.tobytes(), capture_output=True, timeout=300)


def write_scene(path):
    """a tracking scene (tests/scenes.py) in the flat layout tests/cpp/test_matcher_adaptor.cpp reads"""
    import scenes
    sc = scenes.projection_scene(71, 640, 480, nfeat=1000, copies=3)
    fa, lms = sc["frame_args"], sc["lms"]
    with open(path, "wb") as f:
        np.array([len(fa["kps"]), len(lms), fa["sensor"], 640, 480, 0, 0, 0], np.int32).tofile(f)
        np.concatenate([np.asarray(fa["Rcw"], np.float32).reshape(-1), np.asarray(fa["tcw"], np.float32),
                        np.array([fa["fx"], fa["fy"], fa["cx"], fa["cy"], fa["mbf"]], np.float32)]).tofile(f)
        np.ascontiguousarray(fa["kps"]).tofile(f)
        np.ascontiguousarray(fa["desc"], np.uint8).tofile(f)
        np.ascontiguousarray(fa["uR"], np.float32).tofile(f)
        np.ascontiguousarray(fa["kp_lm_obs"], np.int32).tofile(f)
        np.ascontiguousarray(lms).tofile(f)


def run_matcher(exe=EXE_M):
    scene = os.path.join(BUILD, "scene.bin")
    write_scene(scene)
    return subprocess.run([exe, scene], capture_output=True, timeout=600)


def run_bench(w=640, h=480, reps=6, n_lm=5000):
    """tests/cpp/bench_adaptor.cpp: ImageProcessing::ProcessStereoImage + TrackLocalMap + LandMarkTriangulator through the adaptors, timed"""
    from hyslam_amd.synth import synth_stereo_pair
    L, R = synth_stereo_pair(2, w, h)
    fl, fr = os.path.join(BUILD, "bench_L.raw"), os.path.join(BUILD, "bench_R.raw")
    L.tofile(fl); R.tofile(fr)
    return subprocess.run([EXE_B, str(w), str(h), fl, fr, str(reps), str(n_lm)], capture_output=True, timeout=900)


def test_planned_association_replay_equals_the_reference_loop():
    """hyslam_amd/host/HipAssociationReplay.h: the subsequence of Frame::associateLandMark calls the matcher adaptor makes after a projection search
    leaves views_to_landmarks, outliers AND n_matches exactly as the reference's loop over every match does (FeatureMatcher.cc:113-118,
    LandMarkMatches.cpp:26-51) — 3 000 randomised frames (stale outliers entries, landmarks on several views, landmarks that move, fresh frames) +
    BASELINE config 4's shape (14 000 matches onto 2 000 views -> one call per view).  Host only, under ASan + UBSan."""
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "test_replay")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(ROOT, "tests", "cpp", "test_replay.cpp"), "-o", exe])
    for args in (["3000"], ["1500", "991"]):
        r = subprocess.run([exe] + args, capture_output=True, timeout=600)
        assert r.returncode == 0 and b"REPLAY OK" in r.stdout, r.stdout + r.stderr
        assert b"config-4 shape" in r.stdout


def test_adaptor_compiles_and_fails_loudly_without_gpu():
    build()
    r = run()
    assert r.returncode == 0, r.stdout + r.stderr
    assert b"NO DEVICE" in r.stdout or b"ADAPTOR OK" in r.stdout


def test_matcher_adaptor_and_factory_compile_and_fail_loudly_without_gpu():
    build("test_matcher_adaptor.cpp", EXE_M)
    r = run_matcher()
    assert r.returncode == 0, r.stdout + r.stderr
    assert b"NO DEVICE" in r.stdout or b"MATCHER ADAPTOR OK" in r.stdout


def test_matcher_replacement_unit_compiles_against_the_unpatched_declarations():
    build_replacement()
    r = run_matcher(EXE_R)
    assert r.returncode == 0, r.stdout + r.stderr
    assert b"NO DEVICE" in r.stdout or b"MATCHER ADAPTOR OK" in r.stdout


def test_bench_adaptor_compiles():
    build("bench_adaptor.cpp", EXE_B)
    r = run_bench()
    assert r.returncode == 0, r.stdout + r.stderr
    assert b"NO DEVICE" in r.stdout or b"ProcessStereoImage_ms" in r.stdout


@pytest.mark.gpu
def test_bench_adaptor_runs_on_gpu(gpu):
    """the call-site timing binary runs end to end: stereo front end with two extractor threads + HipStereomatcher,
    a local-map projection search, a triangulation search — all with matches"""
    import json
    build("bench_adaptor.cpp", EXE_B)
    r = run_bench()
    assert r.returncode == 0, r.stdout + r.stderr
    j = json.loads(r.stdout.decode())
    assert j["keypoints"] > 500 and j["stereo_matches"] > 100
    assert j["TrackLocalMap_SearchByProjection_ms"]["matches"] > 200 and j["SearchForTriangulation_ms"]["matches"] > 100
    assert j["ProcessStereoImage_ms"]["total"] > 0
    # the extractor adaptors publish their frames; the stereo matcher and the projection search must have run on the device copies
    assert j["ProcessStereoImage_ms"]["stereo_frames_on_device"] == 1 and j["TrackLocalMap_SearchByProjection_ms"]["frame_on_device"] == 1
    t = j["TrackLocalMap_SearchByProjection_ms"]
    assert t["associateLandMark_calls"] <= j["keypoints"] < t["of_full_replay"] or t["of_full_replay"] <= j["keypoints"]


@pytest.mark.gpu
def test_adaptor_bit_exact_on_gpu(gpu):
    build()
    r = run()
    assert r.returncode == 0 and b"ADAPTOR OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_matcher_adaptor_replays_associations_in_address_order_on_gpu(gpu):
    """HipORBFactory -> unique_ptr<FeatureMatcher> -> HipFeatureMatcher: gather -> C ABI -> associateLandMark replay (D6: address-sorted
    landmarks) == oracle + reference-order replay; Fuse, SearchByBoW(KF, Frame), the reference-signature Stereomatcher, and the key-frame entry
    points SearchByProjection(pKF, Scw, ...), SearchBySim3, SearchForTriangulation (all views / stereo only), SearchByBoW2 and the empty Fuse(Scw)"""
    build("test_matcher_adaptor.cpp", EXE_M)
    r = run_matcher()
    assert r.returncode == 0 and b"MATCHER ADAPTOR OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_matcher_replacement_unit_on_gpu(gpu):
    """integration (b): the same ten searches through a PLAIN FeatureMatcher object of the reference's unpatched declarations (no virtual function, made by the
    base class's non-virtual FeatureFactory::getFeatureMatcher()), whose member functions hyslam_amd/host/replace/FeatureMatcher.cc defines over the C ABI"""
    build_replacement()
    r = run_matcher(EXE_R)
    assert r.returncode == 0 and b"MATCHER ADAPTOR OK" in r.stdout, r.stdout + r.stderr
