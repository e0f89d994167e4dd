"""hs_comm_* — the config-5 exchange in C (an RCCL all-gather of the frame records, include/hyslam_amd.h) — through a C++ program with one
process per rank and through the Python binding.  CPU-only boxes start the ranks and stop cleanly before ncclCommInitRank."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "cpp", "_build")
EXE = os.path.join(BUILD, "test_comm")


def build():
    os.makedirs(BUILD, exist_ok=True)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "test_comm.cpp"), "-o", EXE, "-L" + os.path.join(ROOT, "hyslam_amd"), "-lhyslam_amd",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "hyslam_amd"), "-Wl,-rpath,/opt/rocm/lib"])


def run_world(world, tag):
    id_file = os.path.join(BUILD, "comm_id_%s_%d" % (tag, os.getpid()))
    if os.path.exists(id_file):
        os.remove(id_file)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([EXE, id_file, str(world), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append((p.communicate(timeout=300)[0].decode(), p.returncode))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
            outs.append(("timeout", 1))
    if os.path.exists(id_file):
        os.remove(id_file)
    return outs


def test_comm_two_processes_stop_cleanly_without_gpu_or_run():
    """world 2 as two processes: without a GPU both ranks report NO DEVICE before ncclCommInitRank; with one GPU they report NOT ENOUGH
    DEVICES (one device per rank); with two or more they gather each other's records"""
    build()
    outs = run_world(2, "w2")
    for text, rc in outs:
        assert rc == 0, outs
        assert "NO DEVICE" in text or "NOT ENOUGH DEVICES" in text or "COMM OK" in text, outs


def test_comm_binding_rejects_bad_arguments():
    from hyslam_amd import _native as N
    lib = N.lib()
    assert lib.hs_comm_get_unique_id(None) == N.HS_ERR_INVALID
    c = C.c_void_p()
    ident = (C.c_uint8 * 128)()
    assert lib.hs_comm_create(None, ident, 1, 0, C.byref(c)) == N.HS_ERR_INVALID and not c.value
    assert lib.hs_comm_world(None) == 0 and lib.hs_comm_rank(None) == -1
    lib.hs_comm_destroy(None)


@pytest.mark.gpu
def test_comm_world1_cpp_on_gpu(gpu):
    build()
    outs = run_world(1, "w1")
    assert outs[0][1] == 0 and "COMM OK rank 0 of 1" in outs[0][0], outs


@pytest.mark.gpu
def test_comm_world1_extract_gather_match_on_one_stream(gpu):
    """a config-5 step at world 1 through the C path only: extract straight into the record slot, hs_comm all-gather IN PLACE on the handle's
    stream, 2-NN on the gathered buffer — no event, no host synchronisation in between; the gathered record equals the oracle's extraction"""
    import hipmem
    import oracle
    import hyslam_amd as HS
    from hyslam_amd import distributed as D
    from hyslam_amd.synth import synth_image
    W, H, NF = 640, 480, 1000
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=NF))
    ex.reserve(W, H, 1)
    cap = ex.max_keypoints()
    rb = D.record_bytes(cap)
    o_n, o_k, o_d = D.record_offsets(cap)
    xc = D.RecordExchange(ex, D.RecordExchange.unique_id(), 1, 0)
    img = synth_image(200, W, H)
    d_f, recs = hipmem.DevBuf.from_numpy(img), hipmem.DevBuf(rb)
    outs = [hipmem.DevBuf(cap * 4) for _ in range(3)]
    ex.extract_batch_device(d_f.ptr, 1, W, H, W, W * H, recs.ptr + o_k, recs.ptr + o_d, recs.ptr + o_n, cap, 0)
    xc.allgather(recs.ptr, recs.ptr, rb, 0)
    D.records_knn2_device(ex, recs.ptr, rb, 1, 0, cap, outs[0].ptr, outs[1].ptr, outs[2].ptr, 0)
    ex.synchronize()
    k, d = D.unpack_record(recs.to_numpy(np.uint8, rb), cap)
    ok, od = oracle.extract(oracle.default_params(NF), img)
    assert k.tobytes() == ok.tobytes() and np.array_equal(d, od)
    xc.close()


def test_comm_probe_and_borrowers_symbols():
    """hs_comm_available is a non-collective probe (an answer, not a hang or a crash, with or without librccl / a GPU); a null handle borrows nothing"""
    from hyslam_amd import _native as N
    from hyslam_amd import distributed as D
    lib = N.lib()
    ok, why = D.comm_available()
    assert isinstance(ok, bool) and (ok or why)
    assert lib.hs_orb_borrowers(None) == 0


@pytest.mark.gpu
def test_comm_after_torch_uses_the_rccl_next_to_the_loaded_hip_runtime(gpu):
    """the order of a multi-GPU bench run: PyTorch first (its own libamdhip64 + librccl become the process's ROCm stack), then hs_comm.  The library
    must pick the RCCL that lives next to the HIP runtime in use — the system's librccl on PyTorch's runtime, or PyTorch's on the system's, failed in
    ncclCommInitRank ('unhandled cuda error').  In a fresh process: communicator at world 1, all-gather in place, clean exit."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, torch\n"
            "torch.cuda.set_device(0); _ = torch.zeros(4, device='cuda')\n"
            "import hipmem, hyslam_amd as HS\n"
            "from hyslam_amd import distributed as D\n"
            "ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=500))\n"
            "xc = D.RecordExchange(ex, D.RecordExchange.unique_id(), 1, 0)\n"
            "src, dst = hipmem.DevBuf.from_numpy(np.arange(4096, dtype=np.uint8)), hipmem.DevBuf(4096)\n"
            "xc.allgather(src.ptr, dst.ptr, 4096, 0); ex.synchronize()\n"
            "libs = sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'librccl' in l or 'libamdhip64' in l))\n"
            "print('GATHER_OK' if np.array_equal(dst.to_numpy(np.uint8, 4096), np.arange(4096, dtype=np.uint8)) else 'GATHER_BAD', libs)\n"
            "xc.close()\n") % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "GATHER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if "GATHER_OK" in l][0]
    assert line.count("librccl") == 1 and line.count("libamdhip64") == 1, "one ROCm stack in the process: " + line


def test_probe_before_torch_does_not_kill_the_process_at_exit():
    """librccl is loaded RTLD_LOCAL: a process that asks the C library for RCCL BEFORE it imports PyTorch ends up with two copies of RCCL (PyTorch ships
    its own and loads it by path); with RTLD_GLOBAL the first copy interposed on the second and the interpreter died in the static destructors at exit ('double free or
    corruption', exit code 134 — which is how a green test run turned into a failed one)"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from hyslam_amd import _native as N\n"
            "rc = N.lib().hs_comm_available()\n"
            "import torch\n"
            "print('probe', rc, 'torch', torch.__version__)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "probe" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_handle_destroyed_before_its_communicator_is_deferred(gpu):
    """a communicator borrows its handle: hs_orb_destroy on a borrowed handle only drops the owner's reference, the last hs_comm_destroy frees it — the order the header
    used to forbid (use-after-free in the first multi-rank teardown that got it wrong) is safe now, and the communicator still works in between"""
    import ctypes as C
    import hipmem
    from hyslam_amd import _native as N
    from hyslam_amd import distributed as D
    lib = N.lib()
    h = C.c_void_p()
    params = N.OrbParams()
    lib.hs_orb_default_params(C.byref(params))
    assert lib.hs_orb_create(C.byref(params), 0, C.byref(h)) == N.HS_OK
    assert lib.hs_orb_borrowers(h) == 0
    ident = (C.c_uint8 * 128)()
    assert lib.hs_comm_get_unique_id(ident) == N.HS_OK
    c1, c2 = C.c_void_p(), C.c_void_p()
    assert lib.hs_comm_create(h, ident, 1, 0, C.byref(c1)) == N.HS_OK and lib.hs_orb_borrowers(h) == 1
    assert lib.hs_comm_get_unique_id(ident) == N.HS_OK
    assert lib.hs_comm_create(h, ident, 1, 0, C.byref(c2)) == N.HS_OK and lib.hs_orb_borrowers(h) == 2
    lib.hs_orb_destroy(h)                                     # deferred: two communicators still use the handle's device and stream
    assert lib.hs_orb_borrowers(h) == 2
    rb = 4096
    src, dst = hipmem.DevBuf.from_numpy(np.arange(rb, dtype=np.uint8)), hipmem.DevBuf(rb)
    assert lib.hs_comm_allgather_records(c1, C.c_void_p(src.ptr), C.c_void_p(dst.ptr), rb, None) == N.HS_OK
    hipmem.sync()
    assert np.array_equal(dst.to_numpy(np.uint8, rb), np.arange(rb, dtype=np.uint8))
    lib.hs_comm_destroy(c1)
    assert lib.hs_orb_borrowers(h) == 1
    lib.hs_comm_destroy(c2)                                   # the last borrower: the handle is freed here (nothing to assert on a freed handle)
