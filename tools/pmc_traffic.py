#!/usr/bin/env python3
"""HBM traffic of the extraction kernels from rocprofv3 PMC counters, calibrated on a kernel of known traffic.

Run on the GPU box (three separate counter passes; never combined with trace domains):
    rocprofv3 --pmc FETCH_SIZE  --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/pmc_traffic.py run
    rocprofv3 --pmc WRITE_SIZE  --output-format csv -d gpurun_out/pmc_write -- python3 tools/pmc_traffic.py run
    python3 tools/pmc_traffic.py report gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/...
`run` executes (a) two calibration copies of 1 GiB (4 B/lane and 16 B/lane: known 1 GiB read + 1 GiB written, far beyond the
256 MiB Infinity Cache) and (b) 3 steps of the bench workload (16 stereo pairs of 1920x1080).  `report` divides every kernel's
counter by the calibration factor of the matching access width (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide streaming
reads by 2x on gfx950; other widths must be calibrated in the kernel's own access pattern)."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GIB = 1 << 30
def _default_pairs():
    import re
    m = re.search(r'^DEFAULT_PAIRS = (\d+)', open(os.path.join(ROOT, 'bench.py')).read(), re.M)
    return int(os.environ.get('HS_PROFILE_PAIRS', m.group(1) if m else 16))


PAIRS = _default_pairs()      # bench.py's default batch: the counter passes profile the launch shapes the bench line reports
LANES = 1            # one launch sequence of 16 pairs per step: the launch shapes of bench.py's default (one handle)


def run():
    import numpy as np
    import torch
    import hyslam_amd as HS
    from hyslam_amd import _native as N
    from hyslam_amd.synth import synth_stereo_pair
    dev = torch.device("cuda", 0)
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=2000))
    ex.set_lanes(LANES)
    src = torch.randint(0, 255, (GIB,), dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    st = torch.cuda.current_stream().cuda_stream
    for width in (4, 16):
        N.check(ex._h, ex._lib.hs_debug_stream_copy(ex._h, dst.data_ptr(), src.data_ptr(), GIB, width, st))
    torch.cuda.synchronize()
    del src, dst
    W, H = 1920, 1080
    pairs = [synth_stereo_pair(1000 + i, W, H) for i in range(4)]
    left = torch.from_numpy(np.stack([pairs[i % 4][0] for i in range(PAIRS)])).to(dev)
    right = torch.from_numpy(np.stack([pairs[i % 4][1] for i in range(PAIRS)])).to(dev)
    cap = ex.max_keypoints()
    kb = N.KP_DTYPE.itemsize
    mk = lambda n, dt=torch.uint8: torch.zeros(n, dtype=dt, device=dev)
    kL, kR, dL, dR = mk(PAIRS * cap * kb), mk(PAIRS * cap * kb), mk(PAIRS * cap * 32), mk(PAIRS * cap * 32)
    nL, nR = mk(PAIRS, torch.int32), mk(PAIRS, torch.int32)
    uR, depth = mk(PAIRS * cap, torch.float32), mk(PAIRS * cap, torch.float32)
    sp = HS.stereo_params(HS.Camera(1050.0, 126.0, 1080.0))
    for _ in range(3):
        ex.stereo_frontend_batch_device(left.data_ptr(), right.data_ptr(), PAIRS, W, H, W, W * H, kL.data_ptr(), dL.data_ptr(), nL.data_ptr(),
                                        kR.data_ptr(), dR.data_ptr(), nR.data_ptr(), cap, sp, uR.data_ptr(), depth.data_ptr(), st)
    torch.cuda.synchronize()


def load(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out.setdefault((r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    return out


def report(dfetch, dwrite):
    F, Wr = load(dfetch), load(dwrite)
    cal = {}
    for name, width in (("k_copy_u32", 4), ("k_copy_u128", 16)):
        f = F[(name, "FETCH_SIZE")][0] * 1024 / GIB       # counter unit: KiB
        w = Wr[(name, "WRITE_SIZE")][0] * 1024 / GIB
        cal[width] = (f, w)
        print("# calibration %-11s (1 GiB read + 1 GiB written, %2d B/lane): FETCH_SIZE reports %.3f x, WRITE_SIZE reports %.3f x of the true bytes" % (name, width, f, w))
    print("kernel,launches,FETCH_SIZE_KiB_avg,WRITE_SIZE_KiB_avg,read_MB_per_launch_corrected,written_MB_per_launch_corrected,frames_per_launch")
    # the streaming kernels (pyramid, FAST, describe) read 16 B/lane; the rest reads and writes 4 B/lane
    kernels = sorted({k for (k, c) in F if k.startswith("k_") and not k.startswith("k_copy")})
    for k in kernels:
        wd = 16 if k.startswith(("k_resize_level_lds", "k_fast_rows", "k_describe")) else 4
        f = F[(k, "FETCH_SIZE")]; w = Wr[(k, "WRITE_SIZE")]
        fa, wa = sum(f) / len(f), sum(w) / len(w)
        print("%s,%d,%.1f,%.1f,%.2f,%.2f,%d" % (k.replace(",", ";"), len(f), fa, wa, fa * 1024 / cal[wd][0] / 1e6, wa * 1024 / cal[wd][1] / 1e6, 2 * PAIRS // LANES))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2], sys.argv[3])
