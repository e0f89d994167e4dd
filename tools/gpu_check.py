#!/usr/bin/env python3
"""Stage-by-stage GPU-vs-oracle report (dev tool; the asserting versions live in tests/test_gpu_*.py)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
import hyslam_amd as H
from hyslam_amd.synth import synth_image, synth_stereo_pair


def check_image(img, nfeat, tag):
    p = oracle.default_params(nfeat)
    ok, od, dbg = oracle.extract(p, img, debug=True)
    ex = H.ORBExtractor(H.FeatureExtractorSettings(nFeatures=nfeat)); ex.set_debug(True)
    t = time.time(); gk, gd = ex(img); dt = time.time() - t
    print(f"[{tag}] {img.shape} nfeat={nfeat}: oracle {len(ok)} kps, gpu {len(gk)} kps ({dt*1e3:.1f} ms incl. alloc)")
    allok = True
    for l in range(p.nlevels):
        gl = ex.debug_level(0, l)
        same = gl.shape == dbg['pyramid'][l].shape and np.array_equal(gl, dbg['pyramid'][l])
        gc = ex.debug_candidates(0, l)
        oc = dbg['candidates'][l].astype(np.int32)
        gcs = gc[np.lexsort((gc[:, 0], gc[:, 1]))] if len(gc) else gc
        ocs = oc[np.lexsort((oc[:, 0], oc[:, 1]))] if len(oc) else oc
        csame = gcs.shape == ocs.shape and np.array_equal(gcs, ocs)
        gs = ex.debug_selected(0, l)
        osel = np.stack([ok['x'][ok['octave'] == l], ok['y'][ok['octave'] == l]], 1)
        sc = oracle.scale_tables(p)[0][l]
        gsx = gs[:, :2].astype(np.float32) * (np.float32(sc) if l else np.float32(1))
        ssame = gsx.shape == osel.shape and np.array_equal(gsx, osel)
        print(f"   L{l}: pyramid {'OK' if same else 'DIFF'}  cand gpu={len(gc)} oracle={len(oc)} {'OK' if csame else 'DIFF'}  "
              f"selected gpu={len(gs)} oracle={len(osel)} {'OK' if ssame else 'DIFF'}")
        if not same and gl.shape == dbg['pyramid'][l].shape:
            d = np.argwhere(gl != dbg['pyramid'][l]); print("      first pyramid diffs", d[:5], len(d))
        if not csame:
            sg = set(map(tuple, gc.tolist())); so = set(map(tuple, oc.tolist()))
            print("      only gpu", list(sg - so)[:5], "only oracle", list(so - sg)[:5])
        if not ssame and len(gs) == len(osel):
            bad = np.argwhere((gsx != osel).any(1)).ravel(); print("      first selected diffs idx", bad[:5], gsx[bad[:3]], osel[bad[:3]])
        allok = allok and same and csame and ssame
    if len(gk) == len(ok):
        for f in ('x', 'y', 'size', 'angle', 'response', 'octave'):
            nb = int((gk[f] != ok[f]).sum())
            if nb: print(f"   field {f}: {nb} differ, e.g.", gk[f][gk[f] != ok[f]][:3], ok[f][gk[f] != ok[f]][:3]); allok = False
        nd = int((gd != od).any(1).sum())
        print(f"   descriptors differing rows: {nd} / {len(od)}")
        allok = allok and nd == 0
    else:
        allok = False
    print(f"[{tag}] {'BIT-EXACT' if allok else 'MISMATCH'}")
    return allok, ex, (gk, gd, ok, od)


if __name__ == "__main__":
    r1, *_ = check_image(synth_image(1, 640, 480), 1000, "C1")
    L, R = synth_stereo_pair(2, 1920, 1080)
    r2, ex, (gkL, gdL, okL, odL) = check_image(L, 2000, "C2-left")
    r3, ex2, (gkR, gdR, okR, odR) = check_image(R, 2000, "C2-right")
    sp = oracle.stereo_params()
    ouR, odepth, _, _ = oracle.stereo_match(okL, odL, okR, odR, sp)
    sm = H.Stereomatcher(okL, okR, odL, odR, H.Camera(), extractor=ex)
    sm.computeStereoMatches(); guR, gdepth = sm.getData()
    print("stereo: matches oracle", int((odepth > 0).sum()), "gpu", int((gdepth > 0).sum()),
          "uRight equal", np.array_equal(guR, ouR), "depth equal", np.array_equal(gdepth, odepth))
    if not np.array_equal(guR, ouR):
        bad = np.argwhere(guR != ouR).ravel(); print("  first diffs", bad[:5], guR[bad[:5]], ouR[bad[:5]])
    print("ALL", r1 and r2 and r3 and np.array_equal(guR, ouR) and np.array_equal(gdepth, odepth))
