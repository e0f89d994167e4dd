#!/usr/bin/env python3
"""Randomised extraction + stereo parity sweep: the HIP path against the oracle on random frame sizes, row strides, feature counts, scale
factors, level counts and image statistics (structured scenes, noise, checkerboards that saturate FAST, nearly flat frames).
Runs on the GPU box:  python3 tools/fuzz_parity.py --cases 300 --seed 1      (every case is printed; exit code 1 on the first mismatch)

--stress (round 4) shifts the draw to where capacity limits and batch-dependent paths live: frames up to 4000 x 3000, checkerboards / noise / clusters /
frames whose texture sits in one or two bands, quotas of 20-120 and 3 000-12 000, the cell edge (12-62 px), the FAST threshold (0-200) and the blur taps
drawn as well; in a share of the cases 2-4 frames per call, the stereo front end through the ingest tickets on 1-17 pairs, and the same handle reused
on three more frames of other sizes and content.  A mismatch is diagnosed on the spot (repeatability, first differing stage per level through the
debug taps, the differing records) and the frame is saved to gpurun_out/fuzz_mismatch.npz."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle                                   # noqa: E402  (tests/oracle.py: the checker)
import hyslam_amd as HS                         # noqa: E402
from hyslam_amd import _native as N             # noqa: E402
from hyslam_amd.synth import synth_image, synth_stereo_pair   # noqa: E402


def make_image(rng, kind, w, h, seed):
    if kind == "scene":
        return synth_image(seed, w, h)
    if kind == "noise":
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == "checker":                      # high-contrast blobs of random pitch: corner-saturated, exercises the list-overflow paths
        p = int(rng.integers(3, 12))
        yy, xx = np.mgrid[0:h, 0:w]
        img = (((yy // p) + (xx // p)) % 2 * int(rng.integers(120, 255))).astype(np.int32) + rng.integers(0, 30, (h, w))
        return np.clip(img, 0, 255).astype(np.uint8)
    if kind == "flat":                         # nearly empty: a handful of corners, many empty cells and levels
        img = np.full((h, w), int(rng.integers(20, 230)), np.int32) + rng.integers(0, 6, (h, w))
        for _ in range(int(rng.integers(0, 12))):
            x, y, s = int(rng.integers(0, w - 8)), int(rng.integers(0, h - 8)), int(rng.integers(3, 9))
            img[y:y + s, x:x + s] += int(rng.integers(60, 120))
        return np.clip(img, 0, 255).astype(np.uint8)
    if kind == "cluster":                      # a flat frame with a few dense patches: the quadtree needs nodes deeper than its histogram pyramid
        img = np.full((h, w), int(rng.integers(40, 200)), np.int32) + rng.integers(0, 4, (h, w))
        for _ in range(int(rng.integers(1, 6))):
            pw, ph = int(rng.integers(12, max(13, w // 4))), int(rng.integers(12, max(13, h // 4)))
            x, y = int(rng.integers(0, max(1, w - pw))), int(rng.integers(0, max(1, h - ph)))
            img[y:y + ph, x:x + pw] = rng.integers(0, 256, img[y:y + ph, x:x + pw].shape)
        return np.clip(img, 0, 255).astype(np.uint8)
    if kind == "band":                         # all the texture in one or two narrow horizontal bands: the keypoints of a frame crowd into a few of the stereo matcher's 32-row strips
        img = np.full((h, w), int(rng.integers(40, 200)), np.int32) + rng.integers(0, 4, (h, w))
        for _ in range(int(rng.integers(1, 3))):
            bh = int(rng.integers(40, max(41, min(160, h // 2))))
            y = int(rng.integers(0, max(1, h - bh)))
            img[y:y + bh] = synth_image(seed, w, bh).astype(np.int32) if rng.random() < 0.5 else rng.integers(0, 256, (bh, w))
        return np.clip(img, 0, 255).astype(np.uint8)
    # gradient + mid-frequency texture
    yy, xx = np.mgrid[0:h, 0:w]
    img = 128 + 60 * np.sin(xx / float(rng.integers(3, 40))) * np.cos(yy / float(rng.integers(3, 40))) + (xx * 40.0 / w) + rng.normal(0, float(rng.uniform(0, 12)), (h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def diagnose(ex, view, img, p, ok, od, gk, gd, big):
    """a mismatch: is it repeatable, which stage differs first (pyramid bytes / candidate sets / selection per level, through the debug taps), which records"""
    out = []
    try:
        reps = [ex(view) for _ in range(3)]
        out.append("repeat runs equal to the first: %s" % [bool(k.tobytes() == gk.tobytes() and np.array_equal(d, gd)) for k, d in reps])
        out.append("repeat runs equal to the oracle: %s" % [bool(len(k) == len(ok) and k.tobytes() == ok.tobytes() and np.array_equal(d, od)) for k, d in reps])
        _, _, dbg = oracle.extract(p, np.ascontiguousarray(img), cap=big, debug=True, cand_cap=1 << 21)
        ex.set_debug(True)
        gk2, gd2 = ex(view)
        out.append("debug-mode run equal to the oracle: %s" % bool(len(gk2) == len(ok) and gk2.tobytes() == ok.tobytes() and np.array_equal(gd2, od)))
        for l in range(p.nlevels):
            pyr_ok = np.array_equal(ex.debug_level(0, l), dbg["pyramid"][l])
            gc = ex.debug_candidates(0, l); oc = dbg["candidates"][l].astype(np.int32)
            cand_ok = len(gc) == len(oc) and (len(gc) == 0 or np.array_equal(gc[np.lexsort((gc[:, 0], gc[:, 1]))], oc[np.lexsort((oc[:, 0], oc[:, 1]))]))
            gs = ex.debug_selected(0, l)
            out.append("level %d: pyramid %s, candidates %s (gpu %d, oracle %d), selected gpu %d oracle %d" % (l, pyr_ok, cand_ok, len(gc), len(oc), len(gs), int(dbg["n_selected"][l])))
        ex.set_debug(False)
        n = min(len(gk), len(ok))
        bad = [i for i in range(n) if gk[i].tobytes() != ok[i].tobytes() or not np.array_equal(gd[i], od[i])]
        out.append("records that differ (product run): %d of %d, first %s" % (len(bad), n, bad[:8]))
        for i in bad[:4]:
            out.append("  [%d] gpu %s | oracle %s | descriptor equal %s" % (i, gk[i], ok[i], bool(np.array_equal(gd[i], od[i]))))
        os.makedirs("gpurun_out", exist_ok=True)
        np.savez_compressed("gpurun_out/fuzz_mismatch.npz", img=img, stride=np.int64(view.strides[0]), nfeat=p.nfeatures, scale=p.scale_factor, levels=p.nlevels, gk=gk, gd=gd, ok=ok, od=od)
    except Exception as e:                      # the diagnosis must never hide the mismatch itself
        out.append("diagnosis failed: %r" % (e,))
    return "\n    " + "\n    ".join(out)


STRESS = False                                  # --stress: big and saturated frames, tiny and huge quotas, banded frames, several frames per call


REFUSALS = {}          # reason -> [count, of which the reference itself has no defined behaviour]


def refusal_reason(where, err, w, h, nfeat, scale, levels, cell):
    """Classifies a refusal and checks it against an INDEPENDENT statement of the rules: returns (reason, expected).  A configuration the
    library refuses although no rule says so is a FAILURE of the campaign — the reference handles it, the drop-in does not.
    `reference-undefined`: the reference divides by zero / indexes an empty vector / asserts there; `library-limit`: the reference would run."""
    f32 = np.float32
    sc = [f32(1)]
    for _ in range(1, levels):
        sc.append(f32(np.float64(sc[-1]) * np.float64(f32(scale))))
    sizes = [(int(np.rint(f32(w) * (f32(1) / s))), int(np.rint(f32(h) * (f32(1) / s)))) for s in sc]
    factor = f32(np.float64(1.0) / np.float64(f32(scale)))
    nd = f32(nfeat) * (f32(1) - factor) / (f32(1) - f32(np.float64(factor) ** levels))
    quota, tot = [], 0
    for _ in range(levels - 1):
        q = int(np.rint(nd)); quota.append(q); tot += q; nd = nd * factor
    quota.append(max(nfeat - tot, 0))
    rules = []
    if max(quota) + 8 > 3328:
        rules.append(("library-limit: a level quota above 3320 (quadtree list capacity)", False))
    for (lw, lh) in sizes:
        if lw < 1 or lh < 1:
            rules.append(("reference-undefined: a pyramid level collapses to zero size (cv::resize asserts)", True)); break
    for (lw, lh) in sizes:
        qw, qh = lw - 32, lh - 32                 # minBorder = EDGE_THRESHOLD - 3 = 16 on both sides (ORBExtractor.cpp:413-416)
        ncols, nrows = (int(f32(qw) / f32(cell)) if qw > 0 else 0), (int(f32(qh) / f32(cell)) if qh > 0 else 0)
        if ncols < 1 or nrows < 1:
            continue
        wc, hc = int(np.ceil(f32(qw) / f32(ncols))), int(np.ceil(f32(qh) / f32(nrows)))
        if wc > 247 or hc > 125:
            rules.append(("library-limit: FAST cell wider than 247 px or taller than 125 px", False))
        n_ini = int(np.round(f32(qw) / f32(qh)))
        if n_ini < 1:
            rules.append(("reference-undefined: aspect ratio w/h < 0.5 (nIni == 0: division by zero, ORBExtractor.cpp:183)", True))
        elif n_ini > (3328 if max(quota) + 8 > 2048 else 2048) // 4:
            rules.append(("library-limit: more than %d root nodes" % ((3328 if max(quota) + 8 > 2048 else 2048) // 4), False))
    if where == "create":                       # hs_orb_create looks at the parameters and the quotas only (the geometry is checked at the first frame)
        rules = [r for r in rules if "quota" in r[0]]
        if levels < 1 or levels > 16 or not scale > 1.0 or nfeat < 1 or cell < 8:
            rules.insert(0, ("library-limit: parameters out of range (levels outside 1..16, scale <= 1, cell < 8 px)", False))
    if not rules:
        reason, undefined, expected = "UNEXPECTED (%s): %s" % (where, err), False, False
    else:
        reason, undefined, expected = rules[0][0], rules[0][1], True
    c = REFUSALS.setdefault(reason, [0, 0]); c[0] += 1; c[1] += int(undefined)
    return reason, expected


def one_case(rng, i):
    if STRESS:
        kind = ["checker", "checker", "checker", "noise", "cluster", "band", "band", "scene"][int(rng.integers(0, 8))]
        large = rng.random() < 0.45
        w = int(rng.integers(1500, 4001)) if large else int(rng.integers(64, 1500))
        h = int(rng.integers(1000, 3001)) if large else int(rng.integers(64, 1100))
    else:
        kind = ["scene", "scene", "noise", "checker", "flat", "texture", "cluster"][int(rng.integers(0, 7))]
        w = int(rng.integers(64, 1500)) if rng.random() < 0.85 else int(rng.integers(1500, 2600))
        h = int(rng.integers(64, 1100)) if rng.random() < 0.85 else int(rng.integers(1100, 1700))
    r_shape = rng.random()
    if r_shape < 0.80:
        h = min(h, w)                           # landscape / square: (w_l - 32) / (h_l - 32) >= 1 on every level, never refused for its aspect ratio
    elif r_shape < 0.95:
        h = min(h, 2 * w - 1)                   # portrait up to 1 : 2 — w/h < 0.5 on a level with cells gives zero root nodes in the reference (division by zero, :183): refused by the
                                                # library; the deeper levels of a portrait frame cross that line sooner than level 0 (round 5-6 campaigns: 20-25 % of ALL draws were such refusals)
    nfeat = int(rng.integers(20, 4000))
    if STRESS:
        r = rng.random()
        nfeat = int(rng.integers(20, 120)) if r < 0.35 else (int(rng.integers(3000, 12000)) if r < 0.6 else nfeat)
    scale = float(np.float32(rng.choice([1.1, 1.2, 1.2, 1.25, 1.3, 1.4, 1.5, 2.0])))
    levels = int(rng.choice([1, 2, 4, 6, 8, 8, 8, 10, 12]))
    if levels <= 2:
        nfeat = min(nfeat, int(rng.integers(20, 3300)))      # a level's quota is limited to 3320 (quadtree list in LDS; 2040 until round 6)
    seed = int(rng.integers(0, 1 << 30))
    cell, fth = 30, 20
    if STRESS and rng.random() < 0.4:           # parameters the reference never varies but the C ABI accepts: the cell edge (N_CELLS) and the FAST threshold
        cell = int(rng.choice([12, 16, 20, 25, 30, 37, 45, 62]))
        fth = int(rng.choice([0, 1, 3, 5, 7, 10, 15, 20, 25, 40, 63, 64, 100, 128, 200]))
    taps = None
    if STRESS and rng.random() < 0.15:          # blur taps other than OpenCV's {18,34,49,55,49,34,18}: byte-sized sets (the fast path, with and without saturation),
        tk = int(rng.integers(0, 5))            # sets that overflow a byte or whose sum needs the saturating generic path, the identity, zeros
        taps = [[16, 34, 50, 56, 50, 34, 16], [int(v) for v in rng.integers(0, 74, 7)], [int(v) for v in rng.integers(0, 1200, 7)],
                [0, 0, 0, 256, 0, 0, 0], [255, 255, 255, 255, 255, 255, 255]][tk]
    desc = "case %d: %s %dx%d nfeat %d scale %.2f levels %d seed %d" % (i, kind, w, h, nfeat, scale, levels, seed)
    if (cell, fth) != (30, 20):
        desc += " cell %d threshold %d" % (cell, fth)
    if taps is not None:
        desc += " taps %s" % taps
    try:
        ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=nfeat, fScaleFactor=scale, nLevels=levels, N_CELLS=cell), blur_taps=taps, fast_threshold=fth)
    except Exception as e:                      # configuration the library refuses: must be refused cleanly AND for a stated reason
        reason, expected = refusal_reason("create", str(e)[:60], w, h, nfeat, scale, levels, cell)
        return desc + "  -> refused at create [%s]: %s" % (reason, str(e)[:60]), expected
    img = make_image(rng, kind, w, h, seed)
    pad = int(rng.choice([0, 0, 1, 3, 16, 61]))
    if pad:                                     # a row stride larger than the width (odd strides take the unaligned load path)
        buf = np.zeros((h, w + pad), np.uint8)
        buf[:, :w] = img
        view = buf[:, :w]
    else:
        view = img
    try:
        gk, gd = ex(view)
    except Exception as e:
        reason, expected = refusal_reason("extract", str(e)[:80], w, h, nfeat, scale, levels, cell)
        return desc + "  -> refused at extract [%s]: %s" % (reason, str(e)[:80]), expected
    p = oracle.default_params(nfeat, scale, levels)
    p.cell_px, p.fast_threshold = cell, fth
    if taps is not None:
        for j, tv in enumerate(taps):
            p.blur_taps[j] = tv
    big = 4 * nfeat + 64 * levels + 1024         # small quotas overshoot: a breadth-first pass quadruples the list before the size is checked
    ok, od = oracle.extract(p, np.ascontiguousarray(img), cap=big)
    good = len(gk) == len(ok) and gk.tobytes() == ok.tobytes() and np.array_equal(gd, od)
    msg = desc + " pad %d -> %d keypoints %s" % (pad, len(ok), "ok" if good else "MISMATCH (gpu %d)" % len(gk))
    if not good:
        msg += diagnose(ex, view, img, p, ok, od, gk, gd, big)
    if good and STRESS and rng.random() < 0.3:             # several frames of the same size in ONE call (the batch paths: item lists, schedules, keys by batch size)
        nb = int(rng.integers(2, 5))
        frames = [img] + [make_image(rng, kind, w, h, seed + 1 + j) for j in range(nb - 1)]
        bks, bds = ex.extract_batch(frames)
        for j, (bk, bd) in enumerate(zip(bks, bds)):
            okj, odj = (ok, od) if j == 0 else oracle.extract(p, np.ascontiguousarray(frames[j]), cap=big)
            if not (len(bk) == len(okj) and bk.tobytes() == okj.tobytes() and np.array_equal(bd, odj)):
                good = False
                msg += "; batch of %d: frame %d MISMATCH" % (nb, j)
        if good:
            msg += "; batch of %d ok" % nb
    if good and STRESS and rng.random() < 0.3:
        # the SAME handle on a sequence of other frames: other sizes (reconfiguration, workspace reuse), saturated after nearly empty and back (whatever an
        # earlier call left in the candidate slots, counters, key histograms and lists must not leak into the next result)
        for q in range(3):
            k2 = ["flat", "checker", "noise", "scene", "cluster", "band"][int(rng.integers(0, 6))]
            if rng.random() < 0.5:
                w2, h2 = w, h
            else:
                w2, h2 = int(rng.integers(64, 1400)), int(rng.integers(64, 1000))
                h2 = min(h2, 2 * w2 - 1)
            img2 = make_image(rng, k2, w2, h2, seed + 1000 + q)
            try:
                g2k, g2d = ex(img2)
            except Exception as e:
                msg += "; reuse %d (%s %dx%d) refused: %s" % (q, k2, w2, h2, str(e)[:40])
                continue
            o2k, o2d = oracle.extract(p, img2, cap=big)
            if not (len(g2k) == len(o2k) and g2k.tobytes() == o2k.tobytes() and np.array_equal(g2d, o2d)):
                good = False
                msg += "; handle reuse %d (%s %dx%d after %s %dx%d) MISMATCH (gpu %d oracle %d)" % (q, k2, w2, h2, kind, w, h, len(g2k), len(o2k))
                break
        else:
            msg += "; handle reused on 3 frames ok"
    if good and STRESS and w * h <= 700 * 700 and rng.random() < 0.35:
        # the stereo FRONT END on several pairs in one call, through the ingest tickets (submit / wait: the strips binned inside the describe launch, the
        # batch-dependent choices of item width, schedule and keys): n pairs of distinct content, right = shifted + noisy left
        npairs = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 17]))
        lefts = [img] + [make_image(rng, kind, w, h, seed + 100 + j) for j in range(npairs - 1)]
        sh = int(rng.integers(1, 40))
        rights = [np.clip(np.roll(L, -sh, axis=1).astype(np.int32) + rng.integers(-3, 4, L.shape), 0, 255).astype(np.uint8) for L in lefts]
        fx = float(rng.uniform(300, 1500))
        osp = oracle.stereo_params(fx=fx, mbf=fx * 0.12, n_rows=h)
        gsp = N.StereoParams(fx, fx * 0.12, h, 100.0, 50.0, 31.0)
        t = ex.submit_batch([np.ascontiguousarray(a) for a in lefts + rights], gsp)
        n, fk, fd, fu, fz = ex.wait(t)
        fgood = True
        for j in range(npairs):
            okL, odL, okR, odR, ouR, oz = oracle.stereo_frontend(p, osp, lefts[j], rights[j], cap=big)
            a, b = int(n[j]), int(n[npairs + j])
            same = (a == len(okL) and b == len(okR) and fk[j, :a].tobytes() == okL.tobytes() and np.array_equal(fd[j, :a], odL)
                    and fk[npairs + j, :b].tobytes() == okR.tobytes() and np.array_equal(fd[npairs + j, :b], odR)
                    and np.array_equal(fu[j, :a], ouR) and np.array_equal(fz[j, :a], oz))
            if not same:
                fgood = False
                msg += "; front end, %d pairs: pair %d MISMATCH (n %d/%d oracle %d/%d)" % (npairs, j, a, b, len(okL), len(okR))
        good = good and fgood
        if fgood:
            msg += "; front end, %d pairs ok" % npairs
    if good and w * h <= 900 * 700 and rng.random() < (0.25 if STRESS else 0.12):
        # the camera-frame entry point (round 6: ImageProcessing::PreProcessImg on the device): a 1 / 3 / 4-channel frame of another size whose scaled grey version
        # has THIS case's size class, a random camera scale (the copy, the exact-0.5 area path, bilinear both ways) and colour order, odd row strides
        cn = int(rng.choice([1, 3, 3, 4]))
        cscale = float(np.float32(rng.choice([1.0, 0.5, 0.5, 0.75, 0.6, 0.3, 1.25, 2.0])))
        cw, ch = max(8, int(round(w / cscale))), max(8, int(round(h / cscale)))
        if cw * ch * cn <= 8_000_000:
            base = make_image(rng, kind, cw, ch, seed + 500)
            if cn == 1:
                cam = base
            else:
                chans = [base, np.roll(base, 3, axis=1), np.clip(base.astype(np.int32) + rng.integers(-20, 21, base.shape), 0, 255).astype(np.uint8)]
                if cn == 4:
                    chans.append(rng.integers(0, 256, base.shape, dtype=np.uint8))
                cam = np.ascontiguousarray(np.stack(chans, axis=2))
            crgb = bool(rng.integers(0, 2))
            try:
                og = oracle.preprocess(cam, crgb, cscale)
            except ValueError:
                og = None
            if og is not None:
                try:
                    (ck, cd, cg) = ex.extract_camera_batch([cam], crgb, cscale, want_grey=True)
                    okc, odc = oracle.extract(p, og, cap=big)
                    cgood = np.array_equal(cg[0], og) and len(ck[0]) == len(okc) and ck[0].tobytes() == okc.tobytes() and np.array_equal(cd[0], odc)
                    msg += "; camera %dx%dx%d scale %.2f %s -> %dx%d %s" % (cw, ch, cn, cscale, "RGB" if crgb else "BGR", og.shape[1], og.shape[0], "ok" if cgood else "MISMATCH")
                    good = good and cgood
                except Exception as e:
                    reason, expected = refusal_reason("extract", str(e)[:80], og.shape[1], og.shape[0], nfeat, scale, levels, cell)
                    msg += "; camera frame refused [%s]" % reason
                    good = good and expected
    if good and len(ok) > 20 and rng.random() < (0.5 if STRESS else 0.3):      # stereo: shifted copy with noise as the right frame
        sh = int(rng.integers(1, 40))
        right = np.roll(img, -sh, axis=1).copy()
        right = np.clip(right.astype(np.int32) + rng.integers(-3, 4, right.shape), 0, 255).astype(np.uint8)
        gkR, gdR = ex(right)
        okR, odR = oracle.extract(p, right, cap=big)
        good = gkR.tobytes() == okR.tobytes() and np.array_equal(gdR, odR)
        fx = float(rng.uniform(300, 1500))
        sp = oracle.stereo_params(fx=fx, mbf=fx * 0.12, n_rows=h)
        ouR, odepth, _, _ = oracle.stereo_match(ok, od, okR, odR, sp)
        sm = HS.Stereomatcher(gk, gkR, gd, gdR, HS.Camera(fx, fx * 0.12, float(h)), extractor=ex)
        sm.computeStereoMatches()
        guR, gdepth = sm.getData()
        sgood = np.array_equal(guR, ouR) and np.array_equal(gdepth, odepth)
        msg += "; stereo shift %d: %d matches %s" % (sh, int((odepth > 0).sum()), "ok" if (good and sgood) else "MISMATCH")
        good = good and sgood
    return msg, good


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=1e9, help="stop after this much wall time")
    ap.add_argument("--stress", action="store_true", help="big / saturated / banded frames, tiny and huge quotas, several frames per call")
    a = ap.parse_args()
    global STRESS
    STRESS = a.stress
    rng = np.random.default_rng(a.seed)
    t0, bad = time.time(), 0
    for i in range(a.cases):
        msg, good = one_case(rng, i)
        print(msg, flush=True)
        bad += not good
        if not good or time.time() - t0 > a.seconds:
            break
    for reason, (cnt, undef) in sorted(REFUSALS.items(), key=lambda kv: -kv[1][0]):
        print("refusals: %4d  %s" % (cnt, reason), flush=True)
    print("fuzz: %d cases, %d mismatches or unexpected refusals, %d refused (%d of them where the reference itself is undefined), %.0f s"
          % (i + 1, bad, sum(c for c, _ in REFUSALS.values()), sum(u for _, u in REFUSALS.values()), time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
