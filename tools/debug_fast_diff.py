#!/usr/bin/env python3
"""Per-level diff of the FAST candidates (x, y, score) between the library and the oracle (GPU box):  python3 tools/debug_fast_diff.py [w h seed]"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
import hyslam_amd as HS
from hyslam_amd.synth import synth_image
w, h, seed = (int(a) for a in (sys.argv[1:4] + ["640", "480", "1"][len(sys.argv) - 1:]))
img = synth_image(seed, w, h)
p = oracle.default_params(1000, 1.2)
ok, od, dbg = oracle.extract(p, img, debug=True)
ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=1000, fScaleFactor=1.2, nLevels=8)); ex.set_debug(True)
ex(img)
for l in range(p.nlevels):
    gc = ex.debug_candidates(0, l); oc = dbg["candidates"][l].astype(np.int32)
    gs = {(int(a), int(b)): int(c) for a, b, c in gc}; os_ = {(int(a), int(b)): int(c) for a, b, c in oc}
    miss = sorted(set(os_) - set(gs)); extra = sorted(set(gs) - set(os_)); diff = [(k, gs[k], os_[k]) for k in sorted(set(gs) & set(os_)) if gs[k] != os_[k]]
    print("level %d: gpu %d oracle %d missing %s extra %s score diffs (xy, gpu, oracle) %s" % (l, len(gc), len(oc), miss[:8], extra[:8], diff[:8]))
