#!/usr/bin/env python3
"""Clock stamps inside one workgroup (the middle tile of image 0) of the LAST k_resize_chain launch of a call (library built with
`make -C hyslam_amd/csrc EXTRA=-DHS_PYR_PROFILE`): per stage, the time until the source is in LDS (stage 0: the global loads), the horizontal
pass, the vertical pass.  usage: pyramid_phase_profile.py [frames=2]"""
import ctypes as C
import sys
sys.path.insert(0, ".")
import hyslam_amd as HS
from hyslam_amd.synth import synth_stereo_pair
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2
L, R = synth_stereo_pair(1, 1920, 1080)
ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=2000))
imgs = [L, R] * (frames // 2)
ex.extract_batch(imgs); ex.extract_batch(imgs)
out = (C.c_ulonglong * 64)()
ex._lib.hs_debug_pyr_profile(out)
k = int(out[63]); t0 = out[0]
names = ["source in LDS", "horizontal pass", "vertical pass"]
print("stamps %d; total %d cycles" % (k, out[k - 1] - t0))
for i in range(1, k):
    st, ph = (i - 1) // 3, (i - 1) % 3
    print("stage %d %-16s +%6d  (total %7d)" % (st, names[ph], out[i] - out[i - 1], out[i] - t0))

import numpy as np
wg = (C.c_ulonglong * 8192)()
ex._lib.hs_debug_pyr_workgroups(wg)
a = np.array(list(wg), dtype=np.float64).reshape(4096, 2)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
print("workgroups %d: start median %.2f us, 90 %% %.2f, max %.2f; life median %.2f us, 90 %% %.2f, max %.2f; end median %.2f us, 90 %% %.2f, max %.2f (= the launch's span from the first workgroup's start)"
      % (len(a), np.median(st), np.percentile(st, 90), st.max(), np.median(en - st), np.percentile(en - st, 90), (en - st).max(), np.median(en), np.percentile(en, 90), en.max()))
late = np.argsort(-en)[:8]
print("latest finishers (index, start, life):", [(int(i), round(float(st[i]), 1), round(float(en[i] - st[i]), 1)) for i in late])
