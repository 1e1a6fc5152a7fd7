#!/bin/bash
# usage (GPU box): tools/rt_dbg.sh  — the asynchronous drop-in mode paced at 22 kHz under a few engine switches
cd $GRAFT_REPO_ROOT
python - <<'P'
import sys, struct, numpy as np
sys.path.insert(0, '.')
from continuous_clustering_amd import synth, capi
st = synth.make_stream(2200 * 3, seed=1234, motion=synth.Motion.translate(10.0))
with open('/tmp/rt_in.bin', 'wb') as f:
    f.write(struct.pack('<iiii', 64, 2200, st.n_firings, 1))
    f.write(st.xyz.astype(np.float32).tobytes()); f.write(st.intensity.astype(np.uint8).tobytes()); f.write(st.poses.astype(np.float64).tobytes())
P
for rep in 1 2; do
export CC_ENABLE_ENV_OPTS=1
for cfg in "CC_X=0" "CC_OPT_SMALL_FRONT=0" "CC_OPT_SMALL_FRONT=0 CC_OPT_SEG_SMALL_MAX=0"; do
  echo "$cfg: $(env $cfg tests/cpp/dropin_demo /tmp/rt_in.bin /dev/null -1 22000 | grep feed)"
  echo "$cfg adaptive: $(env $cfg tests/cpp/dropin_demo /tmp/rt_in.bin /dev/null 0 22000 | grep feed)"
done; done
