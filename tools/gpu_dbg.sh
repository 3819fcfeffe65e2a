#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 0 4; do
FLBGPU_LIB=$PWD/fluent-bit_amd/csrc/libflbgpu_gldbg.so python tools/debug_grep.py r $i 2>&1 | tail -1
done
python - <<'PY'
import numpy as np, collections
for i in (0, 4):
    a = np.load("gpurun_out/dbg_r%d_lane.npy" % i)
    c = collections.Counter(int(x) for x in a)
    print(i, [(hex(k - 1), v) for k, v in c.most_common(12)])
PY
