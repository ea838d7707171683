set -o pipefail
mkdir -p gpurun_out/r4
export TRPL_AUTOBUILD=0
for lib in v_dbg v_dbg2; do TRPL_LIBRARY=$PWD/tools/ab/$lib.so timeout -k 10 120 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4/c35_dbg.txt
import sys, numpy as np
sys.path.insert(0, ".")
import trpl_amd
w = trpl_amd.workloads
X = np.load("gpurun_out_in/c31_X.npy")[[6598]]
ini, lens = w.twothick(128)
T = 13
info = {}
trpl_amd.loglik(X, ini, lens, T * 0.025, 128, T, [np.full(T + 1, 18.0)] * 6, info=info, MAX=1000, kernel="pair")
print(info["status"].T.tolist(), info["iters_total"].T.tolist())
PY
done
