cd $GRAFT_REPO_ROOT
echo "== shipped X3 GEMM"; GTSFM_GEMM_MATH=bf16x3 python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
from gtsfm_amd.runtime import lib as L
lib = L.load(); dev = torch.device("cuda:0")
for k, n in ((256, 768), (512, 512), (512, 256), (256, 512)):
    bench.measure_gemm_roofline(lib, dev, 163840, k, n, reps=10)
    v = [bench.measure_gemm_roofline(lib, dev, 163840, k, n, reps=10) for _ in range(3)]
    print(k, n, [r["avg_launch_ms"] for r in v], [r["algorithmic_frac_of_fp32_mfma_peak"] for r in v], flush=True)
PY
echo "== weights not split (ablation, garbage results)"; GTSFM_LIB=tools/bin/x3_nowsplit.so GTSFM_GEMM_MATH=bf16x3 python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
from gtsfm_amd.runtime import lib as L
lib = L.load(); dev = torch.device("cuda:0")
for k, n in ((256, 768), (512, 512), (512, 256), (256, 512)):
    bench.measure_gemm_roofline(lib, dev, 163840, k, n, reps=10)
    v = [bench.measure_gemm_roofline(lib, dev, 163840, k, n, reps=10) for _ in range(3)]
    print(k, n, [r["avg_launch_ms"] for r in v], [r["algorithmic_frac_of_fp32_mfma_peak"] for r in v], flush=True)
PY
