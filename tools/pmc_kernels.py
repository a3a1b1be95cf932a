"""PMC workload (GPU box, run under rocprofv3 --pmc ...): a few launches of ONE kernel at its workload shape.
    python tools/pmc_kernels.py attention [N pairs] | attention_x3 [N pairs] | attention_f16x2 [N pairs] | gemm ROWS K N | sinkhorn [N pairs] | lg_assign [N pairs] | conv | all
"all" = every kernel bench.py prints a `traffic` figure for, at the headline's launch shapes (16-pair chunks at the 5000-keypoint cap)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, bench
from gtsfm_amd.runtime import lib as L
lib = L.load(); dev = torch.device("cuda:0")
what = sys.argv[1] if len(sys.argv) > 1 else "all"
num = [int(a) for a in sys.argv[2:]]
if what in ("attention", "all"):
    bench.measure_attention_roofline(lib, dev, *(num or [5000, 16]), reps=4)
if what == "attention_x3":
    bench.measure_attention_roofline(lib, dev, *(num or [5000, 16]), reps=4, math=1)
if what == "attention_f16x2":
    bench.measure_attention_roofline(lib, dev, *(num or [5000, 16]), reps=4, math=2)
if what == "gemm":
    bench.measure_gemm_roofline(lib, dev, *num, reps=4)
if what == "all":
    for k, n in ((256, 768), (512, 512), (512, 256)):  # 256 -> 512 shares a grid with 512 -> 512: "gemm 163840 256 512" in a pass of its own
        bench.measure_gemm_roofline(lib, dev, 163840, k, n, reps=4)
    bench.measure_score_gemm_roofline(lib, dev, 5000, 16)
if what in ("lg_assign", "all"):
    bench.measure_lg_assignment_roofline(lib, dev, *(num or [5000, 16]), reps=4)
if what in ("sinkhorn", "all"):
    bench.measure_sinkhorn_roofline(lib, dev, *(num or [5000, 16]), iters=4)
if what in ("conv", "all"):
    bench.measure_conv_roofline(lib, dev, 8, 1024, 1024, reps=2)
