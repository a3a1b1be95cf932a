"""PMC workload (GPU box, run under rocprofv3 --pmc ...): a few launches of ONE kernel at its workload shape.
    python tools/pmc_kernels.py attention | gemm K N | sinkhorn | conv"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, bench
from gtsfm_amd.runtime import lib as L
lib = L.load(); dev = torch.device("cuda:0")
what = sys.argv[1]
if what == "attention":
    bench.measure_attention_roofline(lib, dev, 2048, 32, reps=4)
elif what == "gemm":
    bench.measure_gemm_roofline(lib, dev, 131072, int(sys.argv[2]), int(sys.argv[3]), reps=4)
elif what == "sinkhorn":
    bench.measure_sinkhorn_roofline(lib, dev, 2048, 32, iters=4)
elif what == "conv":
    bench.measure_conv_roofline(lib, dev, 8, 1024, 1024, reps=2)
