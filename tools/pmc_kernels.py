"""PMC workload (GPU box, run under rocprofv3 --pmc ...): a few launches of each MFMA kernel at its workload shape."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, bench
from gtsfm_amd.runtime import lib as L
lib = L.load(); dev = torch.device("cuda:0")
bench.measure_attention_roofline(lib, dev, 2048, 32, reps=6)
bench.measure_gemm_roofline(lib, dev, 131072, 256, 768, reps=6)
bench.measure_gemm_roofline(lib, dev, 131072, 512, 512, reps=6)
bench.measure_conv_roofline(lib, dev, 8, 1024, 1024, reps=2)
