"""Developer microbenchmarks (GPU box): GEMM shapes, attention, LayerNorm+GELU, timed with HIP events."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
from gtsfm_amd.runtime import lib as L
lib = L.load(); dev = torch.device("cuda:0")
# warm the clocks up: the first kernels after idle run ~15-20 % slower
for _ in range(3): bench.measure_attention_roofline(lib, dev, 2048, 32, reps=10)
for rows, k, n in [(131072, 256, 768), (131072, 256, 256), (131072, 512, 512), (131072, 512, 256), (131072, 256, 512), (4096, 256, 768), (16384, 256, 65)]:
    r = bench.measure_gemm_roofline(lib, dev, rows, k, n, reps=10)
    print(f"gemm {rows}x{k}->{n}: {r['avg_launch_ms']:.3f} ms {r['achieved']:.1f} TF ({r['frac']*100:.1f}%)")
for n, P in [(2048, 32), (1024, 32), (2048, 2), (512, 64)]:
    r = bench.measure_attention_roofline(lib, dev, n, P, reps=10)
    print(f"attention N={n} P={P}: {r['avg_launch_ms']:.3f} ms {r['achieved']:.1f} TF ({r['frac']*100:.1f}%)")
r = bench.measure_conv_roofline(lib, dev, 8, 1024, 1024, reps=5)
print(f"conv3x3 batch 8: {r['achieved']:.1f} TF ({r['frac']*100:.1f}%)")
