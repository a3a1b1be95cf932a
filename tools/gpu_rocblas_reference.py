"""Calibration (GPU box): what does the vendor fp32 GEMM (rocBLAS / hipBLASLt through torch.mm) reach on the matcher's
projection shapes? Same warm-up discipline as tools/gpu_gemm_ablation.py. Not used by the product."""
import torch

dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
for rows, k, n in ((131072, 256, 768), (131072, 512, 512), (131072, 512, 256), (131072, 256, 256), (8192, 8192, 8192)):
    a = torch.randn(rows, k, device=dev)
    w = torch.randn(k, n, device=dev)
    out = torch.empty(rows, n, device=dev)
    for _ in range(60):
        torch.mm(a, w, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 20
    for _ in range(reps):
        torch.mm(a, w, out=out)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tf = 2.0 * rows * k * n / (ms * 1e-3) / 1e12
    print(f"torch.mm fp32 {rows}x{k} @ {k}x{n}: {ms:.3f} ms  {tf:.1f} TFLOP/s  ({100 * tf / 157.3:.1f} % of the fp32 MFMA peak)", flush=True)
