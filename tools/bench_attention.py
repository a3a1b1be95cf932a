"""Attention kernel on its own: fused vs split schedule at single-pair and batched launch shapes (TFLOP/s of 1024 N^2 per sequence).

    python tools/bench_attention.py [--quick] [--math 0|1|2]  # N in {2048, 5000}, 1 pair and the bench's chunk, exact fp32, bf16x3 and f16x2 arithmetic"""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from gtsfm_amd.runtime import lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev)


def run(n, npairs, mode, reps=20, math=0):
    nseq = 2 * npairs
    cap = -(-n // 128) * 128
    qkv = torch.randn((nseq * cap, 768), device=dev)
    out = torch.empty((nseq * cap, 256), device=dev)
    probs = torch.tensor([[s * cap, s, s * cap, s] for s in range(nseq)], dtype=torch.int32, device=dev)
    counts = torch.full((nseq,), n, dtype=torch.int32, device=dev)
    ws = torch.empty(int(lib.gtsfm_attention_math_workspace_bytes(nseq, n, n, 4, nseq * cap, math)), dtype=torch.uint8, device=dev)
    args = (qkv.data_ptr(), 768, qkv.data_ptr() + 1024, 768, qkv.data_ptr() + 2048, 768, out.data_ptr(), 256, probs.data_ptr(), counts.data_ptr(), nseq, n, n, 4,
            0.125, mode, math, nseq * cap, ws.data_ptr(), ws.numel(), stream.cuda_stream)
    L.check(lib.gtsfm_attention_math_f32(*args), "attention")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        L.check(lib.gtsfm_attention_math_f32(*args), "attention")
    e1.record(stream)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, 1024.0 * n * n * nseq / (ms * 1e-3) / 1e12


if __name__ == "__main__":
    quick = "--quick" in sys.argv
    only = int(sys.argv[sys.argv.index("--math") + 1]) if "--math" in sys.argv else None  # one arithmetic only (0 fp32, 1 bf16x3, 2 f16x2)
    for n, chunk in ((2048, 32), (5000, 16)) if quick else ((2048, 32), (5000, 8), (5000, 16), (1024, 32)):
        for npairs in ((1, chunk) if quick else (1, 2, chunk)):
            for math, label in ((0, "fp32  "), (1, "bf16x3"), (2, "f16x2 ")):
                if only is not None and math != only:
                    continue
                row = [f"N={n} pairs={npairs:2d} {label}"]
                for name, mode in (("fused", -1), ("split", 1), ("auto", 0)):
                    ms, tf = run(n, npairs, mode, math=math)
                    row.append(f"{name} {ms:7.3f} ms {tf:6.1f} TF/s (x{tf / 157.3:.3f} of the fp32 MFMA peak)")
                print("  ".join(row), flush=True)
