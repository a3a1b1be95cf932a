"""Developer probe: the split-arithmetic attention launch alone, two independent problems sets on two streams at once, each compared with its own
result computed alone. Prints which (problem, head, query tile, rows, channels) differ.   python tools/x3_concurrency_probe.py MATH [N] [reps]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gtsfm_amd.runtime import lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
math = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
nseq = 4
cap = -(-n // 128) * 128


def make(seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    qkv = torch.randn((nseq * cap, 768), generator=g).to(dev)
    out = torch.zeros((nseq * cap, 256), device=dev)
    probs = torch.tensor([[s * cap, s, (s ^ 1) * cap, s ^ 1] for s in range(nseq)], dtype=torch.int32, device=dev)
    counts = torch.full((nseq,), n, dtype=torch.int32, device=dev)
    ws = torch.empty(int(lib.gtsfm_attention_math_workspace_bytes(nseq, n, n, 4, nseq * cap, math)) + 256, dtype=torch.uint8, device=dev)
    return qkv, out, probs, counts, ws


def launch(b, stream):
    qkv, out, probs, counts, ws = b
    L.check(lib.gtsfm_attention_math_f32(qkv.data_ptr(), 768, qkv.data_ptr() + 1024, 768, qkv.data_ptr() + 2048, 768, out.data_ptr(), 256, probs.data_ptr(), counts.data_ptr(),
                                         nseq, n, n, 4, 0.125, 0, math, nseq * cap, ws.data_ptr(), ws.numel(), stream.cuda_stream), "attention")


sets = [make(1), make(2)]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
other = sys.argv[4] if len(sys.argv) > 4 else "none"  # what a third stream runs meanwhile: none | gemm | ln
side = torch.cuda.Stream(dev)
ga = torch.randn((1024, 512), device=dev)
gw = torch.randn((768, 512), device=dev) * 0.05
gb = torch.zeros(768, device=dev)
gc = torch.empty((1024, 768), device=dev)


def side_work():
    if other == "gemm":
        for k, nn in ((256, 768), (512, 512), (512, 256), (256, 512)):
            L.check(lib.gtsfm_linear_rowmajor_f32(ga.data_ptr(), 512, 1024, None, k, gw.data_ptr(), 512, gb.data_ptr(), nn, None, gc.data_ptr(), 768, 0, None, 0, 1.0, 0, side.cuda_stream), "gemm")
refs = []
for b in sets:
    launch(b, torch.cuda.current_stream(dev))
    torch.cuda.synchronize()
    refs.append(b[1].clone())
bad = 0
for it in range(reps):
    for b in sets:
        b[1].zero_()
    torch.cuda.synchronize()
    for k in range(4):  # a few launches back to back per stream, as a layer loop does
        for b, s in zip(sets, streams):
            launch(b, s)
        side_work()
    torch.cuda.synchronize()
    for si, (b, r) in enumerate(zip(sets, refs)):
        if not torch.equal(b[1], r):
            bad += 1
            d = (b[1] != r).cpu().numpy()
            rows, cols = np.nonzero(d)
            err = float((b[1] - r).abs().max())
            print(f"rep {it} set {si}: {d.sum()} values differ, max |d| {err:.3e}; problems {sorted(set(rows // cap))} rows-in-problem {sorted(set(rows % cap))[:12]}.. ({len(set(rows % cap))} rows) "
                  f"heads {sorted(set(cols // 64))} channels {sorted(set(cols % 64))[:10]}.. ({len(set(cols % 64))})")
print(f"math {math} N {n}: launches with a difference: {bad} of {2 * reps}")
