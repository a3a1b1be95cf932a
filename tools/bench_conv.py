"""conv3x3_mfma_kernel launch by launch over the SuperPoint stack (1024x1024, batch 8 by default), fraction of the fp32 MFMA peak.
Round 4 used it to measure a persistent form of the kernel (512 workgroups walking the tiles, the second half of the grid started half a
tile late so that the two workgroups of a CU are out of phase): 0.73 / 0.77 / 0.76 against 0.80 / 0.85 / 0.82 on the 64 -> 64 and 64 -> 128
layers whatever the phase offset, 1024 workgroups 0.755 / 0.79 / 0.78 -- the hardware dispatcher balances better than a static walk; not kept.
    python tools/bench_conv.py [H W batch]"""
import os, sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import bench
from gtsfm_amd.runtime import lib as L
lib = L.load(); dev = torch.device("cuda:0"); stream = torch.cuda.current_stream(dev)
h, w, batch = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 1024, 8)

def layers():
    out = []
    for li, (cin, cout, hh, ww, pool) in enumerate(bench.superpoint_conv3x3_layers(h, w)):
        if li == 0:
            img = torch.randint(0, 256, (batch, hh, ww), dtype=torch.uint8, device=dev)
            y = torch.empty((batch, hh // 2, ww // 2, 64), device=dev)
            w1a, b1a = torch.randn((9, 64), device=dev) * 0.3, torch.zeros(64, device=dev)
            wp = torch.randn(lib.gtsfm_packed_conv3x3_floats(64, 64), device=dev) * 0.05
            bias = torch.zeros(64, device=dev)
            fargs = (img.data_ptr(), 1, w1a.data_ptr(), b1a.data_ptr(), wp.data_ptr(), bias.data_ptr(), batch, hh, ww, 1, y.data_ptr(), stream.cuda_stream)
            out.append(("fused first", 2.0 * 9 * 64 * 64 * hh * ww * batch, (lambda fargs=fargs: L.check(lib.gtsfm_conv1_fused_f32(*fargs), "c1")), (img, y, w1a, b1a, wp, bias)))
            continue
        x = torch.randn((batch, hh, ww, cin), device=dev)
        ho, wo = (hh // 2, ww // 2) if pool else (hh, ww)
        y = torch.empty((batch, ho, wo, cout), device=dev)
        wp = torch.randn(lib.gtsfm_packed_conv3x3_floats(cin, cout), device=dev) * 0.05
        bias = torch.zeros((cout + 63) // 64 * 64, device=dev)
        args = (x.data_ptr(), cin, 0, y.data_ptr(), cout, 0, wp.data_ptr(), bias.data_ptr(), batch, hh, ww, cin, cout, 1, pool, stream.cuda_stream)
        out.append((f"{cin}->{cout} @{hh}{'p' if pool else ''}", 2.0 * 9 * cin * cout * hh * ww * batch, (lambda args=args: L.check(lib.gtsfm_conv3x3_f32(*args), "c3")), (x, y, wp, bias)))
    return out

ls = layers()
for name in ("run 1", "run 2"):
    tot_ms = tot_fl = 0.0
    row = []
    for lname, flops, fn, _ in ls:
        ms = bench._time_launches(fn, stream, 5)
        tot_ms += ms; tot_fl += flops
        row.append(f"{lname} {flops / ms / 1e9 / 157.3:.3f}")
    print(f"{name:22s} stack {tot_fl / tot_ms / 1e9 / 157.3:.3f} ({tot_ms:.2f} ms) | " + " | ".join(row), flush=True)
