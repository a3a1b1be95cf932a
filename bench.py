"""Benchmark of the MI355X deep front-end (BASELINE.json: image-pairs/sec, detect+match @1024 px).

    python bench.py --gpus N --steps K --warmup W            (N > 1 re-launches itself, one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W               (the driver's form; RANK / WORLD_SIZE come from the env)

One "step" = one pass of the hot path over one batch of synthetic input resident in HBM.

``--mode replica`` (default = BASELINE config 3 AS WRITTEN: SuperPoint+LightGlue, 46 synthetic 1024x1024 views, the first 1000 of their
1035 exhaustive pairs): every rank owns ``--images`` seeded 1024x1024 gray views, detects + describes them (SuperPoint) and matches
the first ``--pairs`` exhaustive (i<j) pairs (``--matcher``) at the REFERENCE'S keypoint cap: ``max_keypoints = 5000``
(gtsfm/configs/deep_front_end.yaml:29; the synthetic views yield ~8 200 raw detections, so every image is matched at N = 5000).
The other BASELINE configs ride in the same line under ``secondary``: config 2 (``config2_superpoint_480x640``) and one GPU's share
of config 4 at the cap (``config4_scene_share_cap5000``), each with its own timed region and parity check. Weak scaling, no data-path collective (images and pairs are
independent units); the packed weights are broadcast from rank 0 over RCCL. SURVEY.md section 8(d)'s
"additionally reported" N = 2048 rate, SuperGlue with 20 / 100 Sinkhorn iterations, independent pairs and the per-call
plugin API are timed in the same line under ``secondary``.

``--mode scene`` (BASELINE config 4): ONE scene of ``--images`` (101) views / ``--pairs`` (5000) exhaustive pairs is
sharded over the ranks: cyclic image ownership for detection, one RCCL all-gather of the feature table, 2-D
block-cyclic pair ownership for matching (SuperGlue, 100 Sinkhorn iterations by default), ragged gather of the match
lists. Strong scaling: ``value`` = the scene's pairs / max-over-ranks time.

Rank 0 prints ONE JSON line: the whole-job rate, the roofline of the dominant kernel (measured live with HIP events
on the launch stream), at N = 1 a CPU baseline (the oracle, timed on a bounded sample of the same workload), a
``parity_check`` of the GPU result against that oracle run on the same two images, and ``secondary`` rates.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

from gtsfm_amd.utils import synthetic  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16: 16 x the fp32 rate)
HBM_PEAK_GBS = 8000.0          # same guide: HBM3E spec peak (6.3 TB/s is what a streaming copy reaches)


# ------------------------------------------------------------------------------------------------------------------
# Work formulas
# ------------------------------------------------------------------------------------------------------------------


def superpoint_conv3x3_layers(h: int, w: int):
    """(cin, cout, h, w, pool) of the nine conv3x3 MFMA launches of one SuperPoint forward (superpoint.py:120-131;
    convPa|convDa are fused into one 128->512 launch)."""
    h2, w2, h4, w4, hc, wc = h // 2, w // 2, h // 4, w // 4, h // 8, w // 8
    return [
        (64, 64, h, w, 1), (64, 64, h2, w2, 0), (64, 64, h2, w2, 1), (64, 128, h4, w4, 0), (128, 128, h4, w4, 1),
        (128, 128, hc, wc, 0), (128, 128, hc, wc, 0), (128, 512, hc, wc, 0),
    ]


def superpoint_flops(h: int, w: int) -> float:
    """sum over the 12 convolutions of 2*k^2*Cin*Cout*H*W (177.85 GFLOP at 1024x1024)."""
    hc, wc = h // 8, w // 8
    total = 2 * 9 * 1 * 64 * h * w
    total += sum(2 * 9 * cin * cout * hh * ww for cin, cout, hh, ww, _ in superpoint_conv3x3_layers(h, w))
    total += 2 * 256 * 65 * hc * wc + 2 * 256 * 256 * hc * wc
    return float(total)


def matcher_flops(matcher: str, n: int, layers: float, sinkhorn: int) -> float:
    """SURVEY.md section 8(d) per-pair dense FLOP (N = M keypoints)."""
    if matcher == "superglue":
        return 2 * 217280 * n + 36 * (1310720 * n + 1024 * n * n) + 262144 * n + 512 * n * n
    return layers * 2 * (2490368 * n + 1792 * n * n) + 262144 * n + 512 * n * n


def folded_projection_flops(matcher: str, n: int, layers: float) -> float:
    """FLOPs of SURVEY's formula that the shipped path never executes: the attention output projections (SuperGlue ``attn.merge``, LightGlue
    ``out_proj`` / ``to_out``: one 256 x 256 product per block and image) are folded into the following layer's weights at load time
    (matcher_engine._fold_projection). Per pair: blocks x 2 images x 2 * 256 * 256 * N."""
    blocks = 18.0 if matcher == "superglue" else 2.0 * layers
    return blocks * 2 * 2.0 * 256 * 256 * n


def first_block_flops(matcher: str, n: int) -> float:
    """Dense FLOP of the block of the first matcher layer that sees ONE image (N keypoints): SuperGlue's keypoint encoder
    (217 280 N) + one GNN layer (1 310 720 N + 1024 N^2); LightGlue's first self block (Wqkv, out_proj, FFN: 1 310 720 N;
    self-attention 1024 N^2). With --share-first-layer it is counted once per image instead of twice per pair."""
    return (217280 * n if matcher == "superglue" else 0) + 1310720 * n + 1024 * n * n


PMC_TRAFFIC_FILE = "r06_pmc_traffic.json"  # collected on THIS round's tree (tools/prof_r06.sh); older rounds' files are history, never cited


def pmc_traffic(kernel: str):
    """HBM-side bytes per launch measured with rocprofv3 PMC passes on this round's tree and committed under profiles/ (bench.py cannot
    collect counters itself); None when the file does not hold the kernel -- a figure from an earlier round's kernels is never printed
    under this round's headline."""
    try:
        entry = json.loads((REPO / "profiles" / PMC_TRAFFIC_FILE).read_text()).get(kernel)
    except (OSError, ValueError):
        entry = None
    return None if entry is None else dict(entry, source=f"profiles/{PMC_TRAFFIC_FILE}")


# ------------------------------------------------------------------------------------------------------------------
# Per-kernel rooflines, measured live with HIP events on the launch stream
# ------------------------------------------------------------------------------------------------------------------


ROOFLINE_WARM_S = 0.25   # a roofline leg starts on a chip that has idled (host-side legs ran before it): launches for this long before the timed ones,
ROOFLINE_TIMED_S = 0.15  # and at least this much timed work, so that the figure is the kernel at the clocks the workload holds, not the DVFS ramp


def _time_launches(fn, stream, reps: int) -> float:
    """Average launch duration by HIP events on the launch stream. Round 6: five launches right after an idle phase measured the clock ramp on some
    boxes (the attention launch 6.65 ms in the bench line against 6.25 ms in the kernel trace of the same run and in tools/bench_attention.py on the
    same box): the timed launches now follow ROOFLINE_WARM_S of the same launches and cover at least ROOFLINE_TIMED_S."""
    import time

    fn()
    torch.cuda.synchronize()
    t0, warm = time.perf_counter(), 1
    while time.perf_counter() - t0 < ROOFLINE_WARM_S:
        fn()
        torch.cuda.synchronize()
        warm += 1
    per = (time.perf_counter() - t0) / max(1, warm - 1)
    reps = max(reps, min(2000, int(ROOFLINE_TIMED_S / max(per, 1e-6)) + 1))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def measure_conv_roofline(lib, device, batch: int, h: int, w: int, reps: int = 5):
    """conv3x3_mfma_kernel launch by launch over the SuperPoint stack."""
    from gtsfm_amd.runtime import lib as L

    stream = torch.cuda.current_stream(device)
    total_ms, total_flops, launches, first = 0.0, 0.0, 0, {}
    for li, (cin, cout, hh, ww, pool) in enumerate(superpoint_conv3x3_layers(h, w)):
        if li == 0:  # the first launch as gtsfm_sp_forward runs it: relu(conv1a(u8 image / 255)) recomputed in conv1b's halo staging
            img = torch.randint(0, 256, (batch, hh, ww), dtype=torch.uint8, device=device)
            y = torch.empty((batch, hh // 2, ww // 2, 64), device=device)
            w1a, b1a = torch.randn((9, 64), device=device) * 0.3, torch.zeros(64, device=device)
            wp = torch.randn(lib.gtsfm_packed_conv3x3_floats(64, 64), device=device) * 0.05
            bias = torch.zeros(64, device=device)
            fargs = (img.data_ptr(), 1, w1a.data_ptr(), b1a.data_ptr(), wp.data_ptr(), bias.data_ptr(), batch, hh, ww, 1, y.data_ptr(), stream.cuda_stream)
            ms0 = _time_launches(lambda: L.check(lib.gtsfm_conv1_fused_f32(*fargs), "conv1_fused"), stream, reps)
            first = {"first_layer_fused_ms": round(ms0, 4), "first_layer_fused_frac": round(2.0 * 9 * 64 * 64 * hh * ww * batch / (ms0 * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)}
            total_ms += ms0
            total_flops += 2.0 * 9 * 64 * 64 * hh * ww * batch  # conv1b's FLOPs; the recomputed conv1a (1.6 %) is not counted
            launches += 1
            del img, y, wp
            continue
        x = torch.randn((batch, hh, ww, cin), device=device)
        ho, wo = (hh // 2, ww // 2) if pool else (hh, ww)
        y = torch.empty((batch, ho, wo, cout), device=device)
        wp = torch.randn(lib.gtsfm_packed_conv3x3_floats(cin, cout), device=device) * 0.05
        bias = torch.zeros((cout + 63) // 64 * 64, device=device)
        args = (x.data_ptr(), cin, 0, y.data_ptr(), cout, 0, wp.data_ptr(), bias.data_ptr(), batch, hh, ww, cin, cout, 1, pool,
                stream.cuda_stream)
        total_ms += _time_launches(lambda: L.check(lib.gtsfm_conv3x3_f32(*args), "conv3x3"), stream, reps)
        total_flops += 2.0 * 9 * cin * cout * hh * ww * batch
        launches += 1
        del x, y, wp
    achieved = total_flops / (total_ms * 1e-3) / 1e12
    t = pmc_traffic("conv3x3_mfma_kernel") if (h, w) == (1024, 1024) else None
    return {
        "bound": "mfma", "kernel": "conv3x3_mfma_kernel", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
        "traffic": None if t is None else t["fetch_bytes_per_image"] + t["write_bytes_per_image"],
        "algorithmic_bytes": None if t is None else t.get("algorithmic_bytes_per_image"),
        "traffic_note": None if t is None else ("HBM-side bytes PER IMAGE over the whole 8-launch stack (fetched + written), beside algorithmic_bytes = every layer's input read once + "
                                                f"every layer's output written once per image; rocprofv3 PMC, {t['source']}"),
        "launches_per_step": launches, "avg_launch_ms": round(total_ms / launches, 4), "flops_per_step": total_flops, **first,
    }


def measure_attention_roofline(lib, device, n: int, npairs: int, reps: int = 5, math: int = 0):
    """Dominant kernel of the detect+match workload (~50 % of GPU time): one launch = the self attention of `npairs`
    pairs (2 sequences x 4 heads each) at N = n. Algorithmic work: 1024 * N^2 FLOP per sequence per layer (SURVEY.md
    section 8a rows a23 / a36). math 1 = the opt-in bf16x3 arithmetic (6 bf16 MFMAs per 32 x 32 x 16 block of either product), math 2 = the opt-in
    f16x2 arithmetic (3 fp16 MFMAs): the same algorithmic FLOPs, priced against the same fp32 roof AND as executed 16-bit work against the bf16 / fp16
    MFMA roof (the same nominal 2.5 PFLOP/s)."""
    from gtsfm_amd.runtime import lib as L

    stream = torch.cuda.current_stream(device)
    nseq = 2 * npairs
    cap = -(-n // 128) * 128  # LightGlue aligns every keypoint set to 128 rows
    qkv = torch.randn((nseq * cap, 768), device=device)
    out = torch.empty((nseq * cap, 256), device=device)
    probs = torch.tensor([[s * cap, s, s * cap, s] for s in range(nseq)], dtype=torch.int32, device=device)
    counts = torch.full((nseq,), n, dtype=torch.int32, device=device)
    # the matchers' call: workspace for the schedule the launch geometry picks (mode 0) -- fused with the merged state parked in the workspace here
    ws = torch.empty(max(256, int(lib.gtsfm_attention_math_workspace_bytes(nseq, n, n, 4, nseq * cap, math))), dtype=torch.uint8, device=device)
    args = (qkv.data_ptr(), 768, qkv.data_ptr() + 256 * 4, 768, qkv.data_ptr() + 512 * 4, 768, out.data_ptr(), 256, probs.data_ptr(),
            counts.data_ptr(), nseq, n, n, 4, 0.125, 0, math, nseq * cap, ws.data_ptr(), ws.numel(), stream.cuda_stream)
    ms = _time_launches(lambda: L.check(lib.gtsfm_attention_math_f32(*args), "attention"), stream, reps)
    flops = 1024.0 * n * n * nseq
    achieved = flops / (ms * 1e-3) / 1e12
    if math in (1, 2):
        per_term = 6.0 if math == 1 else 3.0  # bf16x3: six bf16 products per fp32 product term; f16x2: three fp16 products
        name = "bf16x3" if math == 1 else "f16x2"
        executed = per_term * flops
        tx, ts = pmc_traffic(f"attention_x3_kernel@{nseq}x4x{n}@{name}"), pmc_traffic(f"attention_x3_split_kernel@{nseq}x4x{n}@{name}")
        if math == 1 and tx is None:
            tx, ts = pmc_traffic(f"attention_x3_kernel@{nseq}x4x{n}"), pmc_traffic(f"attention_x3_split_kernel@{nseq}x4x{n}")
        traffic = None if tx is None or ts is None else tx["fetch_bytes"] + tx["write_bytes"] + ts["fetch_bytes"] + ts["write_bytes"]
        return {
            "bound": "mfma", "kernel": f"attention_x3_split_kernel<{3 if math == 1 else 2}> + attention_x3_kernel<.., {3 if math == 1 else 2}>", "achieved": round(executed / (ms * 1e-3) / 1e12, 2),
            "peak": BF16_MFMA_PEAK_TFLOPS, "unit": f"TFLOP/s (executed {'bf16' if math == 1 else 'fp16'})",
            "frac": round(executed / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4), "avg_launch_ms": round(ms, 4),
            "algorithmic_tflops": round(achieved, 2), "algorithmic_frac_of_fp32_mfma_peak": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "flops_per_launch": flops,
            "launch_shape": f"{nseq} sequences x 4 heads, N = {n} queries = keys, head_dim 64 (K / V split pass + attention)", "traffic": traffic,
            "traffic_note": None if traffic is None else f"HBM-side bytes per launch (split pass + attention), rocprofv3 PMC, {tx['source']}",
        }
    t = pmc_traffic(f"attention_dma_kernel@{nseq}x4x{n}")
    return {
        "bound": "mfma", "kernel": "attention_dma_kernel", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
        "traffic": None if t is None else t["fetch_bytes"] + t["write_bytes"],
        "traffic_note": None if t is None else f"HBM bytes per launch, rocprofv3 PMC, {t['source']}",
        "avg_launch_ms": round(ms, 4), "flops_per_launch": flops,
        "launch_shape": f"{nseq} sequences x 4 heads, N = {n} queries = keys, head_dim 64",
    }


def measure_gemm_roofline(lib, device, rows: int, k: int, n: int, reps: int = 5):
    """The matchers' projection GEMM at one of their shapes (rows x k -> n), row-major weights (LDS-DMA kernel). Under GTSFM_GEMM_MATH=bf16x3
    (read by the entry point) the kernel executes six bf16 MFMA products per fp32 product term: `frac` is then EXECUTED bf16 FLOP/s over the
    bf16 MFMA roof, and the algorithmic fp32 rate against the fp32 roof is a side field -- no `frac` in the line exceeds 1."""
    from gtsfm_amd.runtime import lib as L

    stream = torch.cuda.current_stream(device)
    a = torch.randn((rows, k), device=device)
    w = torch.randn((n, k), device=device) * 0.05
    bias = torch.zeros((n + 63) // 64 * 64, device=device)
    c = torch.empty((rows, n), device=device)
    args = (a.data_ptr(), k, rows, None, k, w.data_ptr(), k, bias.data_ptr(), n, None, c.data_ptr(), n, 0, None, 0, 1.0, 0, stream.cuda_stream)
    ms = _time_launches(lambda: L.check(lib.gtsfm_linear_rowmajor_f32(*args), "linear_rowmajor"), stream, reps)
    achieved = 2.0 * rows * k * n / (ms * 1e-3) / 1e12
    gm = os.environ.get("GTSFM_GEMM_MATH") or ""
    if gm[:1] == "b" or gm[:2] == "f1":
        x3 = gm[:1] == "b"
        executed = (6.0 if x3 else 3.0) * achieved
        return {
            "bound": "mfma", "kernel": f"gemm_dma_walk_kernel<.., {1 if x3 else 2}>", "achieved": round(executed, 2), "peak": BF16_MFMA_PEAK_TFLOPS,
            "unit": f"TFLOP/s (executed {'bf16' if x3 else 'fp16'})",
            "frac": round(executed / BF16_MFMA_PEAK_TFLOPS, 4), "avg_launch_ms": round(ms, 4), "launch_shape": f"{rows} x {k} -> {n}",
            "algorithmic_tflops": round(achieved, 2), "algorithmic_frac_of_fp32_mfma_peak": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
            "arithmetic": ("bf16x3: six bf16 MFMA products per fp32 product term, operands split in registers" if x3
                           else "f16x2: three fp16 MFMA products per fp32 product term, operands split in registers"), "traffic": None,
        }
    t = pmc_traffic(f"gemm_dma_walk_kernel@{rows}x{k}x{n}")
    return {
        "bound": "mfma", "kernel": "gemm_dma_walk_kernel", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "avg_launch_ms": round(ms, 4), "launch_shape": f"{rows} x {k} -> {n}",
        "algorithmic_bytes": 4 * (rows * k + n * k + rows * n),
        "traffic": None if t is None else t["fetch_bytes"] + t["write_bytes"],
        "traffic_note": None if t is None else f"HBM bytes per launch, rocprofv3 PMC, {t['source']}",
    }


def measure_score_gemm_roofline(lib, device, n: int, npairs: int, reps: int = 5):
    """The matchers' score GEMM (superglue.py:257-258, LightGlue's sim = mdesc0 mdesc1^T) AS THE WORKLOAD LAUNCHES IT: the `npairs` pairs of a chunk in
    one ragged launch (gtsfm_score_matrices_f32), each N x 256 -> N with image 1's descriptor rows standing in for the weights as they lie. Until round 6
    this entry timed ONE pair through the plain linear entry point: 1600 output tiles on 512 workgroup slots, 3.1 rounds -- a launch no batched step makes
    (`one_pair_frac` keeps that figure: it is what the per-call plugin path runs)."""
    from gtsfm_amd.runtime import lib as L

    one = measure_gemm_roofline(lib, device, n, 256, n)
    if not hasattr(lib, "gtsfm_score_matrices_f32"):
        one["launch_shape"] = f"score matrix of one pair: {n} x 256 -> {n}"
        return one
    stream = torch.cuda.current_stream(device)
    m = np.full((npairs,), n, dtype=np.int32)
    ld = (n + 1 + 3) // 4 * 4
    mdesc = torch.randn((2 * npairs * n, 256), device=device)
    z = torch.empty((npairs * (n + 1) * ld,), device=device)
    ws = torch.empty(int(lib.gtsfm_score_matrices_workspace_bytes(npairs)), dtype=torch.uint8, device=device)
    args = (mdesc.data_ptr(), npairs, m.ctypes.data, m.ctypes.data, 0.0625, z.data_ptr(), ws.data_ptr(), ws.numel(), stream.cuda_stream)
    # every call rebuilds and uploads the small batch descriptor and synchronises once (a stand-alone entry point's cost, not the forward's, which
    # keeps its descriptor resident): time the launch with events around a burst and subtract nothing -- the kernel is ~2 ms, the upload ~20 us
    ms = _time_launches(lambda: L.check(lib.gtsfm_score_matrices_f32(*args), "score_matrices"), stream, reps)
    flops = 2.0 * n * n * 256 * npairs
    achieved = flops / (ms * 1e-3) / 1e12
    if (os.environ.get("GTSFM_GEMM_MATH") or "")[:1] == "b":
        return {"bound": "mfma", "kernel": "gemm_dma_walk_kernel<X3> (score matrices)", "achieved": round(6.0 * achieved, 2), "peak": BF16_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s (executed bf16)", "frac": round(6.0 * achieved / BF16_MFMA_PEAK_TFLOPS, 4), "avg_launch_ms": round(ms, 4),
                "launch_shape": f"score matrices of {npairs} pairs in one ragged launch: {n} x 256 -> {n} each", "algorithmic_tflops": round(achieved, 2), "traffic": None}
    t = pmc_traffic(f"score_matrices@{npairs}x{n}")
    return {
        "bound": "mfma", "kernel": "gemm_dma_walk_kernel (score matrices)", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "avg_launch_ms": round(ms, 4),
        "launch_shape": f"score matrices of {npairs} pairs in one ragged launch: {n} x 256 -> {n} each",
        "algorithmic_bytes": 4 * npairs * (2 * n * 256 + n * n), "one_pair_frac": one.get("frac"), "one_pair_ms": one.get("avg_launch_ms"),
        "traffic": None if t is None else t["fetch_bytes"] + t["write_bytes"],
        "traffic_note": None if t is None else f"HBM bytes per launch, rocprofv3 PMC, {t['source']}",
    }


def measure_sinkhorn_roofline(lib, device, n: int, npairs: int, iters: int = 20):
    """One Sinkhorn iteration (row sweep + column combine) over the (N+1) x (N+1) couplings matrices of `npairs` pairs.
    HBM-bound: algorithmic bytes = one read of Z per iteration = 4 (N+1)^2 B per pair (SURVEY.md section 8a row a29:
    8 (N+1)^2 unfused)."""
    from gtsfm_amd.runtime import lib as L

    if not hasattr(lib, "gtsfm_sinkhorn_f32"):
        return None
    stream = torch.cuda.current_stream(device)
    ld = (n + 1 + 3) // 4 * 4
    z = torch.randn((npairs, n + 1, ld), device=device)
    m = torch.full((npairs,), n, dtype=torch.int32)
    ws = torch.empty(int(lib.gtsfm_sinkhorn_workspace_bytes(npairs, m.data_ptr(), m.data_ptr())), dtype=torch.uint8, device=device)
    u = torch.empty((npairs, n + 1), device=device)
    v = torch.empty((npairs, n + 1), device=device)

    def run(it):
        L.check(lib.gtsfm_sinkhorn_f32(z.data_ptr(), npairs, m.data_ptr(), m.data_ptr(), 1.0, it, ws.data_ptr(), ws.numel(), u.data_ptr(),
                                       v.data_ptr(), stream.cuda_stream), "sinkhorn")

    t1 = _time_launches(lambda: run(iters), stream, 3)
    t2 = _time_launches(lambda: run(2 * iters), stream, 3)
    ms_iter = (t2 - t1) / iters  # the fixed part (dustbin fill, descriptor upload) cancels
    bytes_iter = 4.0 * (n + 1) * (n + 1) * npairs
    achieved = bytes_iter / (ms_iter * 1e-3) / 1e9
    t = pmc_traffic(f"sinkhorn_iteration@{npairs}x{n}")
    return {
        "bound": "hbm", "kernel": ("sinkhorn_rows_kernel" if n <= 2048 else "sinkhorn_rows_wide_kernel") + " + sinkhorn_cols_kernel (one iteration)", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "avg_iteration_ms": round(ms_iter, 4), "algorithmic_bytes": bytes_iter,
        "launch_shape": f"{npairs} pairs, ({n}+1) x ({n}+1) couplings", "traffic": None if t is None else t["fetch_bytes"] + t["write_bytes"],
        "traffic_note": None if t is None else f"HBM bytes per iteration, rocprofv3 PMC, {t['source']}",
    }


def measure_lg_assignment_roofline(lib, device, n: int, npairs: int, reps: int = 5):
    """LightGlue's last stage on `npairs` N x N similarity matrices, its two halves timed separately through gtsfm_lg_assignment_f32: the two
    log-softmax sweeps (lg_rows[_wide]_kernel + lg_cols_kernel) and the match extraction (extract_rows[_wide]_kernel + extract_cols_kernel +
    mutual_matches) -- each ONE read of the matrices, 4 N^2 B per pair, HBM-bound. The extraction is the kernel furthest below its roof in
    the tree (compare / select chains on an under-occupied chip: DESIGN.md section 8); it is in the line for that reason."""
    from gtsfm_amd.runtime import lib as L

    if not hasattr(lib, "gtsfm_lg_assignment_f32"):
        return []
    stream = torch.cuda.current_stream(device)
    ld = (n + 3) // 4 * 4
    cap = -(-n // 128) * 128
    sim = torch.randn((npairs, n, ld), device=device) * 4.0
    zl = torch.randn((2 * npairs * cap,), device=device)
    m = torch.full((npairs,), n, dtype=torch.int32)
    ws = torch.empty(int(lib.gtsfm_lg_assignment_workspace_bytes(npairs, m.data_ptr(), m.data_ptr())), dtype=torch.uint8, device=device)
    matches = torch.empty((2 * npairs * cap,), dtype=torch.int32, device=device)
    ms = torch.empty((2 * npairs * cap,), dtype=torch.float32, device=device)

    def run(stages):
        L.check(lib.gtsfm_lg_assignment_f32(sim.data_ptr(), npairs, m.data_ptr(), m.data_ptr(), zl.data_ptr(), 0.1, stages, ws.data_ptr(), ws.numel(),
                                            matches.data_ptr(), ms.data_ptr(), stream.cuda_stream), "lg_assignment")

    run(3)  # uploads the batch descriptor; the timed calls below reuse it (stages + 4: launches only, no upload, no synchronisation)
    out = []
    bytes_pass = 4.0 * n * n * npairs
    for stages, kernel in ((1, ("lg_rows_kernel" if n <= 2048 else "lg_rows_wide_kernel") + " + lg_cols_kernel (double log-softmax)"),
                           (2, ("extract_rows_kernel" if n <= 2048 else "extract_rows_wide_kernel") + " + extract_cols_kernel + mutual_matches (match extraction)")):
        t = _time_launches(lambda: run(stages + 4), stream, reps)
        achieved = bytes_pass / (t * 1e-3) / 1e9
        pt = pmc_traffic(("lg_double_softmax" if stages == 1 else "lg_extract") + f"@{npairs}x{n}")
        out.append({"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "avg_launch_ms": round(t, 4), "algorithmic_bytes": bytes_pass, "launch_shape": f"{npairs} pairs, {n} x {n} similarities",
                    "traffic": None if pt is None else pt["fetch_bytes"] + pt["write_bytes"],
                    "traffic_note": None if pt is None else f"HBM-side bytes per pass, rocprofv3 PMC, {pt['source']}"})
    return out


def measure_layernorm_roofline(lib, device, rows: int, reps: int = 5):
    """layernorm_gelu_kernel over `rows` token rows of 512 floats, in place: 2 x rows x 512 x 4 B, HBM-bound (1.3 % of the headline step)."""
    from gtsfm_amd.runtime import lib as L

    if not hasattr(lib, "gtsfm_layernorm_gelu_f32"):
        return None
    stream = torch.cuda.current_stream(device)
    x = torch.randn((rows, 512), device=device)
    gamma, beta = torch.ones(512, device=device), torch.zeros(512, device=device)
    scratch = torch.empty(64, dtype=torch.uint8, device=device)
    t = _time_launches(lambda: L.check(lib.gtsfm_layernorm_gelu_f32(x.data_ptr(), 512, rows, gamma.data_ptr(), beta.data_ptr(), scratch.data_ptr(), stream.cuda_stream),
                                       "layernorm_gelu"), stream, reps)
    nbytes = 2.0 * rows * 512 * 4
    achieved = nbytes / (t * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "layernorm_gelu_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "avg_launch_ms": round(t, 4), "algorithmic_bytes": nbytes, "launch_shape": f"{rows} rows x 512 (read + written in place)", "traffic": None}


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle) and the parity check of the timed workload against it
# ------------------------------------------------------------------------------------------------------------------


def cpu_baseline(views: np.ndarray, matcher: str, n_keypoints: int, sinkhorn_iters: int):
    """The oracle (kind "port": restatement of the reference's torch CPU path, bit-exact with it in the build
    container) on a bounded sample: the two views of the workload's first pair are detected and matched once, on the
    host cores. Returns (baseline dict, oracle outputs for parity_check)."""
    from oracle import superpoint_oracle

    cores = min(os.cpu_count() or 1, int(os.environ.get("GTSFM_CPU_BASELINE_THREADS", "16")))
    torch.set_num_threads(cores)
    h, w = views.shape[1:]
    sd = synthetic.synthetic_superpoint_state_dict()
    t_det, feats = [], []
    for gray in views:
        t0 = time.perf_counter()
        c, s, d = superpoint_oracle.detect_and_describe(sd, gray, max_keypoints=1 << 30)  # the model's full row-major list
        sel = synthetic.topk_detection_order(s, n_keypoints)  # top-k by response, kept in detection order
        t_det.append(time.perf_counter() - t0)
        feats.append((c[sel], s[sel], d[sel]))
    det_s = float(np.median(t_det))
    out = {"detect_s_per_image": round(det_s, 3), "cores": cores, "kind": "port"}
    oracle_out = {"features": feats}
    if matcher == "none":
        out.update(value=round(1.0 / det_s, 4), unit="images/s", sample=f"{len(views)} x SuperPoint {h}x{w} (oracle, fp32)")
        return out, oracle_out
    (c0, s0, d0), (c1, s1, d1) = feats
    T = torch.from_numpy
    t0 = time.perf_counter()
    with torch.no_grad():
        if matcher == "superglue":
            from oracle import superglue_oracle

            sg = synthetic.synthetic_superglue_state_dict()
            res = superglue_oracle.superglue_forward(sg, T(c0)[None], T(c1)[None], T(s0)[None], T(s1)[None], T(d0).T[None].contiguous(),
                                                     T(d1).T[None].contiguous(), (h, w), (h, w), sinkhorn_iterations=sinkhorn_iters)
        else:
            from oracle import lightglue_oracle

            lg = synthetic.synthetic_lightglue_state_dict()
            res = lightglue_oracle.lightglue_forward(lg, T(c0)[None], T(c1)[None], T(d0)[None], T(d1)[None], (h, w), (h, w), return_intermediates=True)
    match_s = time.perf_counter() - t0
    oracle_out.update(matches0=res["matches0"][0].numpy(), matching_scores0=res["matching_scores0"][0].numpy())
    # independent-pair cost on the CPU path: 2 detections + 1 match
    out.update(
        match_s_per_pair=round(match_s, 3), value=round(1.0 / (2 * det_s + match_s), 4), unit="image-pairs/s",
        sample=f"2 x SuperPoint {h}x{w} + 1 x {matcher} pair at N={len(c0)},{len(c1)} (oracle, fp32); rate = 1 / (2 detect + 1 match)",
    )
    return out, oracle_out


def reference_cpu_run(matcher: str, n_keypoints: int, sinkhorn_iters: int):
    """The committed run of tools/cpu_baseline.py (SURVEY.md section 8d protocol: the reference's OWN model files, 1 warm-up + 5
    repetitions, median, in the build container where /root/reference is mounted), quoted next to the live single-sample figure."""
    try:
        ref = json.loads((REPO / "profiles" / "r03_cpu_baseline.json").read_text())
    except (OSError, ValueError):
        return None
    out = {"source": "profiles/r03_cpu_baseline.json (tools/cpu_baseline.py)", "protocol": ref["protocol"], "cores": ref["host"]["cores"],
           "superpoint_s_per_image": ref["superpoint"]["s_per_image"], "superpoint_kind": ref["superpoint"]["kind"]}
    key = f"n{n_keypoints}_sinkhorn{sinkhorn_iters}" if matcher == "superglue" else f"n{n_keypoints}"
    entry = ref.get(matcher, {}).get(key)
    if entry is not None:
        out.update(matcher=matcher, matcher_kind=entry["kind"], match_s_per_pair=entry["s_per_pair"], independent_pairs_per_s=entry["independent_pairs_per_s"])
    out["superglue_s_per_pair"] = {k: v["s_per_pair"] for k, v in ref.get("superglue", {}).items()}
    return out


def parity_check(oracle_out, gpu_feats, gpu_rows, gpu_match):
    """GPU result of the timed workload vs the oracle on the SAME two images (the first pair of the step): keypoints
    bit-exact, scores / descriptors / match scores within 1e-4, match indices bit-exact."""
    out = {}
    kp_equal, dscore, ddesc = True, 0.0, 0.0
    for (c, s, d), row in zip(oracle_out["features"], gpu_rows):
        k = int(gpu_feats["count"][row])
        xy = gpu_feats["xy"][row, :k].cpu().numpy()
        same = xy.shape == c.shape and bool(np.array_equal(xy, c))
        kp_equal &= same
        if same:
            dscore = max(dscore, float(np.abs(gpu_feats["scores"][row, :k].cpu().numpy() - s).max()))
            ddesc = max(ddesc, float(np.abs(gpu_feats["descriptors"][row, :k].cpu().numpy() - d).max()))
    out.update(keypoints_equal=bool(kp_equal), keypoints=[len(f[0]) for f in oracle_out["features"]],
               max_dscore_keypoints=dscore, max_ddescriptor=ddesc)
    if gpu_match is not None and "matches0" in oracle_out:
        m0, ms0 = gpu_match
        ref = oracle_out["matches0"]
        equal = m0.shape == ref.shape and bool(np.array_equal(m0.astype(np.int64), ref.astype(np.int64)))
        out.update(matches_equal=equal, matches=int((ref > -1).sum()),
                   max_dscore=float(np.abs(ms0 - oracle_out["matching_scores0"]).max()) if equal else None)
        out["within_tolerance"] = bool(kp_equal and equal and dscore < 1e-4 and ddesc < 1e-4 and out["max_dscore"] < 1e-4)
    else:
        out["within_tolerance"] = bool(kp_equal and dscore < 1e-4 and ddesc < 1e-4)
    return out


# ------------------------------------------------------------------------------------------------------------------
# Launch plumbing
# ------------------------------------------------------------------------------------------------------------------


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def relaunch_one_process_per_gpu(gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run (one
    process per GPU, rendezvous on 127.0.0.1) and hand their output through."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))


def fewest_images_for(pairs: int) -> int:
    """Smallest n with n (n - 1) / 2 >= pairs (P exhaustive pairs <=> n images)."""
    n = 2
    while n * (n - 1) // 2 < pairs:
        n += 1
    return n


def default_pair_chunk(keypoints: int) -> int:
    """Pairs per launch sequence: a power of two in 4 .. 32 with at most ~82 k keypoints per image side of a chunk (16 pairs and ~3.5 GB of
    workspace per stream at the 5000 cap: the LightGlue rate does not depend on it -- 8 / 16 / 25 / 32 pairs: 111.7 / 111.7 / 112.1 / 112.2
    image-pairs/s -- but SuperGlue's Sinkhorn sweeps fill the chip better with 16 score matrices in flight than with 8)."""
    c = 32
    while c > 4 and c * keypoints > 82000:
        c //= 2
    return c


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=["replica", "scene"], default="replica",
                    help="replica: every rank owns its own image set / pair list (BASELINE config 3, weak scaling); scene: one scene "
                         "sharded over the ranks (BASELINE config 4, strong scaling)")
    ap.add_argument("--images", type=int, default=None, help="images per step: per rank (replica, default: the fewest whose exhaustive pairs cover --pairs) or of the scene (default 101)")
    ap.add_argument("--pairs", type=int, default=None,
                    help="exhaustive (i<j) pairs matched per step: per rank (replica, default 1000 = BASELINE config 3 as written: the first 1000 of "
                         "the 1035 exhaustive pairs of 46 views, an ~8.5 s step at the 5000-keypoint cap) or of the scene (default 5000)")
    ap.add_argument("--size", type=int, default=1024, help="square image side (overridden by --height / --width)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--matcher", choices=["none", "lightglue", "superglue"], default=None, help="default: lightglue (replica), superglue (scene)")
    ap.add_argument("--keypoints", type=int, default=5000,
                    help="keypoints kept per image (device top-k by response); default = the reference's cap, gtsfm/configs/deep_front_end.yaml:29")
    ap.add_argument("--sinkhorn", type=int, default=100, help="SuperGlue Sinkhorn iterations (GTSfM runs 20; BASELINE config 4 asks for 100)")
    ap.add_argument("--pair-definition", choices=["exhaustive", "independent"], default="exhaustive",
                    help="exhaustive: (i<j) pairs of --images images, each detected once per step (the headline); independent: "
                         "--pairs disjoint pairs, 2 fresh detections per pair (SURVEY.md section 8d asks for both rates)")
    ap.add_argument("--pair-chunk", type=int, default=0, help="pairs per matcher launch sequence; 0 = by keypoint count (32 at N <= 2048, 8 at the 5000 cap)")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams the pair chunks alternate over")
    ap.add_argument("--graphs", type=int, default=1, help="1: full pair chunks replay a captured hipGraph of the matcher's launch sequence")
    ap.add_argument("--share-first-layer", type=int, default=1,
                    help="1: the matcher block that sees one image (SuperGlue: keypoint encoder + first self layer; LightGlue: first self block) "
                         "runs once per image per step when the pair list reuses images; 0: once per pair side, as the per-pair plugin API does")
    ap.add_argument("--arithmetic", choices=["f32", "bf16x3", "f16x2"], default="f32",
                    help="f32 (default): the headline, exact fp32 everywhere. bf16x3 / f16x2: the WHOLE run under both opt-in switches (GTSFM_ATTENTION_MATH and "
                         "GTSFM_GEMM_MATH; DESIGN.md section 4) -- a side measurement of the same step protocol, labelled in `dtype` and `config.arithmetic`, never the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary rates")
    ap.add_argument("--no-roofline", action="store_true",
                    help="skip the per-kernel roofline micro-launches (for rocprofv3 --kernel-trace runs whose stats should hold the workload's launches only)")
    ap.add_argument("--details-file", default=DETAILS_FILE,
                    help="where the full record goes (secondary legs, every kernel's roofline, notes); stdout carries only the < 4 KB headline line. '' = nowhere")
    ap.add_argument("--dump-matches", type=int, default=0, help="1: add match_digest (SHA-1 over the last step's (K,2) match arrays in pair order) to the line")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help='"nccl" = RCCL; "gloo" only with --plumbing-only')
    ap.add_argument("--plumbing-only", action="store_true",
                    help="no GPU work: random features / matches stand in for the kernels so that the launcher, partitioning, collectives "
                         "and timing protocol can be exercised on CPU (gloo); the printed line is marked and is NOT a measurement")
    args = ap.parse_args(argv)
    scene = args.mode == "scene"
    args.pairs = args.pairs if args.pairs is not None else (5000 if scene else 1000)  # BASELINE config 3: 1000 exhaustive pairs <=> 46 views
    if args.images is None:
        args.images = 101 if scene else fewest_images_for(args.pairs)
    if args.pair_chunk <= 0:
        args.pair_chunk = default_pair_chunk(args.keypoints)
    args.matcher = args.matcher if args.matcher is not None else ("superglue" if scene else "lightglue")
    if scene and args.matcher == "none":
        ap.error("--mode scene needs a matcher")
    if scene and args.pair_definition != "exhaustive":
        ap.error("--mode scene matches the exhaustive pairs of one scene")
    if args.plumbing_only:
        args.backend = "gloo"
    return args


def make_scene_generator(args, plumbing: bool):
    """The product class of ``--mode scene``, built the way a GTSfM config builds it: from the two plugin objects (checkpoints = the seeded
    synthetic weights, written to a temporary directory by every rank for itself; only rank 0 ever reads them -- the others receive the packed
    blobs over RCCL). ``--plumbing-only``: the same class around the stand-in pipeline (no kernels, CPU, gloo)."""
    import tempfile
    import types

    from gtsfm_amd.frontend.correspondence_generator.sharded_det_desc_correspondence_generator import ShardedDetDescCorrespondenceGenerator

    if plumbing:
        from gtsfm_amd.utils.standin import stand_in_pipeline_factory

        return ShardedDetDescCorrespondenceGenerator(None, types.SimpleNamespace(max_keypoints=args.keypoints), pipeline_factory=stand_in_pipeline_factory), None
    from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor
    from gtsfm_amd.frontend.matcher.lightglue_matcher import LightGlueMatcher
    from gtsfm_amd.frontend.matcher.superglue_matcher import SuperGlueMatcher

    tmp = tempfile.TemporaryDirectory()
    torch.save(synthetic.synthetic_superpoint_state_dict(), f"{tmp.name}/sp.pth")
    det = SuperPointDetectorDescriptor(max_keypoints=args.keypoints, weights_path=f"{tmp.name}/sp.pth")
    if args.matcher == "superglue":
        torch.save(synthetic.synthetic_superglue_state_dict(), f"{tmp.name}/sg.pth")
        mt = SuperGlueMatcher(weights_path=f"{tmp.name}/sg.pth")
    else:
        torch.save(synthetic.synthetic_lightglue_state_dict(), f"{tmp.name}/lg.pth")
        mt = LightGlueMatcher("superpoint", weights_path=f"{tmp.name}/lg.pth")
    gen = ShardedDetDescCorrespondenceGenerator(mt, det, pair_batch=args.pair_chunk, pipeline_options={
        "num_streams": args.streams, "use_graphs": bool(args.graphs), "share_first_layer": bool(args.share_first_layer)})
    return gen, tmp  # the directory lives as long as the caller holds it (attach() reads the checkpoints on rank 0)


def matches_to_numpy(results):
    from gtsfm_amd.runtime.pipeline import FrontEndPipeline

    return FrontEndPipeline.matches_to_numpy(results)



# ------------------------------------------------------------------------------------------------------------------
# The printed line: a headline a driver can read (< 4 KB, the LAST and only stdout line); everything else goes to a side file
# ------------------------------------------------------------------------------------------------------------------

HEADLINE_BYTES = 4000  # the driver keeps an 8 KB tail of stdout + stderr: a 21 KB line (round 5) left it nothing to parse
DETAILS_FILE = "gpurun_out/bench_details.json"
_HEADLINE_ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "avg_iteration_ms", "launch_shape")


def _brief_roofline(r: dict) -> dict:
    out = {k: r[k] for k in _HEADLINE_ROOFLINE_KEYS if k in r}
    note = (r.get("traffic_note") or "").lower()
    if r.get("traffic") is not None:  # what the PMC byte count is per (the full note with its counter file is in the details file)
        out["traffic_per"] = next((unit for unit in ("image", "iteration", "pass") if f"per {unit}" in note), "launch")
    return out


def _brief_secondary(sec: dict) -> dict:
    """name -> rate of every secondary leg (the legs themselves, with their workloads, parity checks and CPU baselines, are in the details file)."""
    out = {}
    for name, leg in sec.items():
        if not isinstance(leg, dict):
            continue
        if "error" in leg:
            out[name] = "error"
        elif "value" in leg:
            out[name] = leg["value"]
        else:  # a group of rates (secondary_rates returns {name: {value, ...}})
            for sub, subleg in leg.items():
                if isinstance(subleg, dict) and "value" in subleg:
                    out[sub] = subleg["value"]
    return out


def headline_of(result: dict, details_path) -> dict:
    """The object the driver parses: metric / value / unit / n_gpus / steps / warmup / ms_per_step / dtype / config / roofline / cpu_baseline /
    parity_check, at most three of the other kernels' rooflines (the conv stack, the largest projection GEMM and whichever kernel sits furthest
    below its roof), and the secondary legs as bare rates. Fields are dropped from the end of that list until the line fits HEADLINE_BYTES."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "algorithmic_tflops", "executed_tflops", "step_frac_of_fp32_mfma_peak", "plumbing_only", "match_digest", "scene_check", "exchange", "distributed")
    head = {k: result[k] for k in keep if k in result}
    if "roofline" in result:
        head["roofline"] = _brief_roofline(result["roofline"])
        if result["roofline"].get("flops_per_launch") is not None:
            head["roofline"]["flops_per_launch"] = result["roofline"]["flops_per_launch"]
    if "cpu_baseline" in result:
        head["cpu_baseline"] = {k: v for k, v in result["cpu_baseline"].items() if k != "reference_run"}
    if "parity_check" in result:
        head["parity_check"] = result["parity_check"]
    optional = []
    other = [r for r in result.get("roofline_other", []) if isinstance(r, dict) and "frac" in r]
    if other:
        picked = []
        for want in ("conv3x3_mfma_kernel", "gemm_dma_walk_kernel"):
            hit = next((r for r in other if r.get("kernel") == want), None)
            if hit is not None:
                picked.append(hit)
        rest = [r for r in other if not any(r is q for q in picked)]
        if rest:
            picked.append(min(rest, key=lambda r: r["frac"]))
        optional.append(("roofline_other", [_brief_roofline(r) for r in picked[:3]]))
    if isinstance(result.get("secondary"), dict):
        optional.append(("secondary_rates", _brief_secondary(result["secondary"])))
    if details_path is not None:
        head["details"] = str(details_path)
    for key, val in optional:
        head[key] = val
    droppable = [k for k, _ in reversed(optional)] + ["exchange", "distributed", "scene_check", "parity_check"]
    while len(json.dumps(head)) > HEADLINE_BYTES and droppable:
        head.pop(droppable.pop(0), None)
    if len(json.dumps(head)) > HEADLINE_BYTES:  # the config strings are the only thing left that can grow
        head["config"] = {k: (v if not isinstance(v, str) or len(v) <= 120 else v[:117] + "...") for k, v in head["config"].items()}
    return head


def emit(result: dict, details_file: str) -> str:
    """Write the full record (every leg, every roofline, the notes) to `details_file` and return the headline line for stdout."""
    path = None
    if details_file:
        path = Path(details_file)
        path = path if path.is_absolute() else REPO / path
        try:
            path.parent.mkdir(parents=True, exist_ok=True)
            path.write_text(json.dumps(result, indent=1) + "\n")
            path = path.relative_to(REPO) if path.is_relative_to(REPO) else path
        except OSError as exc:  # a read-only tree must not cost the run its headline
            print(f"bench.py: could not write {path}: {exc}", file=sys.stderr)
            path = None
    line = json.dumps(headline_of(result, path))
    assert len(line) <= HEADLINE_BYTES, len(line)
    return line

# ------------------------------------------------------------------------------------------------------------------


def main() -> None:
    args = parse_args()
    if args.arithmetic != "f32":  # before anything launches (and inherited by the ranks of a re-launch): the C side reads the switches per call
        os.environ["GTSFM_ATTENTION_MATH"] = os.environ["GTSFM_GEMM_MATH"] = args.arithmetic
        args.no_secondary = True  # (the secondary legs are statements about the default arithmetic and switch the environment themselves)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(relaunch_one_process_per_gpu(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    plumbing = args.plumbing_only
    if plumbing:
        device = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X visible to PyTorch-ROCm (no CPU fallback)")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    dist = None
    # GTSFM_BENCH_FORCE_DIST=1 exercises the RCCL code path (init, weight broadcast, barrier, max-reduce) on one GPU
    if world > 1 or os.environ.get("GTSFM_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        import datetime

        # a rank that never arrives fails its peers after this long instead of blocking them for good (rank 0's roofline / CPU legs run after
        # the timed region and take minutes at N = 1 only; at N > 1 the other ranks wait for it in the closing barrier for well under this)
        timeout = datetime.timedelta(minutes=30)
        if plumbing:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timeout)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world, device_id=device, timeout=timeout)

    from gtsfm_amd import parallel

    h = args.height or args.size
    w = args.width or args.size
    scene = args.mode == "scene"
    lib = detector = matcher = gen = None
    if scene:
        # ONE scene sharded over the ranks: the product class does it (RCCL weight broadcast, cyclic detection, the feature exchange to
        # the ranks that match an image, 2-D cyclic pair ownership, the ragged gather of the match lists); this file only times it
        gen, _weights_dir = make_scene_generator(args, plumbing)
        pipe = gen.attach()
        if not plumbing:
            from gtsfm_amd.runtime import lib as L

            lib, detector, matcher = L.load(), gen._detector_descriptor._model, gen._matcher._model
    elif plumbing:
        from gtsfm_amd.utils.standin import StandInPipeline

        pipe = StandInPipeline(min(args.keypoints, 32))
    else:
        from gtsfm_amd.runtime import lib as L
        from gtsfm_amd.runtime import matcher_engine as ME
        from gtsfm_amd.runtime.pipeline import FrontEndPipeline
        from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine, pack_superpoint_weights

        lib = L.load()
        # weights: packed once on rank 0, broadcast over RCCL (xGMI)
        packed = None
        if rank == 0:
            packed = torch.from_numpy(pack_superpoint_weights(synthetic.synthetic_superpoint_state_dict())).to(device)
        detector = SuperPointEngine.from_packed(parallel.broadcast_packed_weights(packed, int(lib.gtsfm_sp_packed_weight_floats()), device))
        if args.matcher == "superglue":
            matcher = ME.SuperGlueEngine(synthetic.synthetic_superglue_state_dict(), device)
        elif args.matcher == "lightglue":
            matcher = ME.LightGlueEngine(synthetic.synthetic_lightglue_state_dict(), device)
        if matcher is not None and dist is not None:
            blob = matcher.weights if rank == 0 else None
            matcher.weights = parallel.broadcast_packed_weights(blob, matcher.weights.numel(), device)
        pipe = FrontEndPipeline(detector, matcher, max_keypoints=args.keypoints, pair_chunk=args.pair_chunk, num_streams=args.streams,
                                use_graphs=bool(args.graphs), share_first_layer=bool(args.share_first_layer))
    have_matcher = args.matcher != "none"
    mk = {"sinkhorn_iterations": args.sinkhorn} if (args.matcher == "superglue" and not plumbing) else {}

    n = args.images
    # overlapping views of one seeded canvas: exhaustive pairs share content (true correspondences) while every view is
    # detected independently. Replica mode: every rank has its own canvas; scene mode: one canvas for the job.
    if plumbing:
        views_np = np.random.default_rng(1000 + (0 if scene else rank)).integers(0, 256, (n, 8, 8)).astype(np.uint8)
    else:
        views_np = synthetic.synthetic_overlapping_views(n, h, w, 1000 + (0 if scene else rank))
    independent = args.pair_definition == "independent" and have_matcher
    my_images = list(range(n))
    plan = None
    if scene:
        all_pairs = parallel.exhaustive_pairs(n)[: args.pairs]
        plan = parallel.ScenePlan(n, all_pairs, rank, world)  # what the generator computes per call; here for the report
        my_images, my_pairs, pairs = plan.my_images, plan.my_pairs, plan.local_pairs
        images = torch.from_numpy(views_np[my_images]).to(device)  # this rank's views, resident before the timed region
        shapes = [(h, w)] * n
    else:
        images = torch.from_numpy(views_np).to(device)  # inputs resident in HBM before the timed region
        all_pairs = my_pairs = pairs = parallel.exhaustive_pairs(n)[: args.pairs] if have_matcher else []
        if independent:
            # 2 P image slots, every slot detected afresh each step (slot s shows view (5 s) % n; nothing is cached or shared
            # between slots), pair p = slots (2p, 2p + 1)
            slots = torch.arange(2 * args.pairs, device=device)
            images = images[(5 * slots) % n].contiguous()
            n = 2 * args.pairs
            all_pairs = my_pairs = pairs = [(2 * p, 2 * p + 1) for p in range(args.pairs)]
        shapes = [(h, w)] * n

    def step():
        if scene:  # detect own views -> exchange -> match own pairs -> gather: ShardedDetDescCorrespondenceGenerator.detect_and_run_scene
            out = gen.detect_and_run_scene(images, args.images, shapes, all_pairs, **mk)
            return out.table, out.results, out.matches
        feats = pipe.detect(images)
        res = pipe.match(feats, pairs, shapes, **mk) if have_matcher else []
        return feats, res, None

    def sync():
        if dist is not None:
            dist.barrier()
        if not plumbing:
            torch.cuda.synchronize(device)

    feats = res = gathered = None
    for _ in range(args.warmup):
        feats, res, gathered = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        feats, res, gathered = step()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    detect_only = not have_matcher
    if scene:
        units_per_step = len(all_pairs)
        assert gathered is not None and sorted(gathered) == sorted(all_pairs), "the gathered match lists do not cover the scene's pairs"
        assert len(gathered) == len(set(all_pairs)) == len(all_pairs)
    else:
        units_per_step = (n if detect_only else len(pairs)) * world
    value = units_per_step / (ms_per_step * 1e-3)

    if rank == 0:
        kcount = feats["count"].tolist()
        nmatch = int(sum(int((r["matches"] > -1).sum()) for r in res)) // 2 if res else 0
        layers, kept_frac = 18.0, 1.0
        if args.matcher == "lightglue" and not plumbing and res:
            layers = float(torch.cat([r["stop"] for r in res]).float().mean())
            kept_frac = float(torch.cat([r["kept"] for r in res]).float().mean()) / max(1, args.keypoints)
        n_img_total = args.images if scene else n * world
        flops_step = superpoint_flops(h, w) * n_img_total
        folded_step = 0.0
        if res:
            folded_step = folded_projection_flops(args.matcher, args.keypoints, layers) * units_per_step
            flops_step += matcher_flops(args.matcher, args.keypoints, layers, args.sinkhorn) * units_per_step
            shared_images = int(getattr(pipe, "last_shared_images", 0))
            if shared_images:  # executed work: the per-image block once per image of this rank's table, not twice per pair
                flops_step -= first_block_flops(args.matcher, args.keypoints) * (2 * len(pairs) - shared_images) * world  # ranks are balanced
                folded_step -= 2.0 * 256 * 256 * args.keypoints * (2 * len(pairs) - shared_images) * world  # that block's folded projection likewise
        if detect_only:
            workload = f"SuperPoint-only: {n} synthetic {h}x{w} gray images per GPU per step"
        elif scene:
            workload = (f"SuperPoint+{args.matcher}, ONE scene sharded over {world} GPU(s) by ShardedDetDescCorrespondenceGenerator: {len(all_pairs)} exhaustive "
                        f"(i<j) pairs of {args.images} synthetic {h}x{w} gray images per step (cyclic image ownership, RCCL all_to_all of every image to the ranks "
                        f"that match it, 2-D block-cyclic pair ownership, match lists gathered), top-{args.keypoints} keypoints per image")
        elif independent:
            workload = (f"SuperPoint+{args.matcher}: {len(pairs)} independent pairs = {n} fresh detections of synthetic {h}x{w} gray images per GPU "
                        f"per step, top-{args.keypoints} keypoints per image")
        else:
            workload = (f"SuperPoint+{args.matcher}: {len(pairs)} exhaustive (i<j) pairs of {n} synthetic {h}x{w} gray images per GPU per step "
                        f"(each image detected once per step), top-{args.keypoints} keypoints per image")
        if scene:
            par = (f"scene sharded over {world} rank(s) on a {'x'.join(map(str, parallel.process_grid(world)))} process grid; rank 0 matches "
                   f"{len(my_pairs)} pairs touching {len(parallel.images_touched(my_pairs))} of {args.images} images")
        else:
            par = f"dp{world}: independent image sets / pair lists per rank, RCCL weight broadcast, no data-path collective"
        result = {
            "metric": f"images/sec (SuperPoint detect+describe) @{h}x{w}" if detect_only else f"image-pairs/sec (detect+match) @{max(h, w)}px",
            "value": round(value, 2),
            "unit": "images/s" if detect_only else "image-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong" if scene else "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.arithmetic == "f32" else (f"f32 via {'3 x bf16' if args.arithmetic == 'bf16x3' else '2 x fp16'} split of the matchers' attention products and GEMMs, "
                                                              "f32 accumulate (opt-in; SuperPoint, sweeps: exact f32)"),
            "data": "synthetic",
            "config": {
                "workload": workload,
                "arithmetic": args.arithmetic,
                "mode": args.mode,
                "pair_definition": args.pair_definition if not detect_only else None,
                "images_per_gpu_per_step": len(my_images) if scene else n,
                "pairs_per_gpu_per_step": len(my_pairs),
                "keypoints_per_image": [int(min(kcount)), int(max(kcount))],
                "matcher_layers_run": layers,
                "sinkhorn_iterations": args.sinkhorn if args.matcher == "superglue" else None,
                "matches_per_pair": round(nmatch / max(1, len(pairs)), 1),
                "weights": "seeded synthetic (gtsfm_amd.utils.synthetic)",
                "parallelism": par,
                "pair_chunk": args.pair_chunk,
                "streams": args.streams,
                "hip_graphs": bool(args.graphs),
                "first_layer_block_per_image": (f"once per image per step ({int(getattr(pipe, 'last_shared_images', 0))} images) instead of twice per pair; "
                                                "bit-identical matches (tests/test_matchers_gpu.py)") if getattr(pipe, "last_shared_images", 0)
                                               else "per pair side (nothing shared)",
            },
            # SURVEY.md section 8(d)'s per-unit formula x the units of a step (the per-image first block counted once per image when shared):
            # the ALGORITHMIC rate, which is what a roofline compares; executed_tflops leaves out the products the folded projections remove
            "algorithmic_tflops": round(flops_step / (ms_per_step * 1e-3) / 1e12, 2),
            "executed_tflops": round((flops_step - folded_step) / (ms_per_step * 1e-3) / 1e12, 2),
        }
        if not detect_only and not plumbing:
            result["step_frac_of_fp32_mfma_peak"] = round(flops_step / (ms_per_step * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)
        if args.dump_matches:
            import hashlib

            per_pair = gathered if gathered is not None else (matches_to_numpy(res) if res else {})
            digest = hashlib.sha1()
            for pair in sorted(per_pair):
                digest.update(np.asarray(pair, dtype=np.int64).tobytes())
                digest.update(np.ascontiguousarray(per_pair[pair], dtype=np.int64).tobytes())
            result["match_digest"] = digest.hexdigest()
        if scene:  # asserted above: every pair of the scene came back from exactly one rank
            empty_ranks = sum(1 for r in range(world) if not parallel.partition_pairs_2d(all_pairs, r, world))
            result["scene_check"] = {"pairs_gathered": len(gathered), "pairs_of_the_scene": len(all_pairs), "each_pair_exactly_once": True,
                                     "ranks_without_pairs": empty_ranks}
            result["exchange"] = {"images_of_the_scene": args.images, "images_in_rank0_table": len(plan.table_images),
                                  "image_blocks_rank0_sends": plan.images_sent(),
                                  "largest_table_over_ranks": max(len(t) for t in plan.needed_by),
                                  "class": "gtsfm_amd.frontend.correspondence_generator.sharded_det_desc_correspondence_generator.ShardedDetDescCorrespondenceGenerator"}
        if dist is not None:
            result["distributed"] = {
                "backend": dist.get_backend(), "world_size": world,
                "collectives": ["broadcast (packed weights)", "barrier", "all_reduce MAX (step time)"]
                               + (["broadcast_object_list (matcher scalars)", "all_to_all_single (feature rows to the ranks that match them)",
                                   "all_gather_into_tensor (ragged match lists)"] if scene else []),
            }
        if kept_frac < 1.0:
            result["tflops_note"] = (f"upper bound: counted at full width N = {args.keypoints} in every layer; point pruning left "
                                     f"{kept_frac:.3f} of the keypoints alive at the final assignment")
        if plumbing:
            result["plumbing_only"] = True
            result["data"] = "NONE (plumbing only: no kernels ran, not a measurement)"
        else:
            chunk_pairs = min(args.pair_chunk, max(1, len(pairs)))
            if not args.no_roofline:
                conv_roof = measure_conv_roofline(lib, device, min(16, len(views_np)), h, w)
                if detect_only:
                    result["roofline"] = conv_roof
                else:  # dominant kernel of this workload first; the other kernels alongside
                    result["roofline"] = measure_attention_roofline(lib, device, args.keypoints, chunk_pairs, math={"f32": 0, "bf16x3": 1, "f16x2": 2}[args.arithmetic])
                    rows = 2 * chunk_pairs * (-(-args.keypoints // 128) * 128)  # LightGlue aligns every keypoint set to 128 rows
                    def guarded(fn, *fargs):  # a secondary kernel's micro-measurement must not cost the line its headline
                        try:
                            return fn(*fargs)
                        except Exception as exc:  # noqa: BLE001
                            torch.cuda.synchronize(device)
                            return {"kernel": fn.__name__, "error": f"{type(exc).__name__}: {str(exc)[:200]}"}

                    other = [guarded(measure_gemm_roofline, lib, device, rows, k, nn) for k, nn in ((256, 768), (512, 512), (512, 256), (256, 512))]
                    other.append(guarded(measure_score_gemm_roofline, lib, device, args.keypoints, chunk_pairs))
                    other.append(guarded(measure_sinkhorn_roofline, lib, device, args.keypoints, chunk_pairs))  # SuperGlue legs (headline or secondary)
                    if args.matcher == "lightglue" or not args.no_secondary:  # the worst kernel of the tree belongs in the line
                        lga = guarded(measure_lg_assignment_roofline, lib, device, args.keypoints, chunk_pairs)
                        other.extend(lga if isinstance(lga, list) else [lga])
                        other.append(guarded(measure_layernorm_roofline, lib, device, rows))
                    result["roofline_other"] = [r for r in other if r is not None] + [conv_roof]
            base = ora = None
            if world == 1 and not args.no_cpu_baseline:  # rank 0 at N = 1 only
                first = all_pairs[0] if all_pairs else (0, min(1, len(views_np) - 1))
                view_of = (lambda s: (5 * s) % args.images) if independent else (lambda s: s)  # noqa: E731
                sample_views = np.stack([views_np[view_of(first[0])], views_np[view_of(first[1])]])
                base, ora = cpu_baseline(sample_views, args.matcher, args.keypoints, args.sinkhorn)
                ref_run = reference_cpu_run(args.matcher, args.keypoints, args.sinkhorn)
                if ref_run is not None:
                    base["reference_run"] = ref_run
                result["cpu_baseline"] = base
                gpu_match = None
                if res:
                    a = res[0]["n0"][0]
                    gpu_match = (res[0]["matches"][:a].cpu().numpy(), res[0]["mscores"][:a].cpu().numpy())
                rows_ = [pairs[0][0], pairs[0][1]] if pairs else [0, min(1, n - 1)]
                result["parity_check"] = parity_check(ora, feats, rows_, gpu_match)
            if world == 1 and not args.no_secondary and not detect_only and not scene and args.pair_definition == "exhaustive":
                # every leg has its own timed region and must not be able to take the headline line down with it
                def leg(name, fn, *fargs):
                    try:
                        return fn(*fargs)
                    except Exception as exc:  # noqa: BLE001
                        torch.cuda.synchronize(device)
                        return {"error": f"{type(exc).__name__}: {str(exc)[:300]}", "leg": name}

                sec = leg("secondary_rates", secondary_rates, args, detector, matcher, device, h, w, mk, not args.no_cpu_baseline)
                result["secondary"] = sec if "error" not in sec else {"rates": sec}
                if args.matcher == "lightglue" and args.keypoints > 1024:
                    result["secondary"]["matcher_bf16x3"] = leg("matcher_bf16x3", attention_bf16x3_rate, args, lib, detector, matcher, images, pairs, shapes, device, ora, True)
                    result["secondary"]["attention_f16x2"] = leg("f16x2", attention_bf16x3_rate, args, lib, detector, matcher, images, pairs, shapes, device, ora, False, "f16x2")
                    result["secondary"]["matcher_f16x2"] = leg("matcher_f16x2", attention_bf16x3_rate, args, lib, detector, matcher, images, pairs, shapes, device, ora, True, "f16x2")
                if getattr(pipe, "last_shared_images", 0):
                    result["secondary"]["headline_per_pair_first_layer"] = leg("unshared", unshared_rate, args, detector, matcher, images, pairs, shapes, mk)
                if args.matcher == "lightglue":
                    result["secondary"]["lightglue_adaptive_depth"] = leg("adaptive", adaptive_depth_rate, args, detector, device, images, pairs, shapes)
                    if args.keypoints > 1024:
                        result["secondary"]["lightglue_adaptive_realistic"] = leg("adaptive_realistic", adaptive_realistic_rate, args, detector, device, h, w, not args.no_cpu_baseline)
                        result["secondary"]["lightglue_adaptive_realistic_f16x2"] = leg("adaptive_realistic_f16x2", _under_switches, "f16x2", adaptive_realistic_rate, args, detector, device, h, w,
                                                                                        not args.no_cpu_baseline)
                result["secondary"]["verifier_stage"] = leg("verifier", verifier_rate, pipe, feats, res, h, w, ms_per_step, device, not args.no_cpu_baseline)
                result["secondary"]["plugin_api"] = leg("plugin_api", plugin_api_rate, args, pipe, views_np, device, h, w)
                result["secondary"]["config2_superpoint_480x640"] = leg("config2", config2_superpoint_rate, lib, detector, device, not args.no_cpu_baseline)
                if args.keypoints > 2500 and (h, w) == (1024, 1024):
                    result["secondary"]["config4_scene_share_cap5000"] = leg("config4", config4_scene_share_rate, args, detector, device, h, w, not args.no_cpu_baseline)
                    result["secondary"]["config4_scene_share_cap5000_f16x2"] = leg("config4_f16x2", config4_scene_share_rate, args, detector, device, h, w,
                                                                                           not args.no_cpu_baseline, "f16x2")
        print(emit(result, args.details_file), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _verifier_kernel_ms(pipe, ver, feats, intr, thr, device):
    """HIP-event time of one gtsfm_verify_essential_f64 call on the step's compacted match lists (gather + RANSAC kernels)."""
    v = ver[0]
    k = feats["xy"].shape[1]
    pairs = v["pairs"]
    intr8 = np.concatenate([intr[[i for i, _ in pairs]], intr[[j for _, j in pairs]]], axis=1)
    args = (feats["xy"].reshape(-1, 2), [i * k for i, _ in pairs], [j * k for _, j in pairs], v["match_idx"], v["match_off"], intr8, thr)
    kw = dict(seeds=[(i << 32) | j for i, j in pairs], match_count=v["match_count"])
    # verify_batch uploads its small index arrays first; the events bracket uploads + both kernels on the current stream
    pipe._verifier.verify_batch(*args, **kw)
    torch.cuda.synchronize(device)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(3):
        pipe._verifier.verify_batch(*args, **kw)
    end.record()
    torch.cuda.synchronize(device)
    return round(start.elapsed_time(end) / 3, 3)


def verifier_rate(pipe, feats, res, h, w, ms_per_step, device, with_oracle: bool):
    """The stage behind the path: five-point RANSAC + pose recovery over the match lists the
    timed step just produced, device to device. Its own timed region; the headline value does not include it. With the
    oracle leg: the first pair re-verified by oracle/verifier_oracle.py on the host (checker and CPU figure)."""
    intr = np.tile(np.array([[1.2 * max(h, w), 1.2 * max(h, w), w / 2.0, h / 2.0]]), (int(feats["xy"].shape[0]), 1))
    thr = 4.0  # estimation_threshold_px of gtsfm/configs/deep_front_end.yaml:49
    ver = pipe.verify(feats, res, intr, thr)
    torch.cuda.synchronize(device)
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        ver = pipe.verify(feats, res, intr, thr)
    torch.cuda.synchronize(device)
    ms = (time.perf_counter() - t0) / steps * 1e3
    kernel_ms = _verifier_kernel_ms(pipe, ver, feats, intr, thr, device)
    npairs = sum(len(v["pairs"]) for v in ver)
    stats = torch.cat([v["stats"] for v in ver]).cpu().numpy()
    counts = torch.cat([v["match_count"] for v in ver]).cpu().numpy()
    out = {
        "value": round(npairs / (ms * 1e-3), 1), "unit": "image-pairs/s", "ms_per_step": round(ms, 3), "steps": steps, "pairs_per_step": npairs,
        "share_of_detect_match_step": round(ms / ms_per_step, 4), "dtype": "f64", "threshold_px": thr,
        "ransac_kernel_ms": kernel_ms,
        "putative_per_pair": round(float(counts.mean()), 1), "verified_per_pair": round(float(stats[:, 0].mean()), 1),
        "hypotheses_per_pair": round(float(stats[:, 1].mean()), 1), "pairs_with_model": int((stats[:, 0] > 0).sum()),
        "workload": "gtsfm_verify_compact_matches + gtsfm_verify_essential_f64 on the match lists of the timed step (5-point MSAC RANSAC, "
                    "<= 1024 hypotheses per pair + 256 drawn from the winner's inliers, cheirality pose choice); PARITY UNPINNED towards OpenCV's USAC",
    }
    if with_oracle:
        from gtsfm_amd.runtime.pipeline import FrontEndPipeline
        from oracle import verifier_oracle as vo

        got = FrontEndPipeline.verified_to_numpy(ver[:1])
        (i, j) = ver[0]["pairs"][0]
        xy, cnt = feats["xy"].cpu().numpy(), feats["count"].cpu().numpy()
        t0 = time.perf_counter()
        ref = vo.verify(xy[i, : cnt[i]], xy[j, : cnt[j]], got[(i, j)]["putative"], tuple(intr[i]), tuple(intr[j]), thr, seed=(i << 32) | j)
        sec = time.perf_counter() - t0
        same = bool(np.array_equal(ref["v_corr_idxs"], got[(i, j)]["v_corr_idxs"]) and ref["hypotheses"] == got[(i, j)]["hypotheses"])
        if ref["R"] is not None and got[(i, j)]["R"] is not None:
            same = same and bool(np.abs(ref["R"] - got[(i, j)]["R"]).max() < 1e-9 and np.abs(ref["t"] - got[(i, j)]["t"]).max() < 1e-9)
        out["parity_check"] = {"verified_equal_oracle": same, "pair": [int(i), int(j)], "verified": int(len(ref["v_corr_idxs"]))}
        out["cpu_baseline"] = {"value": round(1.0 / sec, 2), "unit": "image-pairs/s", "cores": 1, "kind": "port", "sample": "the first pair, numpy float64 oracle"}
    return out


SIDE_LEG_PAIRS = 250  # pairs of the side legs that re-run the headline's pair list under another setting (its first 250 pairs = 23 views)


def unshared_rate(args, detector, matcher, images, pairs, shapes, mk):
    """The headline workload with every pair running the matcher's full forward (the per-image first block NOT shared between
    the pairs of an image: what the reference's per-pair match() calls amount to), own timed region, for comparison."""
    from gtsfm_amd.runtime.pipeline import FrontEndPipeline

    pipe = FrontEndPipeline(detector, matcher, max_keypoints=args.keypoints, pair_chunk=args.pair_chunk, num_streams=args.streams,
                            use_graphs=bool(args.graphs), share_first_layer=False)
    pairs = pairs[:SIDE_LEG_PAIRS]
    images = images[: max(max(p) for p in pairs) + 1]
    _, timing = _time_steps(lambda: pipe.match(pipe.detect(images), pairs, shapes, **mk), SECONDARY_STEPS, 1, images.device)
    return {"value": round(len(pairs) / (timing["ms_per_step"] * 1e-3), 2), "unit": "image-pairs/s", **timing,
            "pairs_per_step": len(pairs), "images_per_step": int(images.shape[0]),
            "workload": f"the first {len(pairs)} pairs of the headline workload with --share-first-layer 0 (first matcher block once per pair side)"}


def adaptive_depth_rate(args, detector, device, images, pairs, shapes):
    """The headline workload with token-confidence heads that DO fire (the headline's seeded heads never reach the exit
    threshold, so it pays all 9 layers for every pair -- the worst case). LightGlue's adaptive depth and width run on the device
    (``lg_stop_check`` / ``lg_prune_*``): pairs leave the batch at different layers inside one launch sequence. Not the headline:
    a different weight set (``conf_bias`` 1, ``conf_gain`` 4: same recipe as the early-stop parity tests; the synthetic views are
    alike, so every pair leaves at the same layer -- real image pairs spread over layers 3 to 9), reported with the layers it ran."""
    from gtsfm_amd.runtime import matcher_engine as ME
    from gtsfm_amd.runtime.pipeline import FrontEndPipeline

    matcher = ME.LightGlueEngine(synthetic.synthetic_lightglue_state_dict(conf_bias=1.0, conf_gain=4.0), device)
    pipe = FrontEndPipeline(detector, matcher, max_keypoints=args.keypoints, pair_chunk=args.pair_chunk, num_streams=args.streams,
                            use_graphs=bool(args.graphs), share_first_layer=bool(args.share_first_layer))
    pairs = pairs[:SIDE_LEG_PAIRS]
    images = images[: max(max(p) for p in pairs) + 1]
    res, timing = _time_steps(lambda: pipe.match(pipe.detect(images), pairs, shapes), SECONDARY_STEPS, 1, device)
    ms = timing["ms_per_step"]
    layers = torch.cat([r["stop"] for r in res]).float()
    kept = torch.cat([r["kept"] for r in res]).float()
    nm = int(sum(int((r["matches"] > -1).sum()) for r in res)) // 2
    return {"value": round(len(pairs) / (ms * 1e-3), 2), "unit": "image-pairs/s", **timing,
            "pairs_per_step": len(pairs), "matcher_layers_run": {"mean": round(float(layers.mean()), 2), "min": int(layers.min()), "max": int(layers.max())},
            "keypoints_alive_at_assignment": round(float(kept.mean()), 1), "matches_per_pair": round(nm / max(1, len(pairs)), 1),
            "workload": f"the first {len(pairs)} pairs of the headline workload with synthetic token-confidence heads that fire (conf_bias 1, conf_gain 4): adaptive depth / width on the device"}


ADAPTIVE_HEADS = dict(delta_gain=0.25, conf_bias=4.5, conf_gain=30.0, conf_ramp=0.6, conf_shared_direction=True, match_bias=-3.5, match_gain=40.0)


_ADAPTIVE_ORACLE_CACHE: dict = {}


def adaptive_realistic_rate(args, detector, device, h, w, with_oracle: bool):
    """LightGlue as the reference configures it -- `LightGlue(features=...)` with upstream's adaptive depth (0.95) and width (0.99) defaults,
    gtsfm/frontend/matcher/lightglue_matcher.py:41 -- on a workload where BOTH mechanisms fire and pairs differ: 250 exhaustive pairs of 23 views cut
    from four canvases of different texture scale (synthetic.synthetic_mixed_scene: overlaps of 100 % .. ~30 % inside a canvas, unrelated pairs across
    canvases, like a real exhaustive visibility graph) with token-confidence heads whose confidence grows with depth at an image-dependent level and
    matchability heads that declare a share of the keypoints unmatchable (ADAPTIVE_HEADS; chosen with tools/tune_adaptive_leg.py). Pairs leave the
    batch at different layers inside one launch sequence and both images of a pair are pruned to different widths: the ragged device-side control flow
    production hits. Reported: the rate, the distribution of layers run and of keypoints alive at the assignment, and a parity check of five pairs
    (one per distinct stop layer where there are that many) against the oracle on the GPU's own features: stop layer, match indices, scores.
    Synthetic heads: the SHAPE of the distributions is a construction, not a prediction for real checkpoints."""
    from gtsfm_amd import parallel
    from gtsfm_amd.runtime import matcher_engine as ME
    from gtsfm_amd.runtime.pipeline import FrontEndPipeline

    sd = synthetic.synthetic_lightglue_state_dict(**ADAPTIVE_HEADS)
    matcher = ME.LightGlueEngine(sd, device)
    pipe = FrontEndPipeline(detector, matcher, max_keypoints=args.keypoints, pair_chunk=args.pair_chunk, num_streams=args.streams,
                            use_graphs=bool(args.graphs), share_first_layer=bool(args.share_first_layer))
    n = fewest_images_for(SIDE_LEG_PAIRS)
    pairs = parallel.exhaustive_pairs(n)[:SIDE_LEG_PAIRS]
    images = torch.from_numpy(synthetic.synthetic_mixed_scene(n, h, w, canvases=4)).to(device)
    shapes = [(h, w)] * n
    state = {}

    def step():
        state["feats"] = pipe.detect(images)
        return pipe.match(state["feats"], pairs, shapes)

    res, timing = _time_steps(step, SECONDARY_STEPS, 1, device)
    ms = timing["ms_per_step"]
    stop = torch.cat([r["stop"] for r in res]).cpu().numpy().astype(np.int64)
    kept = torch.cat([r["kept"] for r in res]).cpu().numpy().reshape(-1).astype(np.float64)
    nm = np.array([len(v) for v in FrontEndPipeline.matches_to_numpy(res).values()])
    k = float(args.keypoints)
    out = {"value": round(len(pairs) / (ms * 1e-3), 2), "unit": "image-pairs/s", **timing, "pairs_per_step": len(pairs), "images_per_step": n,
           "matcher_layers_run": {"mean": round(float(stop.mean()), 2), "histogram_layers_1_to_9": np.bincount(stop, minlength=10)[1:10].tolist()},
           "keypoints_alive_at_assignment_share": {q: round(float(np.quantile(kept, v)) / k, 3) for q, v in (("min", 0.0), ("p25", 0.25), ("median", 0.5), ("p75", 0.75), ("max", 1.0))},
           "matches_per_pair": {"mean": round(float(nm.mean()), 1), "max": int(nm.max())},
           "heads": ADAPTIVE_HEADS,
           "workload": f"{len(pairs)} exhaustive pairs of {n} synthetic {h}x{w} views from 4 canvases of different texture scale (synthetic_mixed_scene), top-{args.keypoints} "
                       "keypoints, LightGlue with upstream's adaptive depth / width defaults and synthetic heads that fire at pair-dependent depths"}
    if with_oracle:
        from oracle import lightglue_oracle

        feats = state["feats"]
        cnt = feats["count"].cpu().numpy()
        flat = [(q, r) for r in res for q in range(len(r["pairs"]))]
        picked, seen = [], set()
        for idx, (q, r) in enumerate(flat):  # one pair per distinct stop layer, cheapest (earliest) layers first, five at most
            layer = int(stop[idx])
            if layer not in seen:
                seen.add(layer)
                picked.append((layer, idx, q, r))
        picked = sorted(picked)[:5]
        T = torch.from_numpy
        checks, t0 = [], time.perf_counter()
        for layer, idx, q, r in picked:
            i, j = r["pairs"][q]
            row = sum(a + b for a, b in zip(r["n0"][:q], r["n1"][:q]))
            a = r["n0"][q]
            got_m = r["matches"][row : row + a].cpu().numpy().astype(np.int64)
            got_s = r["mscores"][row : row + a].cpu().numpy()
            kp = [feats["xy"][v, : cnt[v]].cpu().numpy() for v in (i, j)]
            de = [feats["descriptors"][v, : cnt[v]].cpu().numpy() for v in (i, j)]
            key = (int(i), int(j), h, w, args.keypoints)  # (the leg runs again under the opt-in arithmetic on the same exact-fp32 features: the oracle's answer is the same)
            if key not in _ADAPTIVE_ORACLE_CACHE:
                with torch.no_grad():
                    _ADAPTIVE_ORACLE_CACHE[key] = lightglue_oracle.lightglue_forward(sd, T(kp[0])[None], T(kp[1])[None], T(de[0])[None], T(de[1])[None], (h, w), (h, w))
            ref = _ADAPTIVE_ORACLE_CACHE[key]
            ref_m = ref["matches0"][0].numpy().astype(np.int64)
            equal = bool(np.array_equal(got_m, ref_m))
            checks.append({"pair": [int(i), int(j)], "layers_run": layer, "oracle_layers_run": int(ref["stop"]), "matches": int((ref_m > -1).sum()), "matches_equal": equal,
                           "max_dscore": float(np.abs(got_s - ref["matching_scores0"][0].numpy()).max()) if equal else None})
        out["parity_check"] = {"pairs": checks, "oracle_s": round(time.perf_counter() - t0, 1),
                               "within_tolerance": bool(all(c["matches_equal"] and c["layers_run"] == c["oracle_layers_run"] and c["max_dscore"] < 1e-4 for c in checks))}
    return out


SECONDARY_STEPS = 3  # timed steps of every secondary leg (each step bracketed by a device synchronisation; the MEDIAN is reported, min / max beside it)


def _time_steps(step, steps: int, warmup: int, device):
    """Run `step` warmup + steps times; every timed step is bracketed by torch.cuda.synchronize (a secondary leg is its own timed
    region, and a step lasts 0.1 - 9 s, so the synchronisation costs nothing measurable). Returns (last result, timing dict): the
    median step time decides the rate; min / max show the spread, so that a round-to-round change can be told from noise."""
    out = None
    for _ in range(warmup):
        out = step()
    times = []
    for _ in range(steps):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        out = step()
        torch.cuda.synchronize(device)
        times.append((time.perf_counter() - t0) * 1e3)
    med = float(np.median(times))
    return out, {"ms_per_step": round(med, 3), "ms_per_step_min_max": [round(min(times), 3), round(max(times), 3)], "steps": steps, "warmup": warmup,
                 "statistic": "median over the timed steps"}


def _timed_pipeline(pipe, images, pairs, shapes, steps, warmup, device, mk):
    def step():
        feats = pipe.detect(images)
        return feats, pipe.match(feats, pairs, shapes, **mk)

    (feats, res), timing = _time_steps(step, steps, warmup, device)
    ms = timing["ms_per_step"]
    kc = feats["count"].tolist()
    nm = int(sum(int((r["matches"] > -1).sum()) for r in res)) // 2
    out = {"value": round(len(pairs) / (ms * 1e-3), 2), "unit": "image-pairs/s", **timing,
           "pairs_per_step": len(pairs), "images_per_step": int(images.shape[0]), "keypoints_per_image": [int(min(kc)), int(max(kc))],
           "matches_per_pair": round(nm / max(1, len(pairs)), 1)}
    return out, feats, res


def secondary_rates(args, detector, matcher, device, h, w, mk, with_oracle: bool):
    """Rates next to the headline, each with its own timed region (N = 1), all on the driver-visible line:
    the N = 2048 exhaustive rate SURVEY.md section 8(d) asks to report additionally (round 2's headline); SuperGlue with 20
    (what GTSfM runs) and 100 (BASELINE config 4) Sinkhorn iterations at the headline's keypoint cap, and SuperGlue/100
    at N = 2048 with its own parity check; independent pairs (two fresh detections per pair) at the headline's cap."""
    from gtsfm_amd import parallel
    from gtsfm_amd.runtime import matcher_engine as ME
    from gtsfm_amd.runtime.pipeline import FrontEndPipeline

    out = {}
    nstreams, graphs, share = args.streams, bool(args.graphs), bool(args.share_first_layer)

    def exhaustive(mt, keypoints, npairs, seed, steps, mkw):
        n = fewest_images_for(npairs)
        views_np = synthetic.synthetic_overlapping_views(n, h, w, seed)
        views = torch.from_numpy(views_np).to(device)
        pairs = parallel.exhaustive_pairs(n)[:npairs]
        pipe = FrontEndPipeline(detector, mt, max_keypoints=keypoints, pair_chunk=default_pair_chunk(keypoints), num_streams=nstreams, use_graphs=graphs,
                                share_first_layer=share)
        r, feats, res = _timed_pipeline(pipe, views, pairs, [(h, w)] * n, steps, 1, device, mkw)
        return r, feats, res, views_np, pairs

    # (1) the other keypoint counts SURVEY.md section 8(d) names: fixed N = 2048 and 1024 next to the cap of 5000 (the cap when the headline
    # was asked for another count)
    for other_k, other_pairs in (((2048, 1000), (1024, 1000)) if args.keypoints > 2500 else ((5000, 200),)):
        r, *_ = exhaustive(matcher, other_k, other_pairs, 2000, SECONDARY_STEPS, mk)
        note = {2048: " (SURVEY.md section 8d: additionally reported; round 2's headline)", 1024: " (SURVEY.md section 8d: additionally reported)", 5000: " (GTSfM's default cap)"}[other_k]
        out[f"exhaustive_top{other_k}"] = dict(r, workload=f"SuperPoint+{args.matcher}: {other_pairs} exhaustive pairs of {r['images_per_step']} synthetic {h}x{w} views, "
                                                            f"top-{other_k} keypoints per image" + note)
    # (2) SuperGlue (BASELINE config 4 per GPU; GTSfM itself runs 20 iterations)
    if args.matcher == "lightglue":
        sg = ME.SuperGlueEngine(synthetic.synthetic_superglue_state_dict(), device)
        for iters in (20, 100):
            r, *_ = exhaustive(sg, args.keypoints, 100, 3000, SECONDARY_STEPS, {"sinkhorn_iterations": iters})
            out[f"superglue_sinkhorn{iters}"] = dict(r, sinkhorn_iterations=iters, workload=(
                f"SuperPoint+superglue, {iters} Sinkhorn iterations: 100 exhaustive pairs of {r['images_per_step']} synthetic {h}x{w} views, top-{args.keypoints} keypoints per image"))
        if args.keypoints != 2048:
            r, feats, res, views_np, pairs = exhaustive(sg, 2048, 500, 3000, SECONDARY_STEPS, {"sinkhorn_iterations": 100})
            entry = dict(r, sinkhorn_iterations=100, workload=f"SuperPoint+superglue, 100 Sinkhorn iterations: 500 exhaustive pairs of {r['images_per_step']} synthetic "
                                                              f"{h}x{w} views, top-2048 keypoints per image (round 2's BASELINE-config-4 figure)")
            if with_oracle:  # its own parity check: the oracle on the first pair of this leg
                i, j = pairs[0]
                _, ora = cpu_baseline(np.stack([views_np[i], views_np[j]]), "superglue", 2048, 100)
                a = res[0]["n0"][0]
                entry["parity_check"] = parity_check(ora, feats, [i, j], (res[0]["matches"][:a].cpu().numpy(), res[0]["mscores"][:a].cpu().numpy()))
            out["superglue_sinkhorn100_top2048"] = entry
        del sg
    # (3) independent pairs: 2 fresh detections per pair, nothing shared between pairs
    p = 100 if args.keypoints > 2500 else 500
    base = torch.from_numpy(synthetic.synthetic_overlapping_views(46, h, w, 1000)).to(device)
    images = base[(5 * torch.arange(2 * p, device=device)) % 46].contiguous()
    pairs = [(2 * q, 2 * q + 1) for q in range(p)]
    pipe = FrontEndPipeline(detector, matcher, max_keypoints=args.keypoints, pair_chunk=args.pair_chunk, num_streams=nstreams, use_graphs=graphs, share_first_layer=share)
    r, *_ = _timed_pipeline(pipe, images, pairs, [(h, w)] * (2 * p), SECONDARY_STEPS, 1, device, mk)
    out["independent_pairs"] = dict(r, workload=f"SuperPoint+{args.matcher}: {p} independent pairs = {2 * p} fresh detections of synthetic {h}x{w} views, top-{args.keypoints} keypoints per image")
    return out


def config2_superpoint_rate(lib, detector, device, with_oracle: bool):
    """BASELINE config 2 exactly: SuperPoint only, 256 synthetic 640x480 gray images on one GPU, keypoints against the CPU oracle.
    Own timed region; carries the conv stack's roofline at this shape and its own parity check (image 0 through the oracle)."""
    from gtsfm_amd.runtime.pipeline import FrontEndPipeline

    h, w, n, cap = 480, 640, 256, 5000
    views_np = synthetic.synthetic_overlapping_views(n, h, w, 4000)
    images = torch.from_numpy(views_np).to(device)
    pipe = FrontEndPipeline(detector, None, max_keypoints=cap)
    feats, timing = _time_steps(lambda: pipe.detect(images), SECONDARY_STEPS, 1, device)
    ms = timing["ms_per_step"]
    kc = np.asarray(feats["count"].tolist())
    out = {"value": round(n / (ms * 1e-3), 1), "unit": "images/s", **timing, "images_per_step": n, "dtype": "f32",
           "keypoints_per_image": {"min": int(kc.min()), "median": int(np.median(kc)), "max": int(kc.max())},
           "algorithmic_tflops": round(superpoint_flops(h, w) * n / (ms * 1e-3) / 1e12, 2),
           "workload": f"BASELINE config 2: SuperPoint detect+describe over {n} synthetic {w}x{h} gray uint8 images resident in HBM, batches of 16, top-{cap} keypoints kept on the device",
           "roofline": measure_conv_roofline(lib, device, 16, h, w)}
    if with_oracle:
        base, ora = cpu_baseline(views_np[:1], "none", cap, 0)
        out["cpu_baseline"] = base
        out["parity_check"] = parity_check(ora, feats, [0], None)
    return out


def _under_switches(math: str, fn, *fargs):
    """`fn(*fargs)` with both opt-in arithmetic switches set to `math` (read per call / launch by the C side; graphs are captured under them)."""
    switches = ("GTSFM_ATTENTION_MATH", "GTSFM_GEMM_MATH")
    old = {k: os.environ.get(k) for k in switches}
    for k in switches:
        os.environ[k] = math
    try:
        out = fn(*fargs)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    pieces = "3 x bf16" if math == "bf16x3" else "2 x fp16"
    out["dtype"] = f"f32 via {pieces} split of the attention products and the projection / score GEMMs, f32 accumulate (SuperPoint, Sinkhorn, extraction: exact f32)"
    out["workload"] = out.get("workload", "") + f"; GTSFM_ATTENTION_MATH={math} and GTSFM_GEMM_MATH={math} (opt-in)"
    return out


def config4_scene_share_rate(args, detector, device, h, w, with_oracle: bool, math: str = "f32"):
    """BASELINE config 4 on the one GPU bench.py is given at N = 1: the HEAVIEST rank's share of the 8-rank job -- 101 views, the
    first 5000 exhaustive pairs, SuperGlue with 100 Sinkhorn iterations, AT THE 5000-KEYPOINT CAP -- exactly as ``--mode scene --gpus 8``
    assigns it (gtsfm_amd.parallel: cyclic image ownership, 2-D cyclic pair ownership on the 2 x 4 process grid). A timed step = this
    rank's detections + its pairs matched from the scene's feature table; the rows of the table that the other seven ranks would
    deliver through the all-gather are detected here beforehand, outside the timed region. With balanced ranks (the heaviest holds 3 %
    more pairs than the mean) the 8-GPU job takes this step's time plus the all-gather (523 MB in total, MB-scale per xGMI link)."""
    from gtsfm_amd import parallel
    from gtsfm_amd.runtime import matcher_engine as ME
    from gtsfm_amd.runtime.pipeline import FrontEndPipeline

    if math != "f32":  # the same leg under the opt-in arithmetic of attention and GEMMs
        return _under_switches(math, config4_scene_share_rate, args, detector, device, h, w, with_oracle, "f32")
    n, world, scene_pairs, iters = 101, 8, 5000, 100
    all_pairs = parallel.exhaustive_pairs(n)[:scene_pairs]
    shares = [parallel.partition_pairs_2d(all_pairs, r, world) for r in range(world)]
    rank = int(np.argmax([len(p) for p in shares]))
    my_pairs, my_images = shares[rank], parallel.partition_images(n, rank, world)
    touched = sorted(parallel.images_touched(my_pairs))
    views_np = synthetic.synthetic_overlapping_views(n, h, w, 1000)
    sg = ME.SuperGlueEngine(synthetic.synthetic_superglue_state_dict(), device)
    pipe = FrontEndPipeline(detector, sg, max_keypoints=args.keypoints, pair_chunk=default_pair_chunk(args.keypoints), num_streams=args.streams,
                            use_graphs=bool(args.graphs), share_first_layer=bool(args.share_first_layer))
    k = args.keypoints
    table = {"count": torch.zeros(n, dtype=torch.int32, device=device), "xy": torch.zeros((n, k, 2), device=device),
             "scores": torch.zeros((n, k), device=device), "descriptors": torch.zeros((n, k, 256), device=device)}
    others = [i for i in touched if i not in set(my_images)]
    for c0 in range(0, len(others), 16):  # the all-gather's stand-in: the other ranks' rows of the table, untimed
        idx = others[c0 : c0 + 16]
        f = pipe.detect(torch.from_numpy(views_np[idx]).to(device))
        ii = torch.tensor(idx, dtype=torch.long, device=device)
        for key in table:
            table[key][ii] = f[key]
    own = torch.from_numpy(views_np[my_images]).to(device)
    own_idx = torch.tensor(my_images, dtype=torch.long, device=device)
    shapes = [(h, w)] * n

    def step():
        f = pipe.detect(own)
        for key in table:
            table[key][own_idx] = f[key]
        return pipe.match(table, my_pairs, shapes, sinkhorn_iterations=iters)

    res, timing = _time_steps(step, SECONDARY_STEPS, 1, device)
    ms = timing["ms_per_step"]
    nm = int(sum(int((r["matches"] > -1).sum()) for r in res)) // 2
    kc = table["count"][torch.tensor(touched, device=device)].tolist()
    out = {"value": round(len(my_pairs) / (ms * 1e-3), 2), "unit": "image-pairs/s", **timing, "dtype": "f32",
           "pairs_per_step": len(my_pairs), "images_detected_per_step": len(my_images), "images_touched": len(touched), "rank": rank,
           "pairs_by_rank": [len(p) for p in shares], "sinkhorn_iterations": iters, "keypoints_per_image": [int(min(kc)), int(max(kc))],
           "matches_per_pair": round(nm / max(1, len(my_pairs)), 1),
           "scene_pairs_per_s_if_8_ranks_take_this_long": round(scene_pairs / (ms * 1e-3), 1),
           "workload": (f"BASELINE config 4, one GPU's share: rank {rank} (the heaviest) of 8 on the 2x4 process grid of `--mode scene`: {len(my_images)} of {n} synthetic {h}x{w} views "
                        f"detected + {len(my_pairs)} of the scene's {scene_pairs} exhaustive pairs matched per step, SuperPoint+superglue, {iters} Sinkhorn iterations, top-{k} keypoints "
                        "per image; the other ranks' feature rows are resident before the timed region (the all-gather's stand-in)")}
    if with_oracle:  # its own parity check: the oracle on the first pair of this share (~20 s of CPU at the cap)
        i, j = res[0]["pairs"][0]
        base, ora = cpu_baseline(np.stack([views_np[i], views_np[j]]), "superglue", k, iters)
        a = res[0]["n0"][0]
        out["cpu_baseline"] = base
        out["parity_check"] = parity_check(ora, table, [i, j], (res[0]["matches"][:a].cpu().numpy(), res[0]["mscores"][:a].cpu().numpy()))
    return out


def attention_bf16x3_rate(args, lib, detector, matcher, images, pairs, shapes, device, oracle_out, gemm_too: bool = False, mode: str = "bf16x3"):
    """The opt-in arithmetics GTSFM_ATTENTION_MATH=bf16x3 | f16x2 on the first pairs of the headline workload: both products of every attention
    launch on v_mfma_f32_32x32x16_bf16 with each fp32 operand split exactly into three bf16 pieces (six of the nine piece products), or on
    v_mfma_f32_32x32x16_f16 with two fp16 pieces (three products); fp32 accumulation: fp32-class error per product, NOT the exact-fp32 kernel's
    bits; SuperPoint, the sweeps and -- unless gemm_too -- the GEMMs stay exact fp32. Own timed region; compared pair by pair with the exact-fp32
    pipeline on the same input and, for the first pair, with the oracle."""
    from gtsfm_amd.runtime.pipeline import FrontEndPipeline

    pairs = pairs[:SIDE_LEG_PAIRS]
    images = images[: max(max(p) for p in pairs) + 1]

    def make_pipe():
        return FrontEndPipeline(detector, matcher, max_keypoints=args.keypoints, pair_chunk=args.pair_chunk, num_streams=args.streams,
                                use_graphs=bool(args.graphs), share_first_layer=bool(args.share_first_layer))

    feats = make_pipe().detect(images)
    exact = make_pipe().match(feats, pairs, shapes)
    torch.cuda.synchronize(device)
    switches = ["GTSFM_ATTENTION_MATH"] + (["GTSFM_GEMM_MATH"] if gemm_too else [])
    old = {k: os.environ.get(k) for k in switches}
    for k in switches:
        os.environ[k] = mode  # read per call / per launch by the C entry points; graphs are captured under them
    try:
        pipe = make_pipe()
        res, timing = _time_steps(lambda: pipe.match(pipe.detect(images), pairs, shapes), SECONDARY_STEPS, 1, device)
        roof = measure_attention_roofline(lib, device, args.keypoints, min(args.pair_chunk, len(pairs)), math=1 if mode == "bf16x3" else 2)
        if gemm_too:
            rows = 2 * min(args.pair_chunk, len(pairs)) * (-(-args.keypoints // 128) * 128)
            roof = {"attention": roof, "gemm": [measure_gemm_roofline(lib, device, rows, k, nn) for k, nn in ((256, 768), (512, 512), (512, 256))]}
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    ms = timing["ms_per_step"]
    same_pairs, dmax, nmatch = 0, 0.0, 0
    for a, b in zip(exact, res):
        n0 = a["n0"]
        ma, mb = a["matches"].cpu().numpy(), b["matches"].cpu().numpy()
        sa, sb = a["mscores"].cpu().numpy(), b["mscores"].cpu().numpy()
        off = 0
        for q in range(len(a["pairs"])):
            t = int(n0[q]) + int(a["n1"][q])
            eq = bool(np.array_equal(ma[off : off + t], mb[off : off + t]))
            same_pairs += int(eq)
            if eq:
                dmax = max(dmax, float(np.abs(sa[off : off + t] - sb[off : off + t]).max()))
            nmatch += int((ma[off : off + int(n0[q])] > -1).sum())
            off += t
    out = {"value": round(len(pairs) / (ms * 1e-3), 2), "unit": "image-pairs/s", **timing, "pairs_per_step": len(pairs), "images_per_step": int(images.shape[0]),
           "dtype": ((f"f32 via {'3 x bf16' if mode == 'bf16x3' else '2 x fp16'} split of the attention products AND the matcher's projection / score GEMMs, f32 accumulate "
                      "(SuperPoint, sweeps: exact f32)") if gemm_too
                     else f"f32 via {'3 x bf16' if mode == 'bf16x3' else '2 x fp16'} split of both attention products, f32 accumulate (SuperPoint, GEMMs, sweeps: exact f32)"),
           "matcher_layers_run": float(torch.cat([r["stop"] for r in res]).float().mean()) if res and "stop" in res[0] else None,
           "against_exact_fp32_pipeline": {"pairs_with_identical_match_arrays": same_pairs, "pairs": len(pairs), "max_dscore_on_those": dmax, "matches": nmatch},
           "roofline": roof,
           "workload": f"the first {len(pairs)} pairs of the headline workload with {' and '.join(k + '=' + mode for k in switches)} (opt-in; the headline stays exact fp32)"}
    if oracle_out is not None and res:
        a = res[0]["n0"][0]
        out["parity_check"] = parity_check(oracle_out, feats, [pairs[0][0], pairs[0][1]], (res[0]["matches"][:a].cpu().numpy(), res[0]["mscores"][:a].cpu().numpy()))
    return out


def plugin_api_rate(args, pipe, views_np, device, h, w):
    """The path GTSfM's Dask graph calls UNCHANGED (gtsfm/frontend/correspondence_generator/det_desc_correspondence_generator.py:57-81):
    one ``detect_and_describe(image)`` per image and one ``match(...)`` per pair through the plugin classes, numpy in / numpy out, one
    synchronous call at a time (one worker thread). Host buffers cross PCIe on every call, so this is NOT ``value``; it is the rate a
    user of the drop-in plugins sees without the batched correspondence generator. The first pair is compared with the batched pipeline."""
    import tempfile

    from gtsfm_amd.common.image import Image
    from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor
    from gtsfm_amd.frontend.matcher.lightglue_matcher import LightGlueMatcher
    from gtsfm_amd.frontend.matcher.superglue_matcher import SuperGlueMatcher
    from gtsfm_amd import parallel

    n_img = min(6, len(views_np))
    pairs = parallel.exhaustive_pairs(n_img)[:12]
    with tempfile.TemporaryDirectory() as tmp:
        torch.save(synthetic.synthetic_superpoint_state_dict(), f"{tmp}/sp.pth")
        det = SuperPointDetectorDescriptor(max_keypoints=args.keypoints, weights_path=f"{tmp}/sp.pth")
        if args.matcher == "superglue":
            torch.save(synthetic.synthetic_superglue_state_dict(), f"{tmp}/sg.pth")
            mt = SuperGlueMatcher(weights_path=f"{tmp}/sg.pth")
            mt._config["sinkhorn_iterations"] = args.sinkhorn
        else:
            torch.save(synthetic.synthetic_lightglue_state_dict(), f"{tmp}/lg.pth")
            mt = LightGlueMatcher("superpoint", weights_path=f"{tmp}/lg.pth")
        images = [Image(value_array=views_np[i]) for i in range(n_img)]
        shape = (h, w, 1)
        out_d = [det.detect_and_describe(im) for im in images[:2]]  # warm-up: lazy model build, allocator (the process has just run
        for _ in range(3):                                          # the batched workloads: its caching allocator regroups), first launches
            mt.match(out_d[0][0], out_d[1][0], out_d[0][1], out_d[1][1], shape, shape)
        torch.cuda.synchronize(device)
        feats, each_det = [], []
        for im in images:  # per call, like the matches: the median image decides (one allocator regrouping after the batched legs once cost 50 ms)
            t0 = time.perf_counter()
            feats.append(det.detect_and_describe(im))
            each_det.append(time.perf_counter() - t0)
        t_det = float(np.median(each_det)) * len(images)
        got, each = [], []
        for i, j in pairs:
            t0 = time.perf_counter()
            got.append(mt.match(feats[i][0], feats[j][0], feats[i][1], feats[j][1], shape, shape))
            each.append(time.perf_counter() - t0)
        t_match = float(sum(each))
        # several worker threads on ONE matcher object (GTSfM's --threads_per_worker, gtsfm/runner.py:155,436): every concurrent call runs
        # on a lane of its own (matcher_engine._MatcherBase._lane) and fills the part of the chip one pair's launches leave idle
        threaded = {}
        for nthreads in (2, 3):
            def work(tid, nthreads=nthreads):
                for q in range(tid, 2 * len(pairs), nthreads):
                    i, j = pairs[q % len(pairs)]
                    mt.match(feats[i][0], feats[j][0], feats[i][1], feats[j][1], shape, shape)
            for rep in range(2):  # the first round creates the lanes (workspace, staging buffers)
                ths = [threading.Thread(target=work, args=(tid,)) for tid in range(nthreads)]
                t0 = time.perf_counter()
                for th in ths:
                    th.start()
                for th in ths:
                    th.join()
                threaded[str(nthreads)] = round(2 * len(pairs) / (time.perf_counter() - t0), 1)
        # the same calls under the opt-in arithmetics (both switches; the image cache is emptied so that nothing computed in exact fp32 is reused)
        def switched(mode):
            try:
                mt._model.release_lanes()
                old_env = {k: os.environ.get(k) for k in ("GTSFM_ATTENTION_MATH", "GTSFM_GEMM_MATH")}
                os.environ.update({"GTSFM_ATTENTION_MATH": mode, "GTSFM_GEMM_MATH": mode})
                try:
                    got_x3, each_x3 = [], []
                    for rep in range(2):  # the first round uploads the images (misses), the second is all hits
                        got_x3, each_x3 = [], []
                        for i, j in pairs:
                            t0 = time.perf_counter()
                            got_x3.append(mt.match(feats[i][0], feats[j][0], feats[i][1], feats[j][1], shape, shape))
                            each_x3.append(time.perf_counter() - t0)
                finally:
                    for k, v in old_env.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v
                    mt._model.release_lanes()
                per = float(np.median(each_x3))
                return {"match_ms_per_pair_resident": round(per * 1e3, 2), "pairs_per_s_match_only_resident": round(1.0 / per, 1),
                        "calls_with_match_arrays_identical_to_exact_fp32": int(sum(np.array_equal(a, b) for a, b in zip(got, got_x3))), "calls": len(pairs),
                        "switches": f"GTSFM_ATTENTION_MATH={mode} GTSFM_GEMM_MATH={mode} (opt-in)"}
            except Exception as exc:  # noqa: BLE001 - a side measurement must not cost the leg
                return {"error": f"{type(exc).__name__}: {str(exc)[:200]}"}

        x3, h2 = switched("bf16x3"), switched("f16x2")
    # the same pair through the batched, device-resident pipeline (device top-k keeps detection order, the plugin's Keypoints.get_top_k does
    # not, so the index pairs are compared as coordinate pairs)
    dev_feats = pipe.detect(torch.from_numpy(views_np[:n_img]).to(device))
    res = pipe.match(dev_feats, pairs[:1], [(h, w)] * n_img)
    ref = matches_to_numpy(res)[pairs[0]]
    xy = dev_feats["xy"].cpu().numpy()
    (i, j) = pairs[0]
    ref_set = {(tuple(xy[i, a]), tuple(xy[j, b])) for a, b in ref}
    got_set = {(tuple(feats[i][0].coordinates[a]), tuple(feats[j][0].coordinates[b])) for a, b in got[0]}
    per_pair = t_match / len(pairs)
    per_img = t_det / n_img
    return {
        "detect_ms_per_image": round(per_img * 1e3, 2), "match_ms_per_pair": round(per_pair * 1e3, 2),
        "match_ms_each_call": [round(t * 1e3, 2) for t in each],
        "images_per_s": round(1.0 / per_img, 1), "pairs_per_s_match_only": round(1.0 / per_pair, 1),
        "pairs_per_s_match_only_by_worker_threads": {"1": round(1.0 / per_pair, 1), **threaded},
        "value": round(1.0 / (per_pair + per_img * args.images / max(1, args.pairs)), 1), "unit": "image-pairs/s",
        "value_note": f"exhaustive scene of the headline's shape ({args.images} images, {args.pairs} pairs): 1 / (match + detect x images / pairs), PCIe and per-call synchronisation included",
        "keypoints_per_image": [int(min(len(f[0]) for f in feats)), int(max(len(f[0]) for f in feats))], "matches_first_pair": int(len(got[0])),
        "first_pair_equals_batched_pipeline": bool(ref_set == got_set),
        "bf16x3": x3, "f16x2": h2,
        "match_ms_per_pair_resident": round(float(np.median(each[5:])) * 1e3, 2), "pairs_per_s_match_only_resident": round(1.0 / float(np.median(each[5:])), 1),
        "workload": f"SuperPointDetectorDescriptor.detect_and_describe x {n_img} + {type(mt).__name__}.match x {len(pairs)} (numpy in / numpy out, one call at a time)",
    }


if __name__ == "__main__":
    main()
