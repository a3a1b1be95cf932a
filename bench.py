"""Benchmark of the MI355X deep front-end (BASELINE.json: image-pairs/sec, detect+match @1024 px).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input resident in HBM: ``--images`` seeded
1024x1024 gray images are detected + described (SuperPoint) and all ``images*(images-1)/2`` exhaustive pairs are
matched (``--matcher``). Every rank processes its own batch (weak scaling, no data-path collective: images and pairs
are independent units); weights are packed on rank 0 and broadcast over RCCL. Rank 0 prints
ONE JSON line with the whole-job rate, the roofline of the dominant kernel (measured live with HIP events on the
launch stream) and a CPU baseline (the oracle, timed on a bounded sample of the same workload).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

from gtsfm_amd.utils import synthetic  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
HBM_PEAK_GBS = 8000.0


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


def pmc_traffic(kernel: str):
    """HBM bytes per launch measured with rocprofv3 PMC passes and committed under profiles/ (bench.py cannot collect
    counters itself); None when the file is missing."""
    path = REPO / "profiles" / "r01_pmc_traffic.json"
    try:
        return json.loads(path.read_text()).get(kernel)
    except (OSError, ValueError):
        return None


def measure_conv_roofline(lib, device, batch: int, h: int, w: int, reps: int = 5):
    """Times the dominant kernel (conv3x3_mfma_kernel) launch by launch with HIP events on the launch stream."""
    from gtsfm_amd.runtime import lib as L

    stream = torch.cuda.current_stream(device)
    total_ms, total_flops, launches = 0.0, 0.0, 0
    for cin, cout, hh, ww, pool in superpoint_conv3x3_layers(h, w):
        x = torch.randn((batch, hh, ww, cin), device=device)
        ho, wo = (hh // 2, ww // 2) if pool else (hh, ww)
        y = torch.empty((batch, ho, wo, cout), device=device)
        wp = torch.randn(lib.gtsfm_packed_conv3x3_floats(cin, cout), device=device) * 0.05
        bias = torch.zeros((cout + 63) // 64 * 64, device=device)
        args = (x.data_ptr(), cin, 0, y.data_ptr(), cout, 0, wp.data_ptr(), bias.data_ptr(), batch, hh, ww, cin, cout, 1, pool,
                stream.cuda_stream)
        L.check(lib.gtsfm_conv3x3_f32(*args), "conv3x3")
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            L.check(lib.gtsfm_conv3x3_f32(*args), "conv3x3")
        e1.record(stream)
        e1.synchronize()
        total_ms += e0.elapsed_time(e1) / reps
        total_flops += 2.0 * 9 * cin * cout * hh * ww * batch
        launches += 1
        del x, y, wp
    achieved = total_flops / (total_ms * 1e-3) / 1e12
    t = pmc_traffic("conv3x3_mfma_kernel") if (h, w) == (1024, 1024) else None
    return {
        "bound": "mfma",
        "kernel": "conv3x3_mfma_kernel",
        "achieved": round(achieved, 2),
        "peak": FP32_MFMA_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
        "traffic": None if t is None else (t["fetch_bytes_per_image"] + t["write_bytes_per_image"]) * batch,
        "traffic_note": None if t is None else "HBM bytes per launch (avg over the 8 launches), rocprofv3 PMC, profiles/r01_pmc_traffic.json",
        "launches_per_step": launches,
        "avg_launch_ms": round(total_ms / launches, 4),
        "flops_per_step": total_flops,
    }


def measure_attention_roofline(lib, device, n: int, npairs: int, reps: int = 5):
    """Dominant kernel of the detect+match workload (attention_mfma_kernel, ~50 % of GPU time): one launch = the self
    attention of `npairs` pairs (2 sequences x 4 heads each) at N = n, timed with HIP events on the launch stream.
    Algorithmic work: 1024 * N^2 FLOP per sequence per layer."""
    from gtsfm_amd.runtime import lib as L

    stream = torch.cuda.current_stream(device)
    nseq = 2 * npairs
    qkv = torch.randn((nseq * n, 768), device=device)
    out = torch.empty((nseq * n, 256), device=device)
    probs = torch.tensor([[s * n, s, s * n, s] for s in range(nseq)], dtype=torch.int32, device=device)
    counts = torch.full((nseq,), n, dtype=torch.int32, device=device)
    args = (qkv.data_ptr(), 768, qkv.data_ptr() + 256 * 4, 768, qkv.data_ptr() + 512 * 4, 768, out.data_ptr(), 256, probs.data_ptr(),
            counts.data_ptr(), nseq, n, 4, 0.125, stream.cuda_stream)
    L.check(lib.gtsfm_attention_f32(*args), "attention")
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        L.check(lib.gtsfm_attention_f32(*args), "attention")
    e1.record(stream)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 1024.0 * n * n * nseq
    achieved = flops / (ms * 1e-3) / 1e12
    t = pmc_traffic("attention_mfma_kernel") if (n, nseq) == (2048, 64) else None
    return {
        "bound": "mfma",
        "kernel": "attention_mfma_kernel",
        "achieved": round(achieved, 2),
        "peak": FP32_MFMA_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
        "traffic": None if t is None else t["fetch_bytes"] + t["write_bytes"],
        "traffic_note": None if t is None else "HBM bytes per launch, rocprofv3 PMC, profiles/r01_pmc_traffic.json",
        "avg_launch_ms": round(ms, 4),
        "flops_per_launch": flops,
        "launch_shape": f"{nseq} sequences x 4 heads, N = {n} queries = keys, head_dim 64",
    }


def measure_gemm_roofline(lib, device, rows: int, k: int, n: int, reps: int = 5):
    """gemm_mfma_kernel at one of the matcher's projection shapes (rows x k -> n)."""
    from gtsfm_amd.runtime import lib as L

    stream = torch.cuda.current_stream(device)
    a = torch.randn((rows, k), device=device)
    w = torch.randn(lib.gtsfm_packed_linear_floats(k, n), device=device) * 0.05
    bias = torch.zeros((n + 63) // 64 * 64, device=device)
    c = torch.empty((rows, n), device=device)
    # the matchers run their projections through the row-major (LDS-DMA) entry point when k % 32 == 0; GTSFM_GEMM=mfma
    # measures the register-staged kernel instead
    if k % 32 == 0 and os.environ.get("GTSFM_GEMM", "") != "mfma":
        wr = torch.randn((n, k), device=device) * 0.05
        fn = lib.gtsfm_linear_rowmajor_f32
        args = (a.data_ptr(), k, rows, None, k, wr.data_ptr(), k, bias.data_ptr(), n, None, c.data_ptr(), n, 0, None, 0, 1.0, 0, stream.cuda_stream)
    else:
        fn = lib.gtsfm_linear_f32
        args = (a.data_ptr(), k, rows, None, k, w.data_ptr(), bias.data_ptr(), n, c.data_ptr(), n, 0, None, 0, 1.0, 0, stream.cuda_stream)
    L.check(fn(*args), "linear")
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        L.check(fn(*args), "linear")
    e1.record(stream)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    achieved = 2.0 * rows * k * n / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "gemm_mfma_kernel", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "avg_launch_ms": round(ms, 4), "launch_shape": f"{rows} x {k} -> {n}"}


def cpu_baseline(h: int, w: int, n_images: int, matcher: str, n_keypoints: int, sinkhorn_iters: int):
    """The oracle (kind "port": restatement of the reference's torch CPU path, bit-exact with it in the build
    container) on a bounded sample: ``n_images`` detections and one pair match, on all host cores."""
    from oracle import superpoint_oracle

    cores = min(os.cpu_count() or 1, int(os.environ.get("GTSFM_CPU_BASELINE_THREADS", "16")))
    torch.set_num_threads(cores)
    sd = synthetic.synthetic_superpoint_state_dict()
    t_det = []
    feats = []
    for i in range(n_images):
        gray = synthetic.synthetic_gray_image(h, w, 1000 + i)
        t0 = time.perf_counter()
        c, s, d = superpoint_oracle.detect_and_describe(sd, gray, max_keypoints=n_keypoints)
        t_det.append(time.perf_counter() - t0)
        feats.append((c, s, d))
    det_s = float(np.median(t_det))
    out = {"detect_s_per_image": round(det_s, 3), "cores": cores, "kind": "port"}
    if matcher == "none":
        out.update(value=round(1.0 / det_s, 4), unit="images/s", sample=f"{n_images} x SuperPoint {h}x{w} (oracle, fp32)")
        return out
    (c0, s0, d0), (c1, s1, d1) = feats[0], feats[1]
    t0 = time.perf_counter()
    if matcher == "superglue":
        from oracle import superglue_oracle

        sg = synthetic.synthetic_superglue_state_dict()
        superglue_oracle.match(sg, c0, c1, s0, s1, d0, d1, (h, w, 1), (h, w, 1), sinkhorn_iterations=sinkhorn_iters)
    else:
        from oracle import lightglue_oracle

        lg = synthetic.synthetic_lightglue_state_dict()
        lightglue_oracle.match(lg, c0, c1, d0, d1, (h, w, 1), (h, w, 1))
    match_s = time.perf_counter() - t0
    # independent-pair cost on the CPU path: 2 detections + 1 match
    out.update(
        match_s_per_pair=round(match_s, 3),
        value=round(1.0 / (2 * det_s + match_s), 4),
        unit="image-pairs/s",
        sample=f"{n_images} x SuperPoint {h}x{w} + 1 x {matcher} pair at N={len(c0)},{len(c1)} (oracle, fp32); "
        "rate = 1 / (2 detect + 1 match)",
    )
    return out


def matcher_flops(matcher: str, n: int, layers: float, sinkhorn: int) -> float:
    """SURVEY.md section 8(d) per-pair dense FLOP (N = M keypoints)."""
    if matcher == "superglue":
        return 2 * 217280 * n + 36 * (1310720 * n + 1024 * n * n) + 262144 * n + 512 * n * n
    return layers * 2 * (2490368 * n + 1792 * n * n) + 262144 * n + 512 * n * n


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images", type=int, default=46, help="images per rank per step (46 -> 1035 exhaustive pairs, first --pairs kept)")
    ap.add_argument("--pairs", type=int, default=1000, help="exhaustive (i<j) pairs matched per rank per step")
    ap.add_argument("--size", type=int, default=1024, help="square image side (overridden by --height / --width)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--matcher", choices=["none", "lightglue", "superglue"], default="lightglue")
    ap.add_argument("--keypoints", type=int, default=2048, help="keypoints kept per image (device top-k by response)")
    ap.add_argument("--sinkhorn", type=int, default=100, help="SuperGlue Sinkhorn iterations (GTSfM runs 20; BASELINE config 4 asks for 100)")
    ap.add_argument("--pair-definition", choices=["exhaustive", "independent"], default="exhaustive",
                    help="exhaustive: (i<j) pairs of --images images, each detected once per step (the headline); independent: "
                         "--pairs disjoint pairs, 2 fresh detections per pair (SURVEY.md section 8d asks for both rates)")
    ap.add_argument("--pair-chunk", type=int, default=32)
    ap.add_argument("--streams", type=int, default=2, help="HIP streams the pair chunks alternate over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X visible to PyTorch-ROCm (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    # GTSFM_BENCH_FORCE_DIST=1 exercises the RCCL code path (init, weight broadcast, barrier, max-reduce) on one GPU
    if world > 1 or os.environ.get("GTSFM_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from gtsfm_amd import parallel
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
    matcher = None
    if args.matcher == "superglue":
        matcher = ME.SuperGlueEngine(synthetic.synthetic_superglue_state_dict(), device)
    elif args.matcher == "lightglue":
        matcher = ME.LightGlueEngine(synthetic.synthetic_lightglue_state_dict(), device)
    if matcher is not None and dist is not None:
        blob = matcher.weights if rank == 0 else None
        matcher.weights = parallel.broadcast_packed_weights(blob, matcher.weights.numel(), device)

    h = args.height or args.size
    w = args.width or args.size
    n = args.images
    # overlapping views: every image is a crop of one seeded canvas, shifted by multiples of the 8-px SuperPoint cell, so
    # that exhaustive pairs share content (non-trivial match lists) while every image is still detected independently
    canvas = synthetic.synthetic_gray_image(h + 8 * n, w + 8 * n, 1000 + rank)
    imgs = np.stack([canvas[8 * i : 8 * i + h, 8 * ((7 * i) % n) : 8 * ((7 * i) % n) + w] for i in range(n)])
    images = torch.from_numpy(imgs).to(device)  # inputs resident in HBM before the timed region
    pairs = parallel.exhaustive_pairs(n)[: args.pairs] if matcher is not None else []
    independent = args.pair_definition == "independent" and matcher is not None
    if independent:
        # 2 P image slots, every slot detected afresh each step (slot s shows view (5 s) % n; nothing is cached or shared
        # between slots), pair p = slots (2p, 2p + 1)
        slots = torch.arange(2 * args.pairs, device=device)
        images = images[(5 * slots) % n].contiguous()
        n = 2 * args.pairs
        pairs = [(2 * p, 2 * p + 1) for p in range(args.pairs)]
    pipe = FrontEndPipeline(detector, matcher, max_keypoints=args.keypoints, pair_chunk=args.pair_chunk, num_streams=args.streams)
    shapes = [(h, w)] * n
    mk = {"sinkhorn_iterations": args.sinkhorn} if args.matcher == "superglue" else {}

    def step():
        feats = pipe.detect(images)
        res = pipe.match(feats, pairs, shapes, **mk) if matcher is not None else []
        return feats, res

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        feats, res = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        feats, res = step()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    detect_only = matcher is None
    units_per_step = (n if detect_only else len(pairs)) * world
    value = units_per_step / (ms_per_step * 1e-3)

    if rank == 0:
        kcount = feats["count"].tolist()
        layers = float(torch.cat([r["stop"] for r in res]).float().mean()) if args.matcher == "lightglue" else 18.0
        nmatch = int(sum(int((r["matches"] > -1).sum()) for r in res)) // 2 if res else 0
        flops_step = superpoint_flops(h, w) * n + (matcher_flops(args.matcher, args.keypoints, layers, args.sinkhorn) * len(pairs) if res else 0)
        result = {
            "metric": f"images/sec (SuperPoint detect+describe) @{h}x{w}" if detect_only else f"image-pairs/sec (detect+match) @{max(h, w)}px",
            "value": round(value, 2),
            "unit": "images/s" if detect_only else "image-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (
                    f"SuperPoint-only: {n} synthetic {h}x{w} gray images per GPU per step" if detect_only else
                    f"SuperPoint+{args.matcher}: {len(pairs)} independent pairs = {n} fresh detections of synthetic {h}x{w} gray images per GPU "
                    f"per step, top-{args.keypoints} keypoints per image" if independent else
                    f"SuperPoint+{args.matcher}: {len(pairs)} exhaustive (i<j) pairs of {n} synthetic {h}x{w} gray images per GPU per step "
                    f"(each image detected once per step), top-{args.keypoints} keypoints per image"
                ),
                "pair_definition": args.pair_definition if not detect_only else None,
                "images_per_gpu_per_step": n,
                "pairs_per_gpu_per_step": len(pairs),
                "keypoints_per_image": [int(min(kcount)), int(max(kcount))],
                "matcher_layers_run": layers,
                "sinkhorn_iterations": args.sinkhorn if args.matcher == "superglue" else None,
                "matches_per_pair": round(nmatch / max(1, len(pairs)), 1),
                "weights": "seeded synthetic (gtsfm_amd.utils.synthetic)",
                "parallelism": f"dp{world}: independent image sets / pair lists per rank, RCCL weight broadcast, no data-path collective",
                "pair_chunk": args.pair_chunk,
                "streams": args.streams,
            },
            "tflops": round(flops_step * world / (ms_per_step * 1e-3) / 1e12, 2),
        }
        conv_roof = measure_conv_roofline(lib, device, min(n, 16), h, w)
        if detect_only:
            result["roofline"] = conv_roof
        else:  # dominant kernel of this workload first; the other two MFMA kernels alongside
            result["roofline"] = measure_attention_roofline(lib, device, args.keypoints, min(args.pair_chunk, max(1, len(pairs))))
            result["roofline_other"] = [
                measure_gemm_roofline(lib, device, 2 * min(args.pair_chunk, max(1, len(pairs))) * args.keypoints, 256, 768),
                conv_roof,
            ]
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            result["cpu_baseline"] = cpu_baseline(h, w, 2, args.matcher, args.keypoints, args.sinkhorn)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
