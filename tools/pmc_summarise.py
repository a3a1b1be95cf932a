"""Turn the rocprofv3 PMC passes of tools/prof_r05.sh into (1) pmc_raw_summary.json: per kernel and grid the dispatch count and the average
counter value, (2) r05_pmc_traffic.json: HBM-side bytes per launch under the keys bench.py's `pmc_traffic()` looks up, (3) sq_summary.csv: the
SQ counters per kernel with the derived matrix-pipe utilisation, VALU share and effective clock.

Units and corrections exactly as /opt/skills/guides/MI355X_MICROARCH.md prescribes (section HBM): FETCH_SIZE / WRITE_SIZE are reported in KiB;
on gfx950 FETCH_SIZE reports half the bytes of wide coalesced streaming reads -> doubled; WRITE_SIZE is taken as reported. The counters sit on
the L2's fabric side: Infinity-Cache hits are counted.

    python tools/pmc_summarise.py gpurun_out/prof_r05"""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
ROUND = sys.argv[2] if len(sys.argv) > 2 else "r06"  # prefix of the traffic file bench.py reads (bench.PMC_TRAFFIC_FILE)


def collect(tag_dir):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(tag_dir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0][:90] + "|grid" + r.get("Grid_Size", "?")
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return {k: {"dispatches": n, "avg": s / n, "sum": s} for k, (n, s) in agg.items() if "kernel" in k and "elementwise" not in k and "distribution" not in k}


raw = {os.path.basename(d)[4:]: collect(d) for d in sorted(glob.glob(out + "/pmc_*_*SIZE"))}
json.dump(raw, open(out + "/pmc_raw_summary.json", "w"), indent=1)


def per_launch(tag, needle, launches_per_call=1, which=max):
    """(fetch bytes, write bytes) per launch of the kernels whose name contains `needle` in pass pair `tag`: the per-dispatch averages of the
    heaviest grid (the timed launches; warm-up launches have the same grid) -- None when absent."""
    f, w = raw.get(tag + "_FETCH_SIZE", {}), raw.get(tag + "_WRITE_SIZE", {})
    fk = [k for k in f if needle in k]
    wk = [k for k in w if needle in k]
    if not fk or not wk:
        return None
    # a sweep launches every tier its batch can contain (the blocks of the tiers a pair does not belong to return at once): one dispatch of EACH
    # matching kernel per iteration -> the per-iteration figure is the SUM of the kernels' per-dispatch averages, not their mean
    fsum = sum(f[k]["avg"] for k in fk) * launches_per_call
    wsum = sum(w[k]["avg"] for k in wk) * launches_per_call
    return int(2 * 1024 * fsum), int(1024 * wsum)


def by_grid(tag, needle):
    """{grid: (fetch bytes, write bytes)} per dispatch for kernels matching `needle` (launch shapes differ by grid)."""
    f, w = raw.get(tag + "_FETCH_SIZE", {}), raw.get(tag + "_WRITE_SIZE", {})
    res = {}
    for k in f:
        if needle in k and k in w:
            res[k.split("|grid")[1]] = (int(2 * 1024 * f[k]["avg"]), int(1024 * w[k]["avg"]), f[k]["dispatches"])
    return res


traffic = {"_comment": (f"HBM-side traffic from rocprofv3 PMC passes of round {ROUND[1:].lstrip('0')}, on the round's final tree (tools/prof_{ROUND}.sh -> tools/pmc_summarise.py; FETCH_SIZE and WRITE_SIZE in "
                        "separate passes, KiB units, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md, WRITE_SIZE as reported). Counters sit at the L2's fabric side: "
                        f"Infinity-Cache hits are counted. Raw per-dispatch averages: profiles/{ROUND}_pmc_raw_summary.json. Launch shapes = the headline's (16-pair chunks at the "
                        "5000-keypoint cap).")}
n, pairs, nseq, cap = 5000, 16, 32, 5120
rows = 2 * pairs * cap
a = per_launch("all", "attention_dma_kernel")
if a:
    traffic[f"attention_dma_kernel@{nseq}x4x{n}"] = {"launch_shape": f"{nseq} sequences x 4 heads, N = {n} (fused schedule)", "fetch_bytes": a[0], "write_bytes": a[1],
                                                   "algorithmic_bytes": 4 * 4 * nseq * n * 256}
gem = by_grid("all", "gemm_dma_walk_kernel")
gem2 = by_grid("gemm_163840_256_512", "gemm_dma_walk_kernel")  # 256 -> 512 shares its grid with 512 -> 512: its own pass pair
shapes = {(256, 768): None, (512, 512): None, (512, 256): None}
# grids: ceil(mtiles / 8) * 8 * ceil(N / 128) x 1 -> identify by the column-block count; 512 -> 512 and 256 -> 512 share a grid and are told apart by bytes fetched
mt = -(-rows // 128)
for (k, nn) in list(shapes):
    grid = (-(-mt // 8) * 8) * -(-nn // 128) * 256
    cands = [(g, v) for g, v in gem.items() if g.isdigit() and int(g) == grid]
    if cands:
        shapes[(k, nn)] = cands
for (k, nn), cands in shapes.items():
    if not cands:
        continue
    g, v = cands[0]
    traffic[f"gemm_dma_walk_kernel@{rows}x{k}x{nn}"] = {"launch_shape": f"{rows} x {k} -> {nn}", "fetch_bytes": v[0], "write_bytes": v[1],
                                                        "algorithmic_bytes": 4 * (rows * k + nn * k + rows * nn)}
for g, v in gem2.items():
    if g.isdigit() and int(g) == (-(-mt // 8) * 8) * 4 * 256:
        traffic[f"gemm_dma_walk_kernel@{rows}x256x512"] = {"launch_shape": f"{rows} x 256 -> 512", "fetch_bytes": v[0], "write_bytes": v[1],
                                                          "algorithmic_bytes": 4 * (rows * 256 + 512 * 256 + rows * 512)}
sg = [(g, v) for g, v in gem.items() if g.isdigit() and int(g) == (-(-(-(-n // 128)) // 8) * 8) * -(-n // 128) * 256]
if sg:
    traffic[f"gemm_dma_walk_kernel@{n}x256x{n}"] = {"launch_shape": f"score matrix of one pair: {n} x 256 -> {n}", "fetch_bytes": sg[0][1][0], "write_bytes": sg[0][1][1],
                                                  "algorithmic_bytes": 4 * (2 * n * 256 + n * n)}
# the chunk's score matrices as ONE ragged launch (gtsfm_score_matrices_f32 = what the forward issues): grid = the one-pair grid x `pairs` problems
one_pair_grid = (-(-(-(-n // 128)) // 8) * 8) * -(-n // 128) * 256
sb = [(g, v) for g, v in gem.items() if g.isdigit() and int(g) == one_pair_grid * pairs]
if sb:
    traffic[f"score_matrices@{pairs}x{n}"] = {"launch_shape": f"score matrices of {pairs} pairs in one ragged launch: {n} x 256 -> {n} each", "fetch_bytes": sb[0][1][0],
                                             "write_bytes": sb[0][1][1], "algorithmic_bytes": 4 * pairs * (2 * n * 256 + n * n)}
sr, sc = per_launch("all", "sinkhorn_rows"), per_launch("all", "sinkhorn_cols_kernel")
if sr and sc:
    traffic[f"sinkhorn_iteration@{pairs}x{n}"] = {"launch_shape": f"{pairs} pairs, ({n}+1) x ({n}+1) couplings, one iteration = rows + cols kernel", "fetch_bytes": sr[0] + sc[0],
                                                "write_bytes": sr[1] + sc[1], "algorithmic_bytes": 4 * (n + 1) * (n + 1) * pairs}
lr, lc = per_launch("all", "lg_rows"), per_launch("all", "lg_cols_kernel")
if lr and lc:
    traffic[f"lg_double_softmax@{pairs}x{n}"] = {"launch_shape": f"{pairs} pairs, {n} x {n} similarities: lg_rows_wide + lg_cols", "fetch_bytes": lr[0] + lc[0], "write_bytes": lr[1] + lc[1],
                                               "algorithmic_bytes": 4 * n * n * pairs}
er, ec, mu = per_launch("all", "extract_rows"), per_launch("all", "extract_cols_kernel"), per_launch("all", "mutual_matches")
if er and ec:
    traffic[f"lg_extract@{pairs}x{n}"] = {"launch_shape": f"{pairs} pairs, {n} x {n}: extract_rows_wide + extract_cols + mutual_matches", "fetch_bytes": er[0] + ec[0] + (mu[0] if mu else 0),
                                        "write_bytes": er[1] + ec[1] + (mu[1] if mu else 0), "algorithmic_bytes": 4 * n * n * pairs}
conv = by_grid("all", "conv3x3_mfma_kernel")
if conv:
    batch = 8
    f = sum(v[0] * v[2] for v in conv.values()) / max(1, sum(v[2] for v in conv.values()))
    # every launch of the stack runs the same number of times: per-image bytes = sum over the distinct launches / batch. Launches sharing a grid are averaged by
    # rocprof per grid, so sum (avg x dispatches) / repetitions, repetitions = dispatches of the rarest grid
    reps = min(v[2] for v in conv.values())
    fetch = sum(v[0] * v[2] for v in conv.values()) / reps / batch
    write = sum(v[1] * v[2] for v in conv.values()) / reps / batch
    traffic["conv3x3_mfma_kernel"] = {"launch_shape": "the 8 conv3x3 launches of one SuperPoint forward at 1024x1024, batch 8, PER IMAGE; first launch = the shipped fused form",
                                      "fetch_bytes_per_image": int(fetch), "write_bytes_per_image": int(write), "algorithmic_bytes_per_image": 453800000,
                                      "algorithmic_note": "every layer's input read once (210.5 MB incl. the 1 MB u8 image) + every layer's (pooled) output written once (243.3 MB)",
                                      "per_grid": {g: {"fetch": v[0], "write": v[1], "dispatches": v[2]} for g, v in conv.items()}}
x3, x3s = per_launch("attention_x3_5000_16", "attention_x3_kernel"), per_launch("attention_x3_5000_16", "attention_x3_split_kernel")
if x3:
    traffic[f"attention_x3_kernel@{nseq}x4x{n}"] = {"launch_shape": f"{nseq} sequences x 4 heads, N = {n} (attention_x3_kernel, fused schedule)", "fetch_bytes": x3[0], "write_bytes": x3[1]}
if x3s:
    traffic[f"attention_x3_split_kernel@{nseq}x4x{n}"] = {"launch_shape": f"K and V of {nseq} sequences x 4 heads, N = {n} -> three bf16 pieces each", "fetch_bytes": x3s[0], "write_bytes": x3s[1]}
h2, h2s = per_launch("attention_f16x2_5000_16", "attention_x3_kernel"), per_launch("attention_f16x2_5000_16", "attention_x3_split_kernel")
if h2:
    traffic[f"attention_x3_kernel@{nseq}x4x{n}@f16x2"] = {"launch_shape": f"{nseq} sequences x 4 heads, N = {n} (attention_x3_kernel<.., 2>, fused schedule)", "fetch_bytes": h2[0], "write_bytes": h2[1]}
if h2s:
    traffic[f"attention_x3_split_kernel@{nseq}x4x{n}@f16x2"] = {"launch_shape": f"K and V of {nseq} sequences x 4 heads, N = {n} -> two fp16 pieces each", "fetch_bytes": h2s[0], "write_bytes": h2s[1]}
json.dump(traffic, open(out + f"/{ROUND}_pmc_traffic.json", "w"), indent=1)
for k, v in traffic.items():
    if isinstance(v, dict):
        fb, wb = v.get("fetch_bytes", v.get("fetch_bytes_per_image")), v.get("write_bytes", v.get("write_bytes_per_image"))
        ab = v.get("algorithmic_bytes", v.get("algorithmic_bytes_per_image"))
        print("TRAFFIC", k, "fetch", fb, "write", wb, "algorithmic", ab, "ratio", round((fb + wb) / ab, 3) if ab else None)

with open(out + "/sq_summary.csv", "w") as o:
    for d in sorted(glob.glob(out + "/sq_*")):
        if not os.path.isdir(d):
            continue
        rows_, trace = collections.defaultdict(lambda: collections.defaultdict(list)), {}
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                rows_[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for path in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                trace.setdefault(r["Kernel_Name"].split("(")[0][:70], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in sorted(rows_.items()):
            if not any(s in k for s in ("attention", "extract", "lg_rows", "lg_cols", "mutual")):
                continue
            ns = sum(trace.get(k, [0])) / max(1, len(trace.get(k, [0])))
            vals = {c: sum(x) / len(x) for c, x in v.items()}
            gui = vals.get("GRBM_GUI_ACTIVE", 0) / 8
            wave = max(1.0, vals.get("SQ_WAVE_CYCLES", 0))
            line = (f'"{k}",avg_ns={ns:.0f},clock_GHz={gui / max(ns, 1):.3f},mfma_util={vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1, 1024 * gui):.3f},'
                    f'valu_issue_share_of_wave_cycles={vals.get("SQ_ACTIVE_INST_VALU", 0) / wave:.3f},wait_any_share={vals.get("SQ_WAIT_ANY", 0) / wave:.3f},'
                    f'wait_inst_share={vals.get("SQ_WAIT_INST_ANY", 0) / wave:.3f},' + ",".join(f"{c}={x:.0f}" for c, x in sorted(vals.items())))
            o.write(line + "\n")
            print("SQ", line[:400])
