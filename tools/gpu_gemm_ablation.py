"""GEMM ablation matrix under hot clocks (GPU box). GTSFM_GEMM_DEBUG bits: 1 = skip the epilogue, 2 = skip the A-row
loads. Each configuration runs in a child process; 12 warm-up measurements precede the 5 reported ones (the first
launches after idle run 15-20 % slower while the clocks ramp). DBG_LIST=0,1,2,3 python tools/gpu_gemm_ablation.py"""
import sys, os, time, subprocess
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
if len(sys.argv) > 1:
    import torch, bench
    from gtsfm_amd.runtime import lib as L
    lib = L.load(); dev = torch.device("cuda:0")
    for shape in ((131072, 256, 768), (131072, 512, 512), (131072, 512, 256), (131072, 256, 256), (131072, 256, 512)):
        for _ in range(12): bench.measure_gemm_roofline(lib, dev, *shape, reps=8)
        v = [bench.measure_gemm_roofline(lib, dev, *shape, reps=8)['frac'] * 100 for _ in range(5)]
        print(f"debug={os.environ.get('GTSFM_GEMM_DEBUG','0'):>2} gemm {shape}: " + " ".join(f"{x:.1f}" for x in v), flush=True)
else:
    for d in [int(x) for x in os.environ.get("DBG_LIST", "0").split(",")]:
        env = dict(os.environ, GTSFM_GEMM_DEBUG=str(d))
        subprocess.run([sys.executable, __file__, "child"], env=env)
