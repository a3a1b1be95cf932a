import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, bench
from gtsfm_amd.runtime import lib as L
lib = L.load(); dev = torch.device("cuda:0")
for _ in range(3): bench.measure_conv_roofline(lib, dev, 8, 1024, 1024, reps=3)
for b, h, w in ((4, 1024, 1024), (16, 1024, 1024), (16, 480, 640)):
    v = [bench.measure_conv_roofline(lib, dev, b, h, w, reps=3)["frac"] * 100 for _ in range(3)]
    print(f"conv stack batch {b} {h}x{w}: " + " ".join(f"{x:.1f}" for x in v), flush=True)
