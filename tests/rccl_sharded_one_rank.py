"""ShardedDetDescCorrespondenceGenerator in JOINED mode on a one-rank "nccl" process group (tests/test_rccl_gpu.py runs this as a
subprocess so that the pytest process keeps no process group): init_process_group, the weight broadcast, the all_to_all_single exchange,
the ragged match gather and the keypoint all-gather all execute in RCCL on the device. Prints one JSON line: a digest over keypoints and
match arrays, which the test compares with the single-process generator's."""
import hashlib
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gtsfm_amd import parallel  # noqa: E402


def build(tmp: str, matcher: str, num_gpus):
    from gtsfm_amd.frontend.correspondence_generator.sharded_det_desc_correspondence_generator import ShardedDetDescCorrespondenceGenerator
    from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor
    from gtsfm_amd.frontend.matcher.lightglue_matcher import LightGlueMatcher
    from gtsfm_amd.frontend.matcher.superglue_matcher import SuperGlueMatcher
    from gtsfm_amd.utils import synthetic

    torch.save(synthetic.synthetic_superpoint_state_dict(), f"{tmp}/sp.pth")
    det = SuperPointDetectorDescriptor(max_keypoints=400, weights_path=f"{tmp}/sp.pth")
    if matcher == "superglue":
        torch.save(synthetic.synthetic_superglue_state_dict(num_layers=4), f"{tmp}/sg.pth")
        mt = SuperGlueMatcher(weights_path=f"{tmp}/sg.pth")
    else:
        torch.save(synthetic.synthetic_lightglue_state_dict(num_layers=3), f"{tmp}/lg.pth")
        mt = LightGlueMatcher("superpoint", weights_path=f"{tmp}/lg.pth")
    return ShardedDetDescCorrespondenceGenerator(mt, det, num_gpus=num_gpus)


def scene():
    from gtsfm_amd.common.image import Image
    from gtsfm_amd.utils import synthetic

    views = synthetic.synthetic_overlapping_views(6, 240, 320, 1000)
    images = [Image(value_array=v) for v in views]
    images[3] = Image(value_array=views[3], mask=np.zeros((240, 320), dtype=np.uint8))  # fully masked: no keypoints, its pairs come back empty
    images[4] = Image(value_array=np.repeat(views[4][:, :, None], 3, axis=2))  # an RGB view: gray conversion on the device
    return images, parallel.exhaustive_pairs(6)[:13]


def digest(kps, matches, pairs) -> dict:
    h = hashlib.sha1()
    for k in kps:
        h.update(np.ascontiguousarray(k.coordinates).tobytes())
        h.update(np.ascontiguousarray(k.responses).tobytes())
    for p in pairs:
        h.update(np.asarray(p, dtype=np.int64).tobytes())
        h.update(np.ascontiguousarray(matches[p], dtype=np.int64).tobytes())
    return {"digest": h.hexdigest(), "keypoints": [len(k) for k in kps], "matches": int(sum(len(matches[p]) for p in pairs)),
            "dtype": str(matches[pairs[0]].dtype)}


if __name__ == "__main__":
    import tempfile

    matcher = sys.argv[1]
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    with tempfile.TemporaryDirectory() as tmp:
        gen = build(tmp, matcher, None)
        images, pairs = scene()
        kps, matches = gen.generate_correspondences(None, images, pairs)
        out = digest(kps, matches, pairs)
        out["backend"] = dist.get_backend()
        out["table_images"] = gen.last_scene.plan.table_images
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)
