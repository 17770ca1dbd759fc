#!/usr/bin/env python
"""Dev tool: host-side throughput of the dataset mapper (JPEG decode + ResizeShortestEdge + colour augmentations + box
transforms) on synthetic 640x480 JPEGs with the ctdet_dla_34_1x.yaml input settings (short edge 640..800, max 1333).
usage: python tools/bench_loader.py [n_images] [workers]"""
import json
import os
import sys
import tempfile
import time

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectron2_centernet_amd.config import get_cfg  # noqa: E402
from detectron2_centernet_amd.data import TrafficLightDatasetMapper, load_coco_json  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 16
root = tempfile.mkdtemp(prefix="ctdet_loader_")
rng = np.random.RandomState(0)
images, anns = [], []
for i in range(n):
    base = rng.randint(0, 256, (30, 40, 3)).astype(np.uint8)
    arr = np.asarray(Image.fromarray(base).resize((640, 480), Image.BICUBIC))      # smooth content: realistic JPEG sizes
    Image.fromarray(arr).save(os.path.join(root, f"{i}.jpg"), quality=90)
    images.append({"id": i, "file_name": f"{i}.jpg", "height": 480, "width": 640})
    for j in range(8):
        x, y = rng.randint(0, 500), rng.randint(0, 380)
        anns.append({"id": i * 8 + j, "image_id": i, "bbox": [int(x), int(y), 100, 80], "category_id": 1 + j % 3, "iscrowd": 0})
jf = os.path.join(root, "ann.json")
json.dump({"images": images, "annotations": anns, "categories": [{"id": k, "name": str(k)} for k in (1, 2, 3)]}, open(jf, "w"))
recs = load_coco_json(jf, root)
cfg = get_cfg()
cfg.INPUT.MIN_SIZE_TRAIN = (640, 672, 704, 736, 768, 800)
cfg.INPUT.MAX_SIZE_TRAIN = 1333
mapper = TrafficLightDatasetMapper(cfg, True)
np.random.seed(0)
for r in recs[:4]:
    mapper(r)
t0 = time.perf_counter()
for r in recs:
    mapper(r)
t1 = time.perf_counter()
print(f"1 thread: {n / (t1 - t0):.1f} images/s ({1e3 * (t1 - t0) / n:.1f} ms per image)")
from concurrent.futures import ThreadPoolExecutor  # noqa: E402
with ThreadPoolExecutor(workers) as pool:
    t0 = time.perf_counter()
    for _ in range(3):
        list(pool.map(mapper, recs))
    t1 = time.perf_counter()
print(f"{workers} threads: {3 * n / (t1 - t0):.1f} images/s  (host cores: {os.cpu_count()})")
# the training loader proper: worker processes, batches of 16
from detectron2_centernet_amd.data import build_detection_train_loader, register_coco_instances  # noqa: E402
register_coco_instances("bench_loader_ds", {}, jf, root)
cfg.DATASETS.TRAIN = ("bench_loader_ds",)
cfg.SOLVER.IMS_PER_BATCH = 16
it = build_detection_train_loader(cfg, num_workers=workers)
for _ in range(4):
    next(it)
t0 = time.perf_counter()
nb = 24
for _ in range(nb):
    batch = next(it)
t1 = time.perf_counter()
print(f"DataLoader, {workers} worker processes: {nb * 16 / (t1 - t0):.1f} images/s (batches of 16, "
      f"{tuple(batch[0]['image'].shape)} ...)")
