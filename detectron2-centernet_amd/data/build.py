"""Training sampler and a minimal batched loader (reference: data/samplers/distributed_sampler.py:12-55 TrainingSampler,
data/build.py:270-355 build_detection_train_loader / per-GPU batch size).  The loader is a plain generator over mapped
samples (optionally through a thread pool): the GPU path takes `list[dict]` batches of unequal image sizes as they come."""
import itertools
from concurrent.futures import ThreadPoolExecutor

import torch

from .catalog import DatasetCatalog
from .dataset_mapper import TrafficLightDatasetMapper


class TrainingSampler:
    """infinite stream of indices: every rank draws the same seeded permutations and takes elements rank, rank+W, ..."""

    def __init__(self, size, shuffle=True, seed=0, rank=0, world_size=1):
        assert size > 0
        self._size, self._shuffle, self._seed, self._rank, self._world = size, shuffle, int(seed), rank, world_size

    def __iter__(self):
        yield from itertools.islice(self._infinite_indices(), self._rank, None, self._world)

    def _infinite_indices(self):
        g = torch.Generator()
        g.manual_seed(self._seed)
        while True:
            if self._shuffle:
                yield from torch.randperm(self._size, generator=g).tolist()
            else:
                yield from range(self._size)


def filter_images_with_only_crowd_annotations(dataset_dicts):
    """data/build.py:41-68"""
    def valid(anns):
        return any(a.get("iscrowd", 0) == 0 for a in anns)
    return [d for d in dataset_dicts if valid(d.get("annotations", []))]


def build_detection_train_loader(cfg, mapper=None, rank=0, world_size=1, seed=0, num_workers=None):
    """yields lists of `IMS_PER_BATCH // world_size` mapped samples forever"""
    names = cfg.DATASETS.TRAIN
    dicts = list(itertools.chain.from_iterable(DatasetCatalog.get(n) for n in names))
    if cfg.DATALOADER.FILTER_EMPTY_ANNOTATIONS and dicts and "annotations" in dicts[0]:
        dicts = filter_images_with_only_crowd_annotations(dicts)
    assert len(dicts), "no training images"
    total = cfg.SOLVER.IMS_PER_BATCH
    assert total % world_size == 0, f"IMS_PER_BATCH ({total}) must be divisible by the number of workers ({world_size})"
    per_gpu = total // world_size
    mapper = mapper if mapper is not None else TrafficLightDatasetMapper(cfg, True)
    sampler = iter(TrainingSampler(len(dicts), seed=seed, rank=rank, world_size=world_size))
    workers = cfg.DATALOADER.NUM_WORKERS if num_workers is None else num_workers

    def gen():
        pool = ThreadPoolExecutor(workers) if workers > 0 else None
        try:
            while True:
                idx = [next(sampler) for _ in range(per_gpu)]
                if pool is not None:
                    yield list(pool.map(lambda i: mapper(dicts[i]), idx))
                else:
                    yield [mapper(dicts[i]) for i in idx]
        finally:
            if pool is not None:
                pool.shutdown(wait=False)
    return gen()
