"""Training sampler and a minimal batched loader (reference: data/samplers/distributed_sampler.py:12-55 TrainingSampler,
data/build.py:270-355 build_detection_train_loader / per-GPU batch size).  The GPU path takes `list[dict]` batches of unequal image sizes as they come."""
import itertools
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


class _MapDataset(torch.utils.data.Dataset):
    def __init__(self, dicts, mapper):
        self.dicts, self.mapper = dicts, mapper

    def __len__(self):
        return len(self.dicts)

    def __getitem__(self, i):
        return self.mapper(self.dicts[i])


def _trivial_batch_collator(batch):
    return batch


def _worker_init_reset_seed(worker_id):
    """every worker process gets its own numpy / python RNG stream (data/build.py worker_init_reset_seed)"""
    import random

    import numpy as np
    seed = (torch.initial_seed() + worker_id) % 2 ** 31
    np.random.seed(seed)
    random.seed(seed)


def build_detection_train_loader(cfg, mapper=None, rank=0, world_size=1, seed=0, num_workers=None):
    """iterator over lists of `IMS_PER_BATCH // world_size` mapped samples, forever (data/build.py:300-355): a
    torch DataLoader with worker PROCESSES (the mapper is numpy / PIL code under the GIL), infinite TrainingSampler, batches
    kept as lists -- images in a batch have different sizes"""
    names = cfg.DATASETS.TRAIN
    dicts = list(itertools.chain.from_iterable(DatasetCatalog.get(n) for n in names))
    if cfg.DATALOADER.FILTER_EMPTY_ANNOTATIONS and dicts and "annotations" in dicts[0]:
        dicts = filter_images_with_only_crowd_annotations(dicts)
    assert len(dicts), "no training images"
    total = cfg.SOLVER.IMS_PER_BATCH
    assert total % world_size == 0, f"IMS_PER_BATCH ({total}) must be divisible by the number of workers ({world_size})"
    per_gpu = total // world_size
    mapper = mapper if mapper is not None else TrafficLightDatasetMapper(cfg, True)
    sampler = TrainingSampler(len(dicts), seed=seed, rank=rank, world_size=world_size)
    workers = cfg.DATALOADER.NUM_WORKERS if num_workers is None else num_workers
    batch_sampler = torch.utils.data.BatchSampler(sampler, per_gpu, drop_last=True)
    loader = torch.utils.data.DataLoader(_MapDataset(dicts, mapper), batch_sampler=batch_sampler, num_workers=workers,
                                         collate_fn=_trivial_batch_collator,
                                         worker_init_fn=_worker_init_reset_seed if workers > 0 else None)
    return iter(loader)
