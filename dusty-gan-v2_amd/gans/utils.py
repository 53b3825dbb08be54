"""Training-path utilities (reference: gans/utils.py:21-42, 85-105, 238-271).  Visualisation helpers
of the reference file are out of scope."""
import os
import random

import numpy as np
import torch


def init_random_seed(random_seed=0, rank=0):
    seed = random_seed + rank
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def init_dist_process(rank, temp_dir, num_gpus, random_seed, backend=None):
    """One process per GPU; RCCL ("nccl" backend on ROCm) over xGMI, file:// rendezvous as in the
    reference (utils.py:33-42).  `backend="gloo"` is used by the CPU tests."""
    init_random_seed(random_seed, rank)
    init_method = f"file://{(temp_dir / '.torch_distributed_init').resolve()}"
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    torch.distributed.init_process_group(backend=backend, init_method=init_method, world_size=num_gpus, rank=rank)


def set_requires_grad(net, requires_grad: bool = True):
    for param in net.parameters():
        param.requires_grad = requires_grad


def sigmoid_to_tanh(x):
    """[0,1] -> [-1,+1]"""
    return x * 2.0 - 1.0


def tanh_to_sigmoid(x):
    """[-1,+1] -> [0,1]"""
    return (x + 1.0) / 2.0


class InfiniteSampler(torch.utils.data.Sampler):
    """Rank-sharded infinite shuffled index stream with a sliding re-shuffle window
    (reference: utils.py:238-271, from StyleGAN3)."""

    def __init__(self, dataset, rank=0, num_replicas=1, shuffle=True, seed=0, window_size=0.5):
        assert len(dataset) > 0 and num_replicas > 0 and 0 <= rank < num_replicas and 0 <= window_size <= 1
        self.dataset, self.rank, self.num_replicas = dataset, rank, num_replicas
        self.shuffle, self.seed, self.window_size = shuffle, seed, window_size

    def __iter__(self):
        order = np.arange(len(self.dataset))
        rnd, window = None, 0
        if self.shuffle:
            rnd = np.random.RandomState(self.seed)
            rnd.shuffle(order)
            window = int(np.rint(order.size * self.window_size))
        idx = 0
        while True:
            i = idx % order.size
            if idx % self.num_replicas == self.rank:
                yield order[i]
            if window >= 2:
                j = (i - rnd.randint(window)) % order.size
                order[i], order[j] = order[j], order[i]
            idx += 1
