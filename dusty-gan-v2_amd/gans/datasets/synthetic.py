"""Synthetic real batches for benchmarking without the KITTI files:
depth ~ U(min_depth, max_depth), valid mask ~ Bernoulli(0.85), generated directly in HBM.
The reference's KITTI loader (gans/datasets/kitti.py) is out of scope for this round."""
import torch


class SyntheticRangeImages:
    def __init__(self, shape, min_depth, max_depth, batch_size, device, seed=0, keep_prob=0.85):
        self.shape, self.batch_size, self.device = tuple(shape), batch_size, device
        self.min_depth, self.max_depth, self.keep_prob = float(min_depth), float(max_depth), keep_prob
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)

    def __len__(self):
        return 1 << 30

    def __iter__(self):
        return self

    def __next__(self):
        H, W = self.shape
        B = self.batch_size
        depth = torch.rand(B, 1, H, W, device=self.device, generator=self.gen)
        depth = depth * (self.max_depth - self.min_depth) + self.min_depth
        mask = (torch.rand(B, 1, H, W, device=self.device, generator=self.gen) < self.keep_prob).float()
        return {"depth": depth, "mask": mask}
