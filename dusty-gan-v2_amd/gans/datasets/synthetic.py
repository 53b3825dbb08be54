"""Synthetic real batches for benchmarking without the KITTI files:
depth ~ U(min_depth, max_depth), valid mask ~ Bernoulli(0.85), generated directly in HBM.
The reference's KITTI loader (gans/datasets/kitti.py) is out of scope for this round."""
import torch


class SyntheticRangeImages:
    def __init__(self, shape, min_depth, max_depth, batch_size, device, seed=0, keep_prob=0.85, native_rng=False):
        """native_rng: both draws from ONE dgv2_rng_fill launch on the device's Philox stream (the trainer's: seeded from
        torch's seed, saved in checkpoints) instead of two generator launches and four elementwise ones."""
        self.shape, self.batch_size, self.device = tuple(shape), batch_size, device
        self.min_depth, self.max_depth, self.keep_prob = float(min_depth), float(max_depth), keep_prob
        self.native_rng = bool(native_rng) and torch.device(device).type == "cuda"
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)

    def __len__(self):
        return 1 << 30

    def __iter__(self):
        return self

    def __next__(self):
        H, W = self.shape
        B = self.batch_size
        if self.native_rng:
            from gans.models.ops import native
            depth, mask = native.rng_fill([((B, 1, H, W), native.RNG_UNIFORM, self.min_depth, self.max_depth),
                                           ((B, 1, H, W), native.RNG_BERNOULLI, self.keep_prob, 0.0)], self.device)
            return {"depth": depth, "mask": mask}
        depth = torch.rand(B, 1, H, W, device=self.device, generator=self.gen)
        depth = depth * (self.max_depth - self.min_depth) + self.min_depth
        mask = (torch.rand(B, 1, H, W, device=self.device, generator=self.gen) < self.keep_prob).float()
        return {"depth": depth, "mask": mask}
