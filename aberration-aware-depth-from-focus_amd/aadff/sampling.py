"""Uniform-sample sources for pupil / aperture sampling.

`HostSampler` draws from torch's CPU generator in the reference's call order (one
`torch.rand(n)` per reference call): with the same seed the HIP
path consumes exactly the samples the reference would, which is what the 1e-4 parity
budget requires (seed-to-seed Monte-Carlo noise is 1.4e-2).  `DeviceSampler` draws on
the GPU: statistically equivalent, not sample-for-sample comparable.
"""
import torch


class HostSampler:
    on_device = False

    def rand(self, n):
        return torch.rand(n)

    def rand_block(self, sizes):
        """Concatenation of torch.rand(s) for s in sizes, drawn call by call."""
        return torch.cat([torch.rand(s) for s in sizes])


class DeviceSampler:
    on_device = True

    def __init__(self, device, seed=None):
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device)
        if seed is not None:
            self.gen.manual_seed(seed)

    def rand(self, n):
        return torch.rand(n, device=self.device, generator=self.gen)

    def rand_block(self, sizes):
        return torch.rand(int(sum(sizes)), device=self.device, generator=self.gen)
