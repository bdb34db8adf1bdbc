"""Normaliser of the plane datasets with the reference surface (libs/utilities3.py:76-132 NormalizerGivenMeanStd):
pointwise Gaussian normalisation with given statistics.  The device copies are made lazily (the reference calls
`.cuda()` in the constructor); `cuda_decode` is what trainer.FusedLpLoss fuses into the loss kernels."""
import numpy as np
import torch


class NormalizerGivenMeanStd(object):
    def __init__(self, mean, std, plane_indexs=None, eps=0.00001, time_last=True):
        if plane_indexs is not None:
            mean = mean[:, plane_indexs, :]
            std = std[:, plane_indexs, :]
        if np.sum(abs(np.asarray(mean) - eps)) < eps:
            raise RuntimeError("Provided mean is zero!")
        self.mean = torch.as_tensor(np.asarray(mean))
        self.std = torch.as_tensor(np.asarray(std))
        self.eps = eps
        self.time_last = time_last
        self._dev = {}

    def _on(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (self.mean.to(device), self.std.to(device))
        return self._dev[key]

    @property
    def mean_cuda(self):
        return self._on("cuda")[0]

    @property
    def std_cuda(self):
        return self._on("cuda")[1]

    def encode(self, x):
        return (x - self.mean) / (self.std + self.eps)

    def cuda_encode(self, x):
        mean, std = self._on(x.device)
        return (x - mean) / (std + self.eps)

    def decode(self, x, sample_idx=None):
        if sample_idx is not None:
            raise NotImplementedError("sample_idx masks are outside the observer training path")
        return x * (self.std + self.eps) + self.mean

    def cuda_decode(self, x, sample_idx=None):
        if sample_idx is not None:
            raise NotImplementedError("sample_idx masks are outside the observer training path")
        mean, std = self._on(x.device)
        return x * (std + self.eps) + mean
