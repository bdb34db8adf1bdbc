"""`UNO` keeps its name for `from neuralop.models import UNO` (reference neuralop/models/uno.py:15-139); the U-shaped
operator is not instantiated by any of the accelerated configurations, so constructing it raises."""
import torch.nn.functional as TF
from torch import nn

from .spectral_convolution import _unsupported


class UNO(nn.Module):
    def __init__(self, in_channels, out_channels, hidden_channels, lifting_channels=256, projection_channels=256,
                 n_layers=4, uno_out_channels=None, uno_n_modes=None, uno_scalings=None, horizontal_skips_map=None,
                 incremental_n_modes=None, use_mlp=False, mlp_dropout=0, mlp_expansion=0.5, non_linearity=TF.gelu,
                 norm=None, preactivation=False, fno_skip='linear', horizontal_skip='linear', mlp_skip='soft-gating',
                 separable=False, factorization=None, rank=1.0, joint_factorization=False, fixed_rank_modes=False,
                 implementation='factorized', decomposition_kwargs=dict(), domain_padding=None,
                 domain_padding_mode='one-sided', fft_norm='forward', **kwargs):
        super().__init__()
        _unsupported("UNO")
