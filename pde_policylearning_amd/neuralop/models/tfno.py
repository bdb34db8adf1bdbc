"""FNO / FNO2d / FNO3d with the reference constructor surface (neuralop/models/tfno.py:
Lifting :11-20, Projection :23-38, FNO :107-211, FNO2d :342-458, FNO3d :467-580).
forward() runs the whole model in the HIP engine (functional.fno_model)."""
import torch
import torch.nn.functional as TF
from torch import nn

from ... import functional as F
from .fno_block import FNOBlocks
from .spectral_convolution import SpectralConv, _unsupported


class Lifting(nn.Module):
    def __init__(self, in_channels, out_channels, n_dim=2):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.fc = getattr(nn, f'Conv{n_dim}d')(in_channels, out_channels, 1)

    def forward(self, x):
        return self.fc(x)


class Projection(nn.Module):
    def __init__(self, in_channels, out_channels, hidden_channels=None, n_dim=2, non_linearity=TF.gelu):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.hidden_channels = in_channels if hidden_channels is None else hidden_channels
        self.non_linearity = non_linearity
        Conv = getattr(nn, f'Conv{n_dim}d')
        self.fc1 = Conv(in_channels, hidden_channels, 1)
        self.fc2 = Conv(hidden_channels, out_channels, 1)

    def forward(self, x):
        return self.fc2(self.non_linearity(self.fc1(x)))


class FNO(nn.Module):
    def __init__(self, n_modes, hidden_channels, in_channels=3, out_channels=1, lifting_channels=256,
                 projection_channels=256, n_layers=4, output_scaling_factor=None,
                 incremental_n_modes=None, use_mlp=False, mlp_dropout=0, mlp_expansion=0.5,
                 non_linearity=TF.gelu, norm=None, preactivation=False, fno_skip='linear',
                 mlp_skip='soft-gating', separable=False, factorization=None, rank=1.0,
                 joint_factorization=False, fixed_rank_modes=False, implementation='factorized',
                 decomposition_kwargs=dict(), domain_padding=None, domain_padding_mode='one-sided',
                 fft_norm='forward', SpectralConv=SpectralConv, **kwargs):
        super().__init__()
        if domain_padding is not None and domain_padding > 0:
            _unsupported("domain_padding")
        self.n_dim = len(n_modes)
        self.n_modes = tuple(n_modes)
        self.hidden_channels = hidden_channels
        self.lifting_channels = lifting_channels        # stored, unused (tfno.py:191)
        self.projection_channels = projection_channels
        self.in_channels, self.out_channels, self.n_layers = in_channels, out_channels, n_layers
        self.fft_norm = fft_norm
        self.non_linearity = non_linearity
        self.fno_blocks = FNOBlocks(
            in_channels=hidden_channels, out_channels=hidden_channels, n_modes=self.n_modes,
            output_scaling_factor=output_scaling_factor, use_mlp=use_mlp, mlp_dropout=mlp_dropout,
            mlp_expansion=mlp_expansion, non_linearity=non_linearity, norm=norm,
            preactivation=preactivation, fno_skip=fno_skip, mlp_skip=mlp_skip,
            incremental_n_modes=incremental_n_modes, rank=rank, fft_norm=fft_norm,
            fixed_rank_modes=fixed_rank_modes, implementation=implementation, separable=separable,
            factorization=factorization, decomposition_kwargs=decomposition_kwargs,
            joint_factorization=joint_factorization, SpectralConv=SpectralConv, n_layers=n_layers)
        self.lifting = Lifting(in_channels, hidden_channels, n_dim=self.n_dim)
        self.projection = Projection(hidden_channels, out_channels, hidden_channels=projection_channels,
                                     non_linearity=non_linearity, n_dim=self.n_dim)

    def engine_args(self):
        blk = self.fno_blocks
        L = self.n_layers
        skip_ws = [blk.fno_skips[l].weight for l in range(L)]
        spec_ws = [w for l in range(L) for w in blk.convs.layer_weights(l)]    # sliced under incremental_n_modes
        return dict(lift_w=self.lifting.fc.weight, lift_b=self.lifting.fc.bias, skip_ws=skip_ws,
                    spec_ws=spec_ws, spec_bias=blk.convs.bias,
                    w1=self.projection.fc1.weight, b1=self.projection.fc1.bias,
                    w2=self.projection.fc2.weight, b2=self.projection.fc2.bias,
                    modes=blk.convs.half_n_modes, norm=self.fft_norm)

    def fused_supported(self, x):
        """Shapes the whole-model fused kernels cover (fno_model_plan_create): hidden width 32/64,
        <= 4 input / output channels, projection_channels 256, and either rows that tile the 128 / 256-pixel workgroup tile
        (last dim 32, 64, 128, 256) or "loose rows" (any last dim in 32..320 on planes that tile by 128 pixels: 96 x 96,
        160 x 160, 192 x 192 grids ...; split-precision GEMM mode)."""
        if self.fno_blocks.convs.separable or self.fno_blocks.convs.output_scaling_factor is not None:
            return False           # torch compositions (SpectralConv._torch_composition): layer by layer
        w = x.shape[-1]
        npx = 256 if w > 128 else 128
        plane = 1
        for s in x.shape[2:]:
            plane *= s
        tiled = w % 32 == 0 and w <= 256 and npx % w == 0 and plane % npx == 0
        loose = (not tiled) and 32 <= w <= 320 and plane % 128 == 0 and F._lib.lib().fno_get_gemm_mode() == 1
        if not (self.hidden_channels in (32, 64) and self.in_channels <= 4 and self.out_channels <= 4
                and self.projection_channels == 256 and (tiled or loose) and x.is_cuda
                and (not x.requires_grad or self.in_channels <= 4)):
            return False
        # the engine has the last word (e.g. 256-pixel tiles with many kept modes do not fit LDS)
        gelu_mask = 0
        for l in range(self.n_layers):
            if l < self.n_layers - l:
                gelu_mask |= 1 << l
        return F.model_plan_available(self.n_dim, self.in_channels, self.hidden_channels, self.out_channels,
                                      self.projection_channels, self.n_layers, tuple(x.shape[2:]),
                                      tuple(m // 2 for m in self.n_modes), self.fft_norm, gelu_mask, x.device)

    def forward(self, x):
        sliced = self.fno_blocks.convs.incremental_n_modes is not None
        if self.fused_supported(x) and not (sliced and getattr(self, "_direct_grads", False)):
            # direct_grads: set by trainer.FlatGradBucket(model, direct=True); the engine then writes
            # parameter gradients straight into the flat bucket (one use of each parameter per step; sliced weights
            # under incremental_n_modes are copies, so that combination takes the composition below)
            return F.fno_model(x, direct_grads=getattr(self, "_direct_grads", False),
                               overlap=getattr(self, "_grad_overlap", None), **self.engine_args())
        # other widths / grids: spectral convolutions on the engine, pointwise glue in torch
        if getattr(self, "_direct_grads", False) and torch.is_grad_enabled():
            # a direct-write bucket does not clear this model's gradients; autograd accumulates on this path, so clear them here
            for p in self.parameters():
                if p.grad is not None:
                    p.grad.zero_()
        x = self.lifting(x)
        for l in range(self.n_layers):
            x = self.fno_blocks(x, l)
        return self.projection(x)


class FNO2d(FNO):
    def __init__(self, n_modes_height, n_modes_width, hidden_channels, in_channels=3, out_channels=1,
                 lifting_channels=256, projection_channels=256, n_layers=4, output_scaling_factor=None,
                 incremental_n_modes=None, non_linearity=TF.gelu, use_mlp=False, mlp_dropout=0,
                 mlp_expansion=0.5, norm=None, skip='soft-gating', separable=False, preactivation=False,
                 factorization=None, rank=1.0, joint_factorization=False, fixed_rank_modes=False,
                 implementation='factorized', decomposition_kwargs=dict(), domain_padding=None,
                 domain_padding_mode='one-sided', fft_norm='forward', **kwargs):
        # `skip` is accepted and ignored exactly as in the reference (tfno.py:449 -> **kwargs :131)
        super().__init__(
            n_modes=(n_modes_height, n_modes_width), hidden_channels=hidden_channels,
            in_channels=in_channels, out_channels=out_channels, lifting_channels=lifting_channels,
            projection_channels=projection_channels, n_layers=n_layers, output_scaling_factor=None,
            non_linearity=non_linearity, use_mlp=use_mlp, mlp_dropout=mlp_dropout,
            mlp_expansion=mlp_expansion, incremental_n_modes=incremental_n_modes, norm=norm,
            separable=separable, preactivation=preactivation, factorization=factorization, rank=rank,
            joint_factorization=joint_factorization, fixed_rank_modes=fixed_rank_modes,
            implementation=implementation, decomposition_kwargs=decomposition_kwargs,
            domain_padding=domain_padding, domain_padding_mode=domain_padding_mode, fft_norm=fft_norm)
        self.n_modes_height, self.n_modes_width = n_modes_height, n_modes_width


class FNO3d(FNO):
    def __init__(self, n_modes_height, n_modes_width, n_modes_depth, hidden_channels, in_channels=3,
                 out_channels=1, lifting_channels=256, projection_channels=256, n_layers=4,
                 output_scaling_factor=None, incremental_n_modes=None, non_linearity=TF.gelu,
                 use_mlp=False, mlp_dropout=0, mlp_expansion=0.5, norm=None, skip='soft-gating',
                 separable=False, preactivation=False, factorization=None, rank=1.0,
                 joint_factorization=False, fixed_rank_modes=False, implementation='factorized',
                 decomposition_kwargs=dict(), domain_padding=None, domain_padding_mode='one-sided',
                 fft_norm='forward', **kwargs):
        super().__init__(
            n_modes=(n_modes_height, n_modes_width, n_modes_depth), hidden_channels=hidden_channels,
            in_channels=in_channels, out_channels=out_channels, lifting_channels=lifting_channels,
            projection_channels=projection_channels, n_layers=n_layers, output_scaling_factor=None,
            non_linearity=non_linearity, use_mlp=use_mlp, mlp_dropout=mlp_dropout,
            mlp_expansion=mlp_expansion, incremental_n_modes=incremental_n_modes, norm=norm,
            separable=separable, preactivation=preactivation, factorization=factorization, rank=rank,
            joint_factorization=joint_factorization, fixed_rank_modes=fixed_rank_modes,
            implementation=implementation, decomposition_kwargs=decomposition_kwargs,
            domain_padding=domain_padding, domain_padding_mode=domain_padding_mode, fft_norm=fft_norm)
        self.n_modes_height, self.n_modes_width, self.n_modes_depth = n_modes_height, n_modes_width, n_modes_depth


class FNO1d(FNO):
    """Name kept for `from neuralop.models import FNO1d` (tfno.py:222-340).  One-dimensional grids are not on the
    accelerated path (configs 1-5 are 2-D / 3-D): constructing one raises."""

    def __init__(self, n_modes_height, hidden_channels, in_channels=3, out_channels=1, lifting_channels=256,
                 projection_channels=256, incremental_n_modes=None, n_layers=4, output_scaling_factor=None,
                 non_linearity=TF.gelu, use_mlp=False, mlp_dropout=0, mlp_expansion=0.5, norm=None,
                 skip='soft-gating', separable=False, preactivation=False, factorization=None, rank=1.0,
                 joint_factorization=False, fixed_rank_modes=False, implementation='factorized',
                 decomposition_kwargs=dict(), domain_padding=None, domain_padding_mode='one-sided',
                 fft_norm='forward', **kwargs):
        _unsupported("FNO1d (one-dimensional grids)")


def _with_defaults(new_name, cls, **defaults):
    """A subclass of `cls` whose constructor has other default values (the role of tfno.py:594-615)."""
    def __init__(self, *args, **kwargs):
        for k, v in defaults.items():
            kwargs.setdefault(k, v)
        cls.__init__(self, *args, **kwargs)
    return type(new_name, (cls,), {"__init__": __init__, "__doc__": cls.__doc__})


# Tucker-factorised and spherical variants (tfno.py:619-624): their weights live in tltorch / torch_harmonics containers
# that this image does not have, so the constructors raise (SpectralConv refuses the factorization; SFNO needs the SHT).
TFNO = _with_defaults('TFNO', FNO, factorization='Tucker')
TFNO1d = _with_defaults('TFNO1d', FNO1d, factorization='Tucker')
TFNO2d = _with_defaults('TFNO2d', FNO2d, factorization='Tucker')
TFNO3d = _with_defaults('TFNO3d', FNO3d, factorization='Tucker')


class SFNO(FNO):
    def __init__(self, *args, **kwargs):
        _unsupported("SFNO (spherical harmonic convolution, torch_harmonics)")
