"""PUNetG score network on HIP kernels.

Same constructor, forward protocol ``net(x, t=None, y=None)`` and state_dict key names as the
reference (diffsci/models/nets/punetg.py:80-106,356-416), so reference checkpoints load with
``load_state_dict``.  The torch.nn layers created here are *parameter containers only* -- they
give the reference's key names and default initialisers -- and are never called: every tensor
operation of the forward pass is a launch into libdiffsci_hip.so:

  convin / convout / conv1+time-shift / conv2+residual   ds_conv2d (fp32 MFMA implicit GEMM)
  DownSampler (max-pool -> conv), UpSampler (nearest -> conv) + skip add   ds_conv2d load modes
  GroupNorm(C,C)+SiLU, GroupRMSNorm(C,C)+SiLU              ds_inorm_silu
  GaussianFourierProjection, ResnetTimeBlock MLPs          ds_fourier_features, ds_linear
  TwoDimensionalAttention (nn.MultiheadAttention, 1 head)  ds_conv2d (1x1 projections) + ds_attention
  x + xa (punetg.py:385)                                    folded into the preceding conv epilogue
"""
import math
import os
from typing import Any

import torch

from ... import ops
from ..._native import DS_LOAD_MAXPOOL2, DS_LOAD_UPSAMPLE2
from . import precision
from .punetg_config import PUNetGConfig


class _AffineHolder(torch.nn.Module):
    """weight/bias container for GroupRMSNorm / GroupPixNorm (commonlayers.py:332-361, 387-414); no parameters
    when affine=False."""

    def __init__(self, C, affine=True):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(C)) if affine else None
        self.bias = torch.nn.Parameter(torch.zeros(C)) if affine else None


NORM_KINDS = {"GroupLN": 0, "GroupRMS": 1, "GroupPix": 3}       # anything else: Identity (kind 2), commonlayers.py:882-899


def make_norm(name, C, affine=True):
    """ResnetBlockC.get_normalization_functions (commonlayers.py:882-899) with num_groups = C."""
    if name == "GroupLN":
        return torch.nn.GroupNorm(C, C, affine=affine)
    if name in ("GroupRMS", "GroupPix"):
        return _AffineHolder(C, affine)
    return torch.nn.Identity()


def mp_weight(w):
    """Effective weight of the magnitude-preserving layers in eval mode (normedlayers.py:17-22,46-55,95-99):
    normalize(w) / sqrt(fan_in) with normalize(x) = x / (eps + ||x_row|| * sqrt(1/fan_in)), eps = 1e-4."""
    fan_in = w[0].numel()
    n = torch.linalg.vector_norm(w, dim=list(range(1, w.ndim)), keepdim=True)
    alpha = math.sqrt(n.numel() / w.numel())
    return (w / torch.add(1e-4, n, alpha=alpha)) / math.sqrt(fan_in)


class _MPConv(torch.nn.Module):
    """MagnitudePreservingConv2d parameters (normedlayers.py:26-44): N(0,1) weight, zero bias."""
    mp = True

    def __init__(self, cin, cout, k, bias=True, dim=2):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(cout, cin, *([k] * dim)))
        self.bias = torch.nn.Parameter(torch.zeros(cout)) if bias else None
        self.in_channels, self.out_channels = cin, cout


class _MPLinear(torch.nn.Module):
    """MagnitudePreservingLinear parameters (normedlayers.py:6-15)."""
    mp = True

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(cout, cin))
        self.bias = torch.nn.Parameter(torch.zeros(cout))


class _InHouseAttention(torch.nn.Module):
    """The reference's own MultiHeadAttention (attention.py:110-153), used instead of nn.MultiheadAttention when the
    network is magnitude preserving or attn_type == "cosine" (attention.py:29-52): one head, dk = dv = dmodel, no
    biases; N(0,1) weights renormalised on the fly when magnitude preserving, Xavier-uniform otherwise."""

    def __init__(self, C, magnitude_preserving, cosine):
        super().__init__()
        for n in ("q", "k", "v", "o"):
            w = torch.empty(1, C, C)
            (torch.nn.init.normal_ if magnitude_preserving else torch.nn.init.xavier_uniform_)(w)
            setattr(self, n + "_proj_matrix", torch.nn.Parameter(w))
        self.embed_dim = C
        self.magnitude_preserving, self.cosine = magnitude_preserving, cosine

    def _normalized(self, weight, kind):
        """MultiHeadAttention.normalize_weight (magnitude preserving only) + the unconditional 1/sqrt(fan_in) of
        forward (attention.py:183-196, 232-247)."""
        fan_in = weight.shape[0] * weight.shape[2] if kind == "wo" else weight.shape[1]
        if self.magnitude_preserving:
            norm = torch.linalg.vector_norm(weight, dim=[0, 2] if kind == "wo" else 1, keepdim=True)
            alpha = math.sqrt(norm.numel() / weight.numel())
            weight = weight / (alpha * norm + 1e-4)
        return weight / math.sqrt(fan_in)

    def projection_weights(self):
        """(in_proj [3E, E], out_proj [E, E]) as 1x1-convolution weights: q = x Wq -> rows of Wq^T;
        out[l] = sum_k a[k] wo[0, l, k]."""
        wq, wk, wv = (self._normalized(getattr(self, n + "_proj_matrix").detach(), "w" + n)[0].t() for n in "qkv")
        wo = self._normalized(self.o_proj_matrix.detach(), "wo")[0]
        return torch.cat([wq, wk, wv], dim=0).contiguous(), wo.contiguous()


class _TimeBlock(torch.nn.Module):
    """ResnetTimeBlock parameters: net.{0,2,4} Linear (commonlayers.py:512-522), magnitude-preserving
    linears when the convolutions are."""

    def __init__(self, embed, out, mp=False):
        super().__init__()
        lin = (lambda i, o: _MPLinear(i, o)) if mp else torch.nn.Linear
        self.net = torch.nn.Sequential(
            lin(embed, 4 * embed), torch.nn.Identity(),
            lin(4 * embed, 4 * embed), torch.nn.Identity(),
            lin(4 * embed, out))


class _CircConv(torch.nn.Module):
    """CircularConv2d / CircularConv3d parameters (commonlayers.py:918-1040): the weights live one level down, in .conv."""

    def __init__(self, cin, cout, k, bias=True, dim=2):
        super().__init__()
        self.conv = (torch.nn.Conv3d if dim == 3 else torch.nn.Conv2d)(cin, cout, k, bias=bias)
        self.in_channels, self.out_channels = cin, cout

    @property
    def weight(self):
        return self.conv.weight

    @property
    def bias(self):
        return self.conv.bias


def make_conv(cin, cout, k, kind="default", bias=True, dim=2):
    """choose_conv_cls (punetg.py:217-236): kind = convolution_type ("default" | "circular" | "mp"), dim 2 or 3."""
    if kind is True or kind == "circular":
        return _CircConv(cin, cout, k, bias, dim)
    if kind == "mp":
        return _MPConv(cin, cout, k, bias, dim)
    return (torch.nn.Conv3d if dim == 3 else torch.nn.Conv2d)(cin, cout, k, padding="same", bias=bias)


class _ResBlock(torch.nn.Module):
    """ResnetBlockC parameters (commonlayers.py:766-807)."""

    def __init__(self, C, embed, conv_kind="default", bias=True, norms=("GroupLN", "GroupRMS"), affine=True, dim=2, k=3):
        super().__init__()
        self.gnorm1 = make_norm(norms[0], C, affine)
        self.gnorm2 = make_norm(norms[1], C, affine)
        self.conv1 = make_conv(C, C, k, conv_kind, bias, dim)
        self.conv2 = make_conv(C, C, k, conv_kind, bias, dim)
        self.timeblock = _TimeBlock(embed, C, mp=conv_kind == "mp")


class _Sampler(torch.nn.Module):
    def __init__(self, cin, cout, conv_kind="default", bias=True, dim=2, k=3):
        super().__init__()
        self.conv = make_conv(cin, cout, k, conv_kind, bias, dim)


class _Attn(torch.nn.Module):
    def __init__(self, C, mp=False, cosine=False):
        super().__init__()
        self.mhattn = (_InHouseAttention(C, mp, cosine) if (mp or cosine)
                       else torch.nn.MultiheadAttention(C, num_heads=1, batch_first=True))


class _Fourier(torch.nn.Module):
    def __init__(self, embed_dim, scale):
        super().__init__()
        self.register_buffer("W", torch.randn(embed_dim // 2) * scale)


class _FourierInput(torch.nn.Module):
    """ConvolutionalFourierProjection buffers (commonlayers.py:229-244), bias=False: W [input_dim, embed_dim/2]."""

    def __init__(self, input_dim, embed_dim, scale):
        super().__init__()
        self.register_buffer("W", torch.randn(input_dim, embed_dim // 2) * scale)
        self.in_channels, self.out_channels = input_dim, embed_dim


class _ConditionDrop(torch.nn.Module):
    """ConditionDrop parameters (commonlayers.py:1100-1127): identity outside training; the null embedding is only a
    state_dict entry here."""

    def __init__(self, p, hidden_dim, null_is_learnable=True):
        super().__init__()
        self.p = p
        if null_is_learnable:
            self.null_embedding = torch.nn.Parameter(torch.randn(1, hidden_dim))
        else:
            self.register_buffer("null_embedding", torch.zeros(1, hidden_dim))


def require_eval(net, *rates):
    """Dropout, condition dropout and ConditionDrop are the identity in eval mode -- the only mode the sampling path
    implements.  A network left in training mode with a non-zero rate would silently differ from the reference."""
    if net.training and any(r for r in rates if r):
        raise NotImplementedError("dropout / cond_dropout / cond_drop > 0 in training mode are outside the HIP sampling "
                                  "path: call .eval() (the reference samples under eval() too)")


class _Workspace:
    """Shape-keyed pool of device buffers.  A forward pass takes and gives buffers in a fixed
    order, so after the first pass no allocation happens -- a requirement for hipGraph capture."""

    def __init__(self):
        self.free = {}
        self.frozen = False
        self.bytes = 0

    def take(self, shape, device):
        key = (tuple(shape), str(device))
        lst = self.free.get(key)
        if lst:
            return lst.pop()
        if self.frozen:
            raise RuntimeError(f"workspace is frozen (graph captured) but a new buffer {shape} was requested")
        with torch.inference_mode(False):    # a normal tensor even when the sampler runs under inference_mode: the pool
            t = torch.empty(shape, dtype=torch.float32, device=device)   # outlives the call and serves eager forwards too
        self.bytes += t.numel() * 4
        return t

    def give(self, t):
        self.free.setdefault((tuple(t.shape), str(t.device)), []).append(t)


class _AmaxArena:
    """Per-forward rows of "amax" slots (ops.py: per-sample max |x| as float bits, the activation exponents of the fp16x3
    kernels' raw-input launches), taken from the workspace and zeroed by ONE fill launch; producers' epilogues merge into a
    row (out_amax), the raw-input consumer reads it (in_amax)."""
    ROWS = 256

    def __init__(self, ws, B, dev, zero=True):
        self.ws, self.buf = ws, ws.take((self.ROWS, max(B, 1)), dev)
        self.i32 = self.buf.view(torch.int32)
        if zero:                                  # zero=False: the caller's first act is of_input(), which zeroes the arena itself
            ops.amax_zero(self.i32)
        self.zeroed = zero
        self.n = 0

    def row(self):
        if not self.zeroed:
            ops.amax_zero(self.i32)
            self.zeroed = True
        if self.n >= self.ROWS:
            raise RuntimeError("amax arena exhausted")
        self.n += 1
        return self.i32[self.n - 1]

    def rows(self, n):
        """n consecutive rows as one [n * B] tensor."""
        if not self.zeroed:
            ops.amax_zero(self.i32)
            self.zeroed = True
        if self.n + n > self.ROWS:
            raise RuntimeError("amax arena exhausted")
        self.n += n
        return self.i32[self.n - n:self.n].view(-1)

    def of(self, x, rows=None):
        """Slots filled by a reduction over x (a tensor no epilogue of ours produced)."""
        return ops.absmax_rows(x, rows, out=self.row())

    def of_input(self, x, flag, wmax):
        """The same for a network input x [B, C, ...] (c_in * x next to raw user fields): per-channel maxima first, `flag` raised
        when one exponent per sample cannot serve the input layer given its weights (ops.absmax_channels; precision.input_layer_flag)."""
        C = x.shape[1]
        if not self.zeroed:
            if self.n == 0 and C <= 64 and x[0].numel() <= ops.INPUT_AMAX_MAX_FLOATS and x.shape[0] == self.i32.shape[1]:
                self.n, self.zeroed = 1, True
                return ops.input_amax(self.i32, 0, x, flag, wmax)      # one launch: zero the arena, reduce, apply the channel criterion
            ops.amax_zero(self.i32)
            self.zeroed = True
        if self.n + C + 1 > self.ROWS:
            return self.of(x)
        out = self.row()
        scratch = self.i32[self.n:self.n + C].view(-1)
        self.n += C
        return ops.absmax_channels(x, out, scratch, flag, wmax)

    def release(self):
        self.ws.give(self.buf)


class _FieldShifts:
    """Per-pixel time shifts, computed where they are used.  A field-valued conditional embedding makes the time embedding a
    field te [B, C, He, We] (punetg.py:405-410) and every block's ResnetTimeBlock a per-pixel MLP (commonlayers.py:537-546)
    whose result the block brings to its own resolution by taking the top-left corner of every window (rescale_yt,
    commonlayers.py:838-869).  The MLP is pointwise, so corner-pooling its INPUT gives the same values: each block evaluates
    its three 1x1 convolutions at its own resolution (16x fewer pixels two levels down) from a pooled copy of te that the
    blocks of a level share.  Every buffer comes from the network's workspace, so the evaluation can sit inside a captured run."""

    def __init__(self, te, ws, h3, owned=False):
        self.te, self.ws, self.te_owned = te, ws, owned
        self.am = _AmaxArena(ws, te.shape[0], te.device) if h3 else None
        self.levels = {}

    def level(self, H, W):
        """te at a block's resolution [B, C, H, W] and its amax row (fp16x3)."""
        got = self.levels.get((H, W))
        if got is None:
            B, C, h, w = self.te.shape
            if (h, w) == (H, W):
                t, owned = self.te, False
            elif h > H:
                f = h // H
                if H * f != h or W * f != w:
                    raise ValueError(f"yt_dims {(h, w)} and y_dims {(H, W)} are not compatible")
                t, owned = self.ws.take((B, C, H, W), self.te.device), True
                t.copy_(self.te[:, :, ::f, ::f])                      # CornerPool2d(f): the top-left corner of every window
            else:
                raise NotImplementedError("a conditional-embedding field coarser than a block's resolution (the reference's "
                                          "upscaling branch passes the factor as torch.nn.Upsample's size and fails as well)")
            got = self.levels[(H, W)] = (t, self.am.of(t) if self.am is not None else None, owned)
        return got[0], got[1]

    def release(self):
        for t, _, owned in self.levels.values():
            if owned:
                self.ws.give(t)
        self.levels = {}
        if self.am is not None:
            self.am.release()
            self.am = None
        if self.te_owned:
            self.ws.give(self.te)
            self.te_owned = False


class PUNetG(torch.nn.Module):
    def __init__(self,
                 config: PUNetGConfig,
                 conditional_embedding: torch.nn.Module | None = None,
                 extra_residual: torch.nn.Module | None = None):
        super().__init__()
        why = config.unsupported_reason()
        if why:
            raise NotImplementedError(why)
        # extra_residual (punetg.py:83-92,249-261; commonlayers.py:831-833): ONE user module shared by every residual
        # block, y = (conv2(...) + x) + extra_residual(x).  It is ordinary torch code run as given (it may allocate), so a
        # network that carries one is evaluated launch by launch instead of from a captured graph.
        self.extra_residual = extra_residual
        self.capturable = extra_residual is None
        self.config = config
        mc = config.model_channels
        mult = config.extended_channel_expansion
        self.time_projection = _Fourier(mc, config.time_projection_scale)
        self.conditional_embedding = conditional_embedding
        self.cond_drop = (_ConditionDrop(config.cond_drop, mc, config.cond_drop_learnable)
                          if config.cond_drop is not None and config.cond_drop > 0 else None)     # punetg.py:102-106
        self.circular = config.convolution_type == "circular"
        circ = config.convolution_type                     # conv kind: "default" | "circular" | "mp"
        self.mp = config.convolution_type == "mp"
        self.cosine_attn = config.attn_type == "cosine"
        self.inhouse_attn = self.mp or self.cosine_attn        # attention.py:29-52
        hb = bool(config.bias)
        norms = (config.first_resblock_norm, config.second_resblock_norm)
        self.norm_kinds = tuple(NORM_KINDS.get(n, 2) for n in norms)
        # bias=False: no convolution biases; a constant-one input channel is appended instead (punetg.py:190-191,390-394)
        dim = self.dim = config.dimension
        # the reference's construction sequence and builder names (punetg.py:94-106); the builders return parameter CONTAINERS
        # with the reference's state_dict keys -- the tensor work is in forward_with_shifts
        self.convin, self.convout = self.make_convin_and_convout()
        self.downward_blocks, self.downsamplers = self.make_downward_blocks()
        self.upward_blocks, self.upsamplers = self.make_upward_blocks()
        self.before_block, self.after_block = self.make_non_attn_bottom_blocks()
        self.attn_resnet_block, self.attn_block = self.make_attn_bottom_blocks()
        # Arithmetic of the 3x3 convolutions -- all three give fp32-level error (tests/test_gpu_kernels.py):
        #   "fp16x3": fp16 hi+lo split, 3 MFMA products (default; inputs must stay below 65504 in magnitude)
        #   "bf16x6": exact 3-way bf16 split, 6 MFMA products (no range limit, half the speed)
        #   "fp32"  : exact-fp32 MFMA (1/16 of the 16-bit rate)
        self.conv_precision = "fp16x3"
        # fp16x3 only: a non-finite output from finite inputs means an activation left fp16's range; switch to the
        # range-free "bf16x6" once and recompute (nets/precision.py) instead of handing the user NaNs
        self.auto_precision = True
        # With the fp16x3 kernels the two norms of a residual block are folded into the convolutions
        # around them (statistics from the producer's epilogue, normalise + SiLU in the consumer's
        # loader): the normalised tensors never touch HBM.  False: standalone ds_inorm_silu kernels.
        self.fuse_norm = True
        # ... but only where one workgroup column covers all output channels: with Cout/64 > fuse_max_cot channel
        # tiles every tile's workgroup would redo the activation of the same input patch (5.3x the transcendental
        # work of a standalone pass at Cout = 256), and the standalone kernel wins
        self.fuse_max_cot = 2
        # Standalone norms hand their convolutions pre-split fp16 images (ops.inorm_silu_images / ops.conv_img) where the layer
        # qualifies (_norm_images_ok); DIFFSCI_NORM_IMAGES=0 keeps the fp32 route (A/B runs)
        self.norm_images = os.environ.get("DIFFSCI_NORM_IMAGES", "1") != "0"
        self._packed = None
        self._packed_sig = None
        self._ws = _Workspace()
        self._am = None              # the amax arena of the forward pass in flight
        self._window_cache = {}
        # set by precision.escalate_input when the input's channels differ by more than 2^14 in magnitude within a sample
        self.exact_input_layer = False

    # ------------------------------------------------------------------ builders (punetg.py:122-334: same names and arguments)
    def choose_conv_cls(self):
        """punetg.py:217-236: the convolution constructor of this configuration -- here a callable
        (in_channels, out_channels, kernel_size, bias=True) that builds the parameter container of that convolution type."""
        if self.config.dimension not in (2, 3):
            raise NotImplementedError("1D convolution not implemented yet") if self.config.dimension == 1 \
                else ValueError(f"Invalid dimension {self.config.dimension}")
        kind, dim = self.config.convolution_type, self.config.dimension

        def conv_cls(in_channels, out_channels, kernel_size, bias=True, **_):
            return make_conv(in_channels, out_channels, kernel_size, kind, bias, dim)
        return conv_cls

    def make_convin_and_convout(self):
        """punetg.py:188-215.  bias=False: no convolution biases; a constant-one input channel is appended instead."""
        c = self.config
        conv_cls = self.choose_conv_cls()
        cin = c.input_channels + (0 if c.bias else 1)
        if c.in_embedding:                               # fixed Fourier input embedding instead of a convolution, punetg.py:194-202
            convin = _FourierInput(cin, c.model_channels, c.input_projection_scale)
        else:
            convin = conv_cls(cin, c.model_channels, c.in_out_kernel_size, bias=bool(c.bias))
        convout = conv_cls(c.model_channels, c.output_channels, c.in_out_kernel_size, bias=bool(c.bias))
        return convin, convout

    def resnet_fn(self, input_multiplier: int):
        """punetg.py:238-261: one ResnetBlockC's parameters."""
        c = self.config
        blk = _ResBlock(input_multiplier * c.model_channels, c.model_channels, c.convolution_type, bool(c.bias),
                        (c.first_resblock_norm, c.second_resblock_norm), bool(c.affine_norm), c.dimension, c.kernel_size)
        if self.extra_residual is not None:
            blk.extra_residual = self.extra_residual        # the reference registers the shared module in every block
        return blk

    def resnet_block_fn(self, input_multiplier: int, number_resnet_per_block: int):
        return torch.nn.ModuleList([self.resnet_fn(input_multiplier) for _ in range(number_resnet_per_block)])

    def attn_fn(self, input_multiplier: int):
        """punetg.py:272-289."""
        if self.config.dimension not in (2, 3):
            raise NotImplementedError("1D attention not implemented yet") if self.config.dimension == 1 \
                else ValueError(f"Invalid dimension {self.config.dimension}")
        return _Attn(input_multiplier * self.config.model_channels, self.config.magnitude_preserving,
                     self.config.attn_type == "cosine")

    def attn_block_fn(self, input_multiplier: int, number_resnet_attn_block: int):
        return torch.nn.ModuleList([self.attn_fn(input_multiplier) for _ in range(number_resnet_attn_block - 1)])

    def downsampler_fn(self, input_multiplier: int, output_multiplier: int):
        """punetg.py:300-316: DownSampler (max-pool, then convolution)."""
        c = self.config
        return _Sampler(input_multiplier * c.model_channels, output_multiplier * c.model_channels, c.convolution_type,
                        bool(c.bias), c.dimension, c.transition_kernel_size)

    def upsampler_fn(self, input_multiplier: int, output_multiplier: int):
        """punetg.py:318-334: UpSampler (nearest upsampling, then convolution): the same parameters as a DownSampler."""
        return self.downsampler_fn(input_multiplier, output_multiplier)

    def make_downward_blocks(self):
        mult = self.config.extended_channel_expansion
        blocks, samplers = torch.nn.ModuleList(), torch.nn.ModuleList()
        for i, m in enumerate(mult[:-1]):
            blocks.append(self.resnet_block_fn(m, self.config.number_resnet_downward_block))
            samplers.append(self.downsampler_fn(m, mult[i + 1]))
        return blocks, samplers

    def make_upward_blocks(self):
        rmult = list(reversed(self.config.extended_channel_expansion))
        blocks, samplers = torch.nn.ModuleList(), torch.nn.ModuleList()
        for i, m in enumerate(rmult[:-1]):
            samplers.append(self.upsampler_fn(m, rmult[i + 1]))
            blocks.append(self.resnet_block_fn(rmult[i + 1], self.config.number_resnet_upward_block))
        return blocks, samplers

    def make_non_attn_bottom_blocks(self):
        m = self.config.extended_channel_expansion[-1]
        return (self.resnet_block_fn(m, self.config.number_resnet_before_attn_block),
                self.resnet_block_fn(m, self.config.number_resnet_after_attn_block))

    def make_attn_bottom_blocks(self):
        m, n = self.config.extended_channel_expansion[-1], self.config.number_resnet_attn_block
        return self.resnet_block_fn(m, n), self.attn_block_fn(m, n)

    def calculate_receptive_field(self) -> dict:
        """punetg.py:423-628: theoretical receptive field of the network in input pixels.  A convolution of size k at cumulative
        stride s widens it by (k - 1) s, a residual block by twice that, a max-pool of size p by (p - 1) s before multiplying
        the stride by p; nearest upsampling only divides the stride; global attention makes it infinite."""
        c = self.config
        trace = []
        n_attn = c.number_resnet_attn_block - 1
        if n_attn > 0:
            trace.append(f"{n_attn} global attention layer(s): every output pixel sees every input pixel")
            return {'rf': float('inf'), 'has_attention': True, 'num_attention_layers': n_attn, 'trace': trace,
                    'feasible_chunking': False,
                    'config_summary': {'number_resnet_attn_block': c.number_resnet_attn_block, 'kernel_size': c.kernel_size,
                                       'in_out_kernel_size': c.in_out_kernel_size, 'channel_expansion': c.channel_expansion}}
        rf, stride = 1, 1

        def widen(k, name, times=1):
            nonlocal rf
            rf += times * (k - 1) * stride
            trace.append(f"{name}: {times} x (k = {k}) at stride {stride} -> {rf}")
        if c.in_embedding:
            trace.append("convin is a per-pixel Fourier embedding: unchanged")
        else:
            widen(c.in_out_kernel_size, "convin")
        levels = len(c.channel_expansion)
        for lv in range(levels):
            for b in range(c.number_resnet_downward_block):
                widen(c.kernel_size, f"down[{lv}].resnet[{b}]", 2)
            widen(c.transition_scale_factor, f"down[{lv}].maxpool")
            stride *= c.transition_scale_factor
            widen(c.transition_kernel_size, f"down[{lv}].conv")
        for name, n in (("before_block", c.number_resnet_before_attn_block), ("attn_resnet_block", c.number_resnet_attn_block),
                        ("after_block", c.number_resnet_after_attn_block)):
            for b in range(n):
                widen(c.kernel_size, f"{name}[{b}]", 2)
        for lv in range(levels - 1, -1, -1):
            stride //= c.transition_scale_factor
            widen(c.transition_kernel_size, f"up[{lv}].conv")
            for b in range(c.number_resnet_upward_block):
                widen(c.kernel_size, f"up[{lv}].resnet[{b}]", 2)
        widen(c.in_out_kernel_size, "convout")
        return {'rf': rf, 'has_attention': False, 'num_attention_layers': 0, 'trace': trace, 'feasible_chunking': True,
                'downsampling_factor': c.transition_scale_factor ** levels,
                'config_summary': {k: getattr(c, k) for k in (
                    'number_resnet_attn_block', 'number_resnet_downward_block', 'number_resnet_upward_block',
                    'number_resnet_before_attn_block', 'number_resnet_after_attn_block', 'kernel_size', 'in_out_kernel_size',
                    'transition_kernel_size', 'transition_scale_factor', 'channel_expansion')}}

    # ------------------------------------------------------------------ reference surface
    def export_description(self) -> dict[str, Any]:
        cemb = self.conditional_embedding
        cemb_args = cemb.export_description() if getattr(cemb, "export_description", None) else None
        return dict(config=self.config.export_description(),
                    conditional_embedding_args=cemb_args,
                    has_conditional_embedding=cemb is not None)

    def set_conditional_embedding(self, conditional_embedding: torch.nn.Module | None = None):
        self.conditional_embedding = conditional_embedding

    @ops.device_guard
    def forward(self, x, t=None, y=None):
        """punetg.py:389-416.  x [B, Cin, H, W]; t [B] noise conditioning; y optional condition.  A top-level call: the result
        is checked by the domain guards (nets/precision.py: one device reduction and a host read) and recomputed if one fires;
        the sampler's eager path calls forward_unguarded and checks once per run."""
        out = self.forward_unguarded(x, t, y)
        if precision.needs_escalation(self, out, x):
            precision.escalate(self)
            out = self.forward_unguarded(x, t, y)
        return out

    @ops.device_guard
    def forward_unguarded(self, x, t=None, y=None):
        ops.require_device(x, "x")
        B = x.shape[0]
        ye = self.embed_condition(y)
        if ye is not None and ye.dim() > 2:                       # a field of embeddings: per-pixel time shifts
            shifts = self.field_shifts(None if t is None else self.embed_time(t.reshape(-1).to(x)), ye, B)
        else:
            if t is None:                                          # punetg.py:398-399, 410: zeros (+ ye)
                te = torch.zeros(B, self.config.model_channels, device=x.device)
                if ye is not None:
                    te = te + ye
            else:
                te = self.embed_time(t.reshape(-1).to(x), ye)
            shifts = self.time_shifts(te)
        return self.forward_with_shifts(x.contiguous(), shifts, row=None)

    # ------------------------------------------------------------------ conditioning
    def embed_condition(self, y):
        """ye of punetg.py:400-410 (conditional_embedding is a user module, run as given)."""
        if y is None:
            return None
        ye = y if self.conditional_embedding is None else self.conditional_embedding(y)
        if ye.dim() == 2 + self.dim and self.dim == 2:            # punetg.py:405-407: a field [B or 1, C, H, W]
            if ye.shape[1] != self.config.model_channels:
                raise ValueError("a field-valued conditional embedding must have model_channels channels")
            ops.require_device(ye, "conditional embedding")
            return ye.detach().to(torch.float32).contiguous()         # inference only: the kernels carry no autograd graph
        if ye.dim() != 2:
            raise NotImplementedError("field-valued conditional embeddings are implemented for 2-D networks ([B, C, H, W])")
        return ye.to(torch.float32).contiguous()

    def condition_is_field(self, y):
        """True when conditional_embedding(y) is a field: every evaluation then computes per-pixel time shifts (they depend on
        sigma, so nothing is tabulated per run; engine.ModuleSource evaluates them out of the workspace through `field_shifts`)."""
        ye = None if y is None else self.embed_condition(y)
        return ye is not None and ye.dim() > 2

    def field_shifts(self, te, ye, B):
        """The time embedding as a field, te.reshape(B, C, 1, 1) + ye (punetg.py:405-410; te [1 or B, C] or None for zeros,
        ye [1 or B, C, He, We]), wrapped for the blocks to evaluate their per-pixel time MLPs from (`_FieldShifts`).  All
        buffers are workspace buffers: forward_with_shifts gives them back."""
        if ye.shape[0] not in (1, B):
            raise ValueError("conditional embedding batch must be 1 or match x")
        if te is not None and te.shape[0] not in (1, B):
            raise ValueError("time batch must be 1 or match x")
        ws = self._ws
        field = ws.take((B,) + tuple(ye.shape[1:]), ye.device)
        if te is None:
            field.copy_(ye.expand(B, -1, -1, -1))
        else:
            torch.add(te[:, :, None, None].expand(B, -1, 1, 1), ye, out=field)
        return _FieldShifts(field, ws, self.conv_precision == "fp16x3", owned=True)

    def _field_shift(self, blk, fs, H, W):
        """ResnetTimeBlock of one block on the field at the block's resolution: the three linears as 1x1 convolutions on the
        matrix cores, SiLU in between (commonlayers.py:537-546) -> [B, C_block, H, W] from the workspace (the caller gives it back)."""
        pk, ws = self._timeblock_convs(), fs.ws
        te, a0 = fs.level(H, W)
        B, dev = te.shape[0], te.device
        n = blk.timeblock.net
        a1 = fs.am.row() if fs.am is not None else None
        a2 = fs.am.row() if fs.am is not None else None
        h = ops.conv(te, pk[id(n[0])], bias=n[0].bias, out=ws.take((B, n[0].out_features, H, W), dev),
                     **self._amax_kw(in_amax=a0, out_amax=a1))
        ops.inorm_silu(h, None, None, kind=2, out=h)                 # |SiLU(v)| <= |v|: a1 stays a valid bound
        h2 = ops.conv(h, pk[id(n[2])], bias=n[2].bias, out=ws.take((B, n[2].out_features, H, W), dev),
                      **self._amax_kw(in_amax=a1, out_amax=a2))
        ops.inorm_silu(h2, None, None, kind=2, out=h2)
        out = ops.conv(h2, pk[id(n[4])], bias=n[4].bias, out=ws.take((B, n[4].out_features, H, W), dev), **self._amax_kw(in_amax=a2))
        ws.give(h)
        ws.give(h2)
        return out

    def _timeblock_convs(self):
        lins = list(self._timeblock_linears())
        sig = (self.conv_precision,) + tuple((l.weight.data_ptr(), l.weight._version) for l in lins)
        if getattr(self, "_tb_packed_sig", None) != sig:
            with torch.no_grad():
                self._tb_packed = {}
                for l in lins:
                    w = mp_weight(l.weight.detach()) if self.mp else l.weight.detach()
                    self._tb_packed[id(l)] = ops.pack_conv(w.reshape(w.shape[0], w.shape[1], 1, 1).contiguous(),
                                                           "fp16x3" if self.conv_precision == "fp16x3" else "fp32")
            self._tb_packed_sig = sig
        return self._tb_packed

    def embed_time(self, t, ye=None):
        """te = GaussianFourierProjection(t) [+ ye]  (punetg.py:396-410)."""
        if ye is not None and ye.shape[0] not in (1, t.numel()):
            raise ValueError("conditional embedding batch must be 1 or match t")
        return ops.fourier_features(t.contiguous(), self.time_projection.W, add=ye)

    def time_shifts(self, te):
        """Per-block ResnetTimeBlock(te): list of [M, C_block] tensors in block order."""
        out = []
        pk = self.packed_weights() if self.mp else None

        def wgt(lin):
            return pk[(id(lin), "eff")] if pk is not None else lin.weight

        for blk in self._resblocks():
            n = blk.timeblock.net
            h = ops.linear(te, wgt(n[0]), n[0].bias, act=1)
            h = ops.linear(h, wgt(n[2]), n[2].bias, act=1)
            out.append(ops.linear(h, wgt(n[4]), n[4].bias, act=0))
        return out

    def _resblocks(self):
        for lv in self.downward_blocks:
            yield from lv
        yield from self.before_block
        yield from self.attn_resnet_block
        yield from self.after_block
        for lv in self.upward_blocks:
            yield from lv

    # ------------------------------------------------------------------ the reference's public stages
    # PUNetG.encode / bottom_forward / decode and their helpers (punetg.py:336-387) for callers that drive the
    # stages themselves (e.g. the encoder / decoder halves of punetg_encdec.py).  Eager launches of the same
    # kernels as forward(); te is the [B, model_channels] embedding (time projection + condition); results are
    # fresh tensors, never workspace buffers.
    def _shift_of(self, blk, te):
        pk = self.packed_weights() if self.mp else None
        n = blk.timeblock.net
        wgt = (lambda lin: pk[(id(lin), "eff")]) if pk is not None else (lambda lin: lin.weight)
        h = ops.linear(te, wgt(n[0]), n[0].bias, act=1)
        h = ops.linear(h, wgt(n[2]), n[2].bias, act=1)
        return ops.linear(h, wgt(n[4]), n[4].bias, act=0)

    def _run_blocks(self, x, te, resnet_block, attn_block=()):
        """-> (tensor, stats, owned): owned tensors are workspace buffers the caller must clone and give back."""
        if self.dim != 2:
            raise NotImplementedError("the public stage methods are implemented for 2-D networks")
        pk, ws = self.packed_weights(), self._ws
        h, hs, own = x, None, False
        for i, blk in enumerate(resnet_block):
            h2, hs2 = self._res(blk, h, self._shift_of(blk, te), pk, ws, xs=hs)
            if own:
                ws.give(h)
                if hs is not None:
                    ws.give(hs)
            h, hs, own = h2, hs2, True
            if i < len(attn_block):
                hs2 = self._stats_buf(ws, h.shape[0], h.shape[1], h.shape[2], h.shape[3], h.device)
                h2 = self._attention(attn_block[i], h, pk, ws, tile_stats=hs2)
                ws.give(h)
                if hs is not None:
                    ws.give(hs)
                h, hs = h2, hs2
        return h, hs, own

    def _release(self, h, hs, own):
        out = h.clone() if own else h
        if own:
            self._ws.give(h)
        if hs is not None:
            self._ws.give(hs)
        return out

    @ops.device_guard
    def resnet_block_forward(self, x, te, resnet_block):
        require_eval(self, self.config.dropout, self.config.cond_dropout, self.config.cond_drop)
        ops.require_device(x, "x")
        return self._release(*self._run_blocks(x.contiguous(), te, resnet_block))

    @ops.device_guard
    def resnet_attn_block_forward(self, x, te, resnet_block, attn_block):
        require_eval(self, self.config.dropout, self.config.cond_dropout, self.config.cond_drop)
        ops.require_device(x, "x")
        return self._release(*self._run_blocks(x.contiguous(), te, resnet_block, attn_block))

    def encode(self, x, te):
        """punetg.py:356-365 -> (x at the bottom resolution, [level outputs])."""
        pk = self.packed_weights()
        intermediate_outputs = []
        for resnet_block, downsampler in zip(self.downward_blocks, self.downsamplers):
            x = self.resnet_block_forward(x, te, resnet_block)
            intermediate_outputs.append(x.clone())
            x = self._conv(downsampler.conv, x, pk, load_mode=DS_LOAD_MAXPOOL2)
        return x, intermediate_outputs

    def decode(self, x, te, intermediate_outputs):
        """punetg.py:367-376 (pops the skips, like the reference)."""
        pk = self.packed_weights()
        for resnet_block, upsampler in zip(self.upward_blocks, self.upsamplers):
            x = self._conv(upsampler.conv, x.contiguous(), pk, load_mode=DS_LOAD_UPSAMPLE2, res1=intermediate_outputs.pop())
            x = self.resnet_block_forward(x, te, resnet_block)
        return x

    def bottom_forward(self, x, te):
        """punetg.py:378-387."""
        x = self.resnet_block_forward(x, te, self.before_block)
        xa = self.resnet_attn_block_forward(x, te, self.attn_resnet_block, self.attn_block)
        x = ops.add(x, xa)
        return self.resnet_block_forward(x, te, self.after_block)

    # ------------------------------------------------------------------ weights
    def _conv_modules(self):
        if not isinstance(self.convin, _FourierInput):
            yield self.convin
        yield self.convout
        for blk in self._resblocks():
            yield blk.conv1
            yield blk.conv2
        for s in list(self.downsamplers) + list(self.upsamplers):
            yield s.conv

    def _timeblock_linears(self):
        for blk in self._resblocks():
            n = blk.timeblock.net
            yield from (n[0], n[2], n[4])

    def packed_weights(self):
        """MFMA-operand repack of every conv / projection weight, cached per parameter version.  Magnitude-preserving
        layers contribute their eval-mode effective weights (mp_weight), computed here once per weight version."""
        mods = list(self._conv_modules())
        tracked = [m.weight for m in mods]
        for a in self.attn_block:
            tracked += ([a.mhattn.q_proj_matrix, a.mhattn.k_proj_matrix, a.mhattn.v_proj_matrix, a.mhattn.o_proj_matrix]
                        if self.inhouse_attn else [a.mhattn.in_proj_weight, a.mhattn.out_proj.weight])
        if self.mp:
            tracked += [lin.weight for lin in self._timeblock_linears()]
        sig = (self.conv_precision, getattr(self, "upsample_parity", True), self.exact_input_layer) + tuple((t.data_ptr(), t._version) for t in tracked)
        if self._packed is not None and sig == self._packed_sig:
            return self._packed
        pk = {}
        with torch.no_grad():
            ups = {id(u.conv) for u in self.upsamplers} if getattr(self, "upsample_parity", True) else set()
            for m in mods:
                w = m.weight.detach()
                if getattr(m, "mp", False):
                    w = mp_weight(w)
                    pk[(id(m), "eff")] = w                         # the direct output-layer kernel takes the raw layout
                if self.dim == 2:
                    pk[id(m)] = ops.pack_conv(w, self.conv_precision, upsampled=id(m) in ups)
                    if m is self.convin and self.conv_precision == "fp16x3":
                        pk[(id(m), "wmax")] = w.abs().amax(dim=(0, 2, 3)).contiguous()   # the input layer's channel guard
                        if self.exact_input_layer:
                            pk[(id(m), "exact")] = ops.pack_conv(w, "fp32")
                elif self.conv_precision == "fp16x3":              # volumes on the matrix cores: one packing per depth tap
                    pk[(id(m), "3d")] = ops.pack_conv3d(w, upsampled=id(m) in ups)
                # other precisions: ds_conv3d_direct reads the torch layout
            prec = "fp16x3" if self.conv_precision == "fp16x3" else "fp32"
            for a in self.attn_block:
                E = a.mhattn.embed_dim
                if self.inhouse_attn:
                    w_in, w_out = a.mhattn.projection_weights()
                else:
                    w_in, w_out = a.mhattn.in_proj_weight.detach(), a.mhattn.out_proj.weight.detach()
                pk[(id(a), "in")] = ops.pack_conv(w_in.reshape(3 * E, E, 1, 1), prec)
                pk[(id(a), "out")] = ops.pack_conv(w_out.reshape(E, E, 1, 1), prec)
            if self.mp:
                for lin in self._timeblock_linears():
                    pk[(id(lin), "eff")] = mp_weight(lin.weight.detach()).contiguous()
        self._packed, self._packed_sig = pk, sig
        return pk

    # ------------------------------------------------------------------ the network
    def _amax_kw(self, **kw):
        """in_amax / out_amax are arguments of the fp16x3 kernels only."""
        return kw if self.conv_precision == "fp16x3" else {}

    def _conv(self, m, x, pk, in_amax=None, out_amax=None, **kw):
        return ops.conv(x, pk[id(m)], bias=m.bias, circular=self.circular, **self._amax_kw(in_amax=in_amax, out_amax=out_amax), **kw)

    def _out_is_direct(self, m):
        return m.out_channels <= 4 and getattr(self, "direct_out", True) and self.config.in_out_kernel_size == 3

    def _out_conv(self, m, h, pk, out, circular, in_amax=None):
        """The output layer: Cout <= 4 streams the input once through the direct fp32 kernel instead of
        padding Cout to a 64-channel MFMA tile."""
        if self._out_is_direct(m):
            return ops.conv_direct(h, pk.get((id(m), "eff"), m.weight), m.bias, circular=circular, out=out)
        return ops.conv(h, pk[id(m)], bias=m.bias, circular=circular, out=out, **self._amax_kw(in_amax=in_amax))

    def _fused(self):
        return self.fuse_norm and self.conv_precision == "fp16x3"

    def _stats_buf(self, ws, B, C, H, W, dev):
        """Tile-statistics buffer for a [B, C, H, W] convolution output (None when norms are not fused)."""
        if not self._fused():
            return None
        return ws.take((B, C, ops.conv_tile_count(H, W), 4), dev)

    def _res(self, blk, x, shift, pk, ws, res2=None, xs=None, want_stats=True, out_amax=None):
        """ResnetBlockC.forward (commonlayers.py:824-833); returns (fresh buffer, its tile statistics);
        x untouched.  xs = tile statistics of x (from the convolution that produced it) or None.  out_amax: a zeroed amax row
        that receives the per-sample max |result| (the result feeds a raw-input launch: Down/UpSampler, attention)."""
        B, C, H, W = x.shape
        dev = x.device
        if self.extra_residual is not None:
            er = self.extra_residual(x)
            ops.require_device(er, "extra_residual output")
            if tuple(er.shape) != tuple(x.shape):
                raise ValueError("extra_residual must preserve the shape of its input")
            saved, self.extra_residual = self.extra_residual, None
            try:
                y, _ = self._res(blk, x, shift, pk, ws, res2=None, xs=xs, want_stats=False)
            finally:
                self.extra_residual = saved
            ops.add(y, er.contiguous(), out=y)                      # (conv2 + x) + extra_residual(x)
            if res2 is not None:
                ops.add(y, res2, out=y)                             # x + xa of bottom_forward, after the block as in the reference
            if out_amax is not None:
                ops.absmax_rows(y, out=out_amax)
            return y, None                                          # no tile statistics of the sum: the consumer normalises standalone
        if isinstance(shift, _FieldShifts):                # a field of time shifts: conv1's epilogue adds it as a residual
            yt = self._field_shift(blk, shift, H, W)
            got = self._res_body(blk, x, None, yt, pk, ws, res2, xs, want_stats, out_amax)
            ws.give(yt)
            return got
        yt = None
        if shift is not None and shift.dim() == 4:         # the same, handed over as a tensor [B, C, He, We]
            yt, shift = self._rescale_shift_field(shift, H, W), None
        return self._res_body(blk, x, shift, yt, pk, ws, res2, xs, want_stats, out_amax)

    def _res_body(self, blk, x, shift, yt, pk, ws, res2, xs, want_stats, out_amax):
        B, C, H, W = x.shape
        dev = x.device
        k1, k2 = self.norm_kinds                           # 0 GroupLN, 1 GroupRMS, 2 none, 3 GroupPix (not a table)
        w1, b1 = getattr(blk.gnorm1, "weight", None), getattr(blk.gnorm1, "bias", None)
        w2, b2 = getattr(blk.gnorm2, "weight", None), getattr(blk.gnorm2, "bias", None)
        # The folded route: the table carries the sample's activation exponent in its fourth column (a bound on the activation's
        # argument from the statistics), and the loader produces SiLU(norm(x)) times that power of two -- inside the fp16x3 window
        # whatever the affine parameters or an eps-dominated variance do.  The image / standalone routes below rely on the norm to put its output in the window:
        # real norms with affine parameters of ordinary size (windowed); otherwise the activation's exponent is measured.
        if (self._fused() and xs is not None and (C + 63) // 64 <= self.fuse_max_cot and k1 != 3 and k2 != 3
                and self.config.kernel_size == 3):                                  # the norm+SiLU loader is the 3x3 kernel's
            tab = ws.take((B, ops.table_channels(C), 4), dev)
            ops.inorm_table(xs, w1, b1, k1, H * W, eps=1e-5, out=tab)
            ys = self._stats_buf(ws, B, C, H, W, dev)
            y = self._conv(blk.conv1, x, pk, shift=shift, res1=yt, prenorm=tab, tile_stats=ys, out=ws.take(x.shape, dev))
            ops.inorm_table(ys, w2, b2, k2, H * W, eps=1e-5, out=tab)
            os_ = self._stats_buf(ws, B, C, H, W, dev) if want_stats else None
            out = self._conv(blk.conv2, y, pk, res1=x, res2=res2, prenorm=tab, tile_stats=os_, out=ws.take(x.shape, dev),
                             out_amax=out_amax)
            ws.give(y)
            ws.give(ys)
            ws.give(tab)
            return out, os_
        windowed = k1 in (0, 1) and k2 in (0, 1) and self._norms_in_window(blk)
        if windowed and self._norm_images_ok(blk, C, H, W, k1, k2):
            # standalone norms (the 256-channel level): written as the convolution's pre-split fp16 images, which it stages by
            # LDS-DMA -- same bytes as the fp32 result, bit-identical values, no split in the convolution (ops.conv_img)
            img = ops.inorm_silu_images(x, w1, b1, k1, eps=1e-5, out=ws.take((ops.conv_images_floats(B, C, H, W),), dev))
            y = ops.conv_img(img, pk[id(blk.conv1)], B, C, H, W, bias=blk.conv1.bias, shift=shift, res1=yt, out=ws.take(x.shape, dev))
            ops.inorm_silu_images(y, w2, b2, k2, eps=1e-5, out=img)
            os_ = self._stats_buf(ws, B, C, H, W, dev) if want_stats else None
            out = ops.conv_img(img, pk[id(blk.conv2)], B, C, H, W, bias=blk.conv2.bias, res1=x, res2=res2, tile_stats=os_,
                               out=ws.take(x.shape, dev), out_amax=out_amax)
            ws.give(img)
            ws.give(y)
            return out, os_
        if windowed and self._table_images_ok(blk, C, k1, k2) and self._fused() and xs is not None:
            # planes the image norm kernel does not take (more than 4096 floats): the activation from the fused loader's table
            # (built from the producer's tile statistics), written as images by an apply pass
            tab = ws.take((B, ops.table_channels(C), 4), dev)
            ops.inorm_table(xs, w1, b1, k1, H * W, eps=1e-5, out=tab)
            img = ops.table_apply_images(x, tab, out=ws.take((ops.conv_images_floats(B, C, H, W),), dev))
            ys = self._stats_buf(ws, B, C, H, W, dev)
            y = ops.conv_img(img, pk[id(blk.conv1)], B, C, H, W, bias=blk.conv1.bias, shift=shift, res1=yt, tile_stats=ys,
                             out=ws.take(x.shape, dev))
            ops.inorm_table(ys, w2, b2, k2, H * W, eps=1e-5, out=tab)
            ops.table_apply_images(y, tab, out=img)
            os_ = self._stats_buf(ws, B, C, H, W, dev) if want_stats else None
            out = ops.conv_img(img, pk[id(blk.conv2)], B, C, H, W, bias=blk.conv2.bias, res1=x, res2=res2, tile_stats=os_,
                               out=ws.take(x.shape, dev), out_amax=out_amax)
            for t in (img, y, ys, tab):
                ws.give(t)
            return out, os_
        # standalone norms; the activation's exponent is measured (one reduction pass) unless the norm puts it in the window
        a = ops.inorm_silu(x, w1, b1, kind=k1, eps=1e-5, out=ws.take(x.shape, dev))
        y = self._conv(blk.conv1, a, pk, shift=shift, res1=yt, out=ws.take(x.shape, dev), in_amax=self._act_amax(a, windowed))
        ops.inorm_silu(y, w2, b2, kind=k2, eps=1e-5, out=a)
        os_ = self._stats_buf(ws, B, C, H, W, dev) if want_stats else None
        self._conv(blk.conv2, a, pk, res1=x, res2=res2, tile_stats=os_, out=y, in_amax=self._act_amax(a, windowed), out_amax=out_amax)
        ws.give(a)
        return y, os_

    def _consumes_stats(self, blk, C, H, W):
        """Does _res(blk, x [., C, H, W], xs=...) read the tile statistics of its input?  (The folded-loader route and the
        table -> images route do; the image route and the standalone norms compute their own.)  Producers ask before they spend
        epilogue work on statistics nobody reads: at config 2 that was every launch of the 256-channel level, the last block of
        every level (its result feeds a Down / UpSampler) and the attention's output projection."""
        if blk is None or not self._fused() or self.extra_residual is not None:
            return False
        k1, k2 = self.norm_kinds
        if (C + 63) // 64 <= self.fuse_max_cot and k1 != 3 and k2 != 3 and self.config.kernel_size == 3:
            return True
        windowed = k1 in (0, 1) and k2 in (0, 1) and self._norms_in_window(blk)
        if windowed and self._norm_images_ok(blk, C, H, W, k1, k2):
            return False
        return windowed and self._table_images_ok(blk, C, k1, k2)

    def _act_amax(self, a, windowed):
        """in_amax of a standalone norm + SiLU output: none needed inside the fp16x3 window, else a reduction into an arena row
        (a row of the current forward's arena; outside a forward -- never -- ops reduces into a fresh tensor)."""
        if windowed or self.conv_precision != "fp16x3":
            return ops.NORMALISED
        return self._am.of(a) if self._am is not None else None

    def _norms_in_window(self, blk):
        """Both norms of the block are affine-free or carry affine parameters of ordinary size (|w|, |b| largest entries within
        [2^-6, 2^6] / below 2^6): SiLU(norm(x) * w + b) then sits inside the fp16x3 window (|x| in [2^-3, 2^16) at 22 bits,
        degrading gracefully to 2^-25 absolute) for any input magnitude.  Checked on the host once per parameter version."""
        return precision.norms_in_window(self._window_cache, id(blk), (blk.gnorm1, blk.gnorm2))

    def _table_images_ok(self, blk, C, k1, k2):
        """As _norm_images_ok for the table route (any plane size; GroupLN / GroupRMS / no norm)."""
        return (getattr(self, "norm_images", True) and self.conv_precision == "fp16x3" and not self.circular
                and self.config.kernel_size == 3 and k1 in (0, 1, 2) and k2 in (0, 1, 2) and ((C + 15) // 16) % 2 == 0
                and blk.conv1.out_channels == C and blk.conv2.out_channels == C)

    def _norm_images_ok(self, blk, C, H, W, k1, k2):
        """The standalone norms of this block can hand their convolutions pre-split images: fp16x3 3x3 convolutions with zero
        padding that keep the channel count, an even number of 16-channel chunks, GroupLN / GroupRMS, planes the image kernel
        takes."""
        return (getattr(self, "norm_images", True) and self.conv_precision == "fp16x3" and not self.circular
                and self.config.kernel_size == 3 and k1 in (0, 1) and k2 in (0, 1) and ((C + 15) // 16) % 2 == 0
                and blk.conv1.out_channels == C and blk.conv2.out_channels == C and ops.inorm_silu_images_supported(H, W))

    @staticmethod
    def _rescale_shift_field(yt, H, W):
        """ResnetBlockC.rescale_yt (commonlayers.py:838-869): the top-left corner of every window (CornerPool2d) when the
        field is finer than the block.  A coarser field takes the reference through torch.nn.Upsample(shape_factor), whose
        first argument is the output size -- it fails there unless the block's side equals the factor; not reproduced."""
        h, w = yt.shape[2:]
        if (h, w) == (H, W):
            return yt
        if h > H:
            f = h // H
            if H * f != h or W * f != w:
                raise ValueError(f"yt_dims {(h, w)} and y_dims {(H, W)} are not compatible")
            return yt[:, :, ::f, ::f].contiguous()
        raise NotImplementedError("a conditional-embedding field coarser than a block's resolution (the reference's "
                                  "upscaling branch passes the factor as torch.nn.Upsample's size and fails as well)")

    def forward_with_shifts(self, x, shifts, row=None, out=None):
        """UNet body given the per-block time shifts.  shifts[k] is [M, C_k]; row selects one row
        shared by the whole batch (sampling: sigma is a per-step constant), row=None means one row
        per sample (M == B).  Every activation travels with the tile statistics its producer left."""
        require_eval(self, self.config.dropout, self.config.cond_dropout, self.config.cond_drop)
        if self.dim == 3:
            return self._forward3d(x, shifts, row=row, out=out)
        pk = self.packed_weights()
        ws = self._ws
        cfg = self.config
        B = x.shape[0]
        dev = x.device
        lazy_shifts = shifts if isinstance(shifts, _FieldShifts) else None
        it = iter(range(len(shifts))) if lazy_shifts is None else None

        def sh():
            if lazy_shifts is not None:                # per-pixel shifts: each block evaluates its own (_field_shift)
                return lazy_shifts
            s = shifts[next(it)]
            if s.dim() == 4:                           # [B, C, He, We]: a field of shifts handed over as a tensor (allocates: not for captured runs)
                return s
            if row is not None:
                if s.dim() == 3:                       # [n_evals, B, C]: per-sample conditions in the planned sampler
                    if s.shape[1] != B:
                        raise ValueError("time embedding batch does not match x")
                    return s[row]
                return s[row:row + 1]
            if s.shape[0] not in (1, B):
                raise ValueError("time embedding batch does not match x")
            return s

        def give(t, ts):
            ws.give(t)
            if ts is not None:
                ws.give(ts)

        # Activation exponents (ops.py): every launch that reads a tensor which no norm has put into the fp16x3 window --
        # convin, the Down / UpSamplers, the attention and its projections, a k x k output layer -- takes the per-sample max |x|
        # its producer's epilogue left in a row of this arena (ha travels with h like the tile statistics hs), or a reduction
        # over the tensor where the producer is not one of our epilogues (the network input).
        h3 = self.conv_precision == "fp16x3"
        # (the arena is zeroed by the input layer's own reduction launch when that is the first thing the forward does)
        lazy = h3 and not isinstance(self.convin, _FourierInput) and not self.exact_input_layer
        am = self._am = _AmaxArena(ws, B, dev, zero=not lazy) if h3 else None

        def slot(needed=True):
            return am.row() if (h3 and needed) else None

        def amax_of(t, ta):                                                      # the row that travels with t, else a reduction
            if not h3:
                return None
            return ta if ta is not None else am.of(t)

        try:
            H, W = x.shape[2:]
            xe = None
            if not cfg.bias:                                                         # punetg.py:390-394
                ones = ws.take((B, 1, H, W), dev)
                ones.fill_(1.0)
                xe = ops.concat2(x, ones, out=ws.take((B, x.shape[1] + 1, H, W), dev))
                ws.give(ones)
                x = xe
            ndown = len(self.downward_blocks)
            bottom = list(self.before_block) + list(self.attn_resnet_block) + list(self.after_block)

            def stats_for(nxt, C_, H_, W_):                                          # a statistics buffer only if the consumer reads it
                return self._stats_buf(ws, B, C_, H_, W_, dev) if self._consumes_stats(nxt, C_, H_, W_) else None

            def first_block_after_level(lv):
                if lv + 1 < ndown and len(self.downward_blocks[lv + 1]):
                    return self.downward_blocks[lv + 1][0]
                return bottom[0] if bottom else None
            if isinstance(self.convin, _FourierInput):
                hs = None                                                            # no producer statistics: standalone first norm
                h = ops.fourier_channels(x, self.convin.W, out=ws.take((B, cfg.model_channels, H, W), dev))
            elif h3 and self.exact_input_layer:
                # the input's channels are too far apart in magnitude for one exponent per sample (precision.escalate_input):
                # exact-fp32 kernel, no tile statistics (the first block normalises standalone)
                hs = None
                # (circular= : a periodic network with exact_input_layer set by hand must raise -- the exact-fp32 kernel zero-pads)
                h = ops.conv(x, pk[(id(self.convin), "exact")], bias=self.convin.bias, circular=self.circular,
                             out=ws.take((B, cfg.model_channels, H, W), dev))
            else:
                first = self.downward_blocks[0][0] if (ndown and len(self.downward_blocks[0])) else (bottom[0] if bottom else None)
                hs = stats_for(first, cfg.model_channels, H, W)
                h = self._conv(self.convin, x, pk, tile_stats=hs, out=ws.take((B, cfg.model_channels, H, W), dev),
                               in_amax=am.of_input(x, precision.input_layer_flag(self, dev), pk[(id(self.convin), "wmax")]) if h3 else None)
            ha = None
            if xe is not None:
                ws.give(xe)
            skips = []
            for lv, blocks in enumerate(self.downward_blocks):                      # punetg.py:356-365
                for j, blk in enumerate(blocks):
                    ha2 = slot(j == len(blocks) - 1)                                 # the level's last block feeds the DownSampler
                    nxt = blocks[j + 1] if j + 1 < len(blocks) else None             # ... and the skip: nobody normalises its result
                    h2, hs2 = self._res(blk, h, sh(), pk, ws, xs=hs, out_amax=ha2,
                                        want_stats=self._consumes_stats(nxt, h.shape[1], h.shape[2], h.shape[3]))
                    give(h, hs)
                    h, hs, ha = h2, hs2, ha2
                skips.append(h)
                if hs is not None:
                    ws.give(hs)                                                      # the skip is only added, never normalised
                ds = self.downsamplers[lv].conv
                Ho, Wo = h.shape[2] // 2, h.shape[3] // 2
                hs = stats_for(first_block_after_level(lv), ds.out_channels, Ho, Wo)
                h = self._conv(ds, h, pk, load_mode=DS_LOAD_MAXPOOL2, tile_stats=hs,
                               out=ws.take((B, ds.out_channels, Ho, Wo), dev), in_amax=amax_of(h, ha))
                ha = None
            nattn, nafter = len(self.attn_resnet_block), len(self.after_block)
            for j, blk in enumerate(self.before_block):                               # punetg.py:378-387
                ha2 = slot(j == len(self.before_block) - 1 and nattn == 0 and nafter == 0 and ndown > 0)
                nxt = bottom[j + 1] if (j + 1 < len(self.before_block) or nattn > 0) else None
                h2, hs2 = self._res(blk, h, sh(), pk, ws, xs=hs, out_amax=ha2,
                                    want_stats=self._consumes_stats(nxt, h.shape[1], h.shape[2], h.shape[3]))
                give(h, hs)
                h, hs, ha = h2, hs2, ha2
            xa, xas, xaa = h, hs, ha
            for i, blk in enumerate(self.attn_resnet_block):
                last = i == nattn - 1
                attn_next = i < len(self.attn_block)
                # x + xa is folded into the last residual block's epilogue when no attention follows it
                xaa2 = slot(attn_next or (last and nafter == 0 and ndown > 0))
                # its result feeds the attention (no statistics read), the next block of this group, or -- the last -- after_block
                nb = len(self.before_block)
                nxt = None if attn_next else (bottom[nb + i + 1] if nb + i + 1 < len(bottom) else None)
                xa2, xas2 = self._res(blk, xa, sh(), pk, ws, xs=xas,
                                      res2=h if (last and not attn_next) else None, out_amax=xaa2,
                                      want_stats=self._consumes_stats(nxt, xa.shape[1], xa.shape[2], xa.shape[3]))
                if xa is not h:
                    give(xa, xas)
                xa, xas, xaa = xa2, xas2, xaa2
                if attn_next:
                    nxt = bottom[nb + i + 1] if nb + i + 1 < len(bottom) else None
                    xas2 = stats_for(nxt, xa.shape[1], xa.shape[2], xa.shape[3])
                    xaa2 = slot(last and nafter == 0 and ndown > 0)
                    xa2 = self._attention(self.attn_block[i], xa, pk, ws, res2=h if last else None, tile_stats=xas2,
                                          in_amax=amax_of(xa, xaa), out_amax=xaa2)
                    give(xa, xas)
                    xa, xas, xaa = xa2, xas2, xaa2
            if nattn == 0:
                xa, xas, xaa = ops.add(h, h, out=ws.take(h.shape, dev)), None, None
            give(h, hs if xas is not hs else None)
            h, hs, ha = xa, xas, xaa
            for j, blk in enumerate(self.after_block):
                ha2 = slot(j == nafter - 1 and ndown > 0)                            # feeds the first UpSampler
                nxt = self.after_block[j + 1] if j + 1 < nafter else None
                h2, hs2 = self._res(blk, h, sh(), pk, ws, xs=hs, out_amax=ha2,
                                    want_stats=self._consumes_stats(nxt, h.shape[1], h.shape[2], h.shape[3]))
                give(h, hs)
                h, hs, ha = h2, hs2, ha2
            nup = len(self.upward_blocks)
            direct_out = self._out_is_direct(self.convout)
            for lv, blocks in enumerate(self.upward_blocks):                         # punetg.py:367-376
                us = self.upsamplers[lv].conv
                skip = skips.pop()
                hs2 = stats_for(blocks[0] if len(blocks) else None, skip.shape[1], skip.shape[2], skip.shape[3])
                h2 = self._conv(us, h, pk, load_mode=DS_LOAD_UPSAMPLE2, res1=skip, tile_stats=hs2,
                                out=ws.take(skip.shape, dev), in_amax=amax_of(h, ha))
                give(h, hs)
                ws.give(skip)
                h, hs, ha = h2, hs2, None
                for j, blk in enumerate(blocks):
                    lastb = j == len(blocks) - 1
                    final = lv == nup - 1 and lastb                                  # feeds convout: no norm follows
                    ha2 = slot(lastb and (not final or not direct_out))              # the next UpSampler, or a matrix-core output layer
                    nxt = None if lastb else blocks[j + 1]                           # the last one feeds an UpSampler or convout: no norm follows
                    h2, hs2 = self._res(blk, h, sh(), pk, ws, xs=hs, out_amax=ha2,
                                        want_stats=self._consumes_stats(nxt, h.shape[1], h.shape[2], h.shape[3]))
                    give(h, hs)
                    h, hs, ha = h2, hs2, ha2
            y = self._out_conv(self.convout, h, pk, out, self.circular, in_amax=None if direct_out else amax_of(h, ha))
            give(h, hs)
            return y
        finally:
            if am is not None:
                am.release()
            self._am = None
            if lazy_shifts is not None:
                lazy_shifts.release()

    # ------------------------------------------------------------------ volumes (dimension = 3)
    def _forward3d(self, x, shifts, row=None, out=None):
        """The same network on [B, C, D, H, W] volumes (punetg.py:217-236,389-416 with Conv3d / MaxPool3d /
        Upsample / ThreeDimensionalAttention).  Convolutions: with the default fp16x3 precision three launches of the 2-D
        matrix-core kernels per 3x3x3 convolution over a slice-major copy of the volume (ops.conv3d_mfma); otherwise, and
        for the <= 4-channel output layer, the exact-fp32 direct kernel (ops.conv3d) -- both with the pooling /
        upsampling / skip / residual / time-shift fusions of the 2-D path.  Standalone per-(sample, channel) norms over
        D*H*W, attention over the flattened voxels.  All buffers come from the workspace: the sampler's planner captures
        this path as a hipGraph too."""
        if x.dim() != 5:
            raise ValueError("a dimension=3 network takes [B, C, D, H, W] volumes")
        pk = self.packed_weights()
        ws = self._ws
        cfg = self.config
        B, dev = x.shape[0], x.device
        it = iter(range(len(shifts)))
        k1, k2 = self.norm_kinds

        def sh():
            s = shifts[next(it)]
            if row is not None:
                if s.dim() == 3:                       # [n_evals, B, C]: per-sample conditions in the planned sampler
                    if s.shape[1] != B:
                        raise ValueError("time embedding batch does not match x")
                    return s[row]
                return s[row:row + 1]
            if s.shape[0] not in (1, B):
                raise ValueError("time embedding batch does not match x")
            return s

        # Norm folding on volumes (round 2; fp16x3, 3x3x3 kernels; round 3: periodic padding too): every activation travels with the partial
        # sums its producer's slice -> volume copy left (hs); a block whose input has them runs ops.resblock3d_fused --
        # norm1 inside the volume -> slice copy, the intermediate slice-major with norm2 in conv2's loader -- 20 instead of
        # 52 bytes per element of norm / copy traffic per block.  Without statistics (after the thin input layer or the
        # attention) the block runs the standalone norms and leaves statistics for its successor.
        fold = self._fused() and self.extra_residual is None and k1 != 3 and k2 != 3 and cfg.kernel_size == 3

        def stats_buf(shape):
            Bc, C, D, H, W = shape
            return ws.take((Bc, C, ops.volume_stat_tiles(D, H * W), 4), dev)

        def conv(m, h, load_mode=0, dst=None, fresh=False, want_stats=False, normalised=False, **kw):
            """-> (tensor, statistics or None).  Every buffer comes from the workspace (a captured loop must not allocate);
            fresh: the caller's result.  normalised: h is a norm + SiLU output inside the fp16x3 window; otherwise the
            matrix-core route measures per-slice activation exponents on its slice copy (ops.conv3d_mfma)."""
            f = {0: (1, 1), DS_LOAD_MAXPOOL2: (1, 2), DS_LOAD_UPSAMPLE2: (2, 1)}[load_mode]
            shape = (h.shape[0], m.out_channels) + tuple(v * f[0] // f[1] for v in h.shape[2:])
            if dst is None and not fresh:
                dst = ws.take(shape, dev)
            packs = pk.get((id(m), "3d"))
            # fp16x3 (default): three 2-D MFMA launches per convolution -- 0.30 vs 1.33 ms at 64 -> 64 channels, 8 x 32^3;
            # the thin input / output layers stay on the direct kernel (0.08 vs 0.14 ms for 1 -> 64)
            k = m.weight.shape[-1]
            if packs is None and k != 3:
                raise NotImplementedError(f"{k}x{k}x{k} kernels on volumes are implemented on the fp16x3 convolution only "
                                          f"(conv_precision={self.conv_precision!r})")
            if packs is not None and ((m.out_channels > 4 and m.in_channels > 4) or k != 3):
                st = stats_buf(shape) if (fold and want_stats) else None
                return ops.conv3d_mfma(h, packs, bias=m.bias, circular=self.circular, load_mode=load_mode, out=dst, ws=ws,
                                       out_stats=st, in_amax=ops.NORMALISED if normalised else None, **kw), st
            return ops.conv3d(h, pk.get((id(m), "eff"), m.weight), bias=m.bias, circular=self.circular, load_mode=load_mode,
                              out=dst, **kw), None

        def give(t, ts=None):
            ws.give(t)
            if ts is not None:
                ws.give(ts)

        def res(blk, h, hs, res2=None, want_stats=True):                          # ResnetBlockC.forward; h untouched
            w1, b1 = getattr(blk.gnorm1, "weight", None), getattr(blk.gnorm1, "bias", None)
            w2, b2 = getattr(blk.gnorm2, "weight", None), getattr(blk.gnorm2, "bias", None)
            C = h.shape[1]
            p1, p2 = pk.get((id(blk.conv1), "3d")), pk.get((id(blk.conv2), "3d"))
            windowed = k1 in (0, 1) and k2 in (0, 1) and self._norms_in_window(blk)
            if (fold and windowed and hs is not None and p1 is not None and p2 is not None and C > 4
                    and (C + 63) // 64 <= self.fuse_max_cot):
                tab = ops.inorm_table(hs, w1, b1, k1, h[0, 0].numel(), eps=1e-5, out=ws.take((B, ops.table_channels(C), 4), dev))
                os_ = stats_buf(h.shape) if want_stats else None
                y = ops.resblock3d_fused(h, tab, p1, blk.conv1.bias, sh(), p2, blk.conv2.bias, w2, b2, k2, res2=res2,
                                         out=ws.take(h.shape, dev), out_stats=os_, ws=ws, circular=self.circular)
                ws.give(tab)
                return y, os_
            a = ops.inorm_silu(h, w1, b1, kind=k1, eps=1e-5, out=ws.take(h.shape, dev))
            y, _ = conv(blk.conv1, a, shift=sh(), normalised=windowed)
            ops.inorm_silu(y, w2, b2, kind=k2, eps=1e-5, out=a)
            if self.extra_residual is None:
                _, os_ = conv(blk.conv2, a, res1=h, res2=res2, dst=y, want_stats=want_stats, normalised=windowed)
            else:
                conv(blk.conv2, a, res1=h, dst=y, normalised=windowed)
                ops.add(y, self.extra_residual(h).contiguous(), out=y)
                if res2 is not None:
                    ops.add(y, res2, out=y)
                os_ = None
            ws.give(a)
            return y, os_

        def attn(att, h, res2=None):                                              # ThreeDimensionalAttention
            Bq, E, D, H, W = h.shape
            r2 = None if res2 is None else res2.view(Bq, E, D * H, W)
            y = self._attention(att, h.view(Bq, E, D * H, W), pk, ws, res2=r2)
            o = ws.take(h.shape, dev)
            o.copy_(y.view(h.shape))
            ws.give(y)
            return o

        x = x.contiguous()
        xe = None
        if not cfg.bias:                                                          # punetg.py:390-394
            ones = ws.take((B, 1) + tuple(x.shape[2:]), dev)
            ones.fill_(1.0)
            xe = ops.concat2(x, ones, out=ws.take((B, x.shape[1] + 1) + tuple(x.shape[2:]), dev))
            ws.give(ones)
            x = xe
        hs = None
        if isinstance(self.convin, _FourierInput):
            h = ops.fourier_channels(x, self.convin.W, out=ws.take((B, cfg.model_channels) + tuple(x.shape[2:]), dev))
        else:
            h, hs = conv(self.convin, x, want_stats=True)
        if xe is not None:
            ws.give(xe)
        skips = []
        for lv, blocks in enumerate(self.downward_blocks):
            for blk in blocks:
                h2, hs2 = res(blk, h, hs)
                give(h, hs)
                h, hs = h2, hs2
            skips.append(h)
            if hs is not None:
                ws.give(hs)                                                      # the skip is only added, never normalised
            h, hs = conv(self.downsamplers[lv].conv, h, load_mode=DS_LOAD_MAXPOOL2, want_stats=True)
        for blk in self.before_block:
            h2, hs2 = res(blk, h, hs)
            give(h, hs)
            h, hs = h2, hs2
        xa, xas = h, hs
        nattn = len(self.attn_resnet_block)
        for i, blk in enumerate(self.attn_resnet_block):
            last = i == nattn - 1
            xa2, xas2 = res(blk, xa, xas, res2=h if (last and i >= len(self.attn_block)) else None)
            if xa is not h:
                give(xa, xas)
            xa, xas = xa2, xas2
            if i < len(self.attn_block):
                xa2 = attn(self.attn_block[i], xa, res2=h if last else None)
                give(xa, xas)
                xa, xas = xa2, None
        if nattn == 0:
            xa, xas = ops.add(h, h, out=ws.take(h.shape, dev)), None
        give(h, hs if xas is not hs else None)
        h, hs = xa, xas
        for blk in self.after_block:
            h2, hs2 = res(blk, h, hs)
            give(h, hs)
            h, hs = h2, hs2
        nup = len(self.upward_blocks)
        for lv, blocks in enumerate(self.upward_blocks):
            skip = skips.pop()
            h2, hs2 = conv(self.upsamplers[lv].conv, h, load_mode=DS_LOAD_UPSAMPLE2, res1=skip, want_stats=True)
            give(h, hs)
            ws.give(skip)
            h, hs = h2, hs2
            for j, blk in enumerate(blocks):
                final = lv == nup - 1 and j == len(blocks) - 1                    # feeds convout: no norm follows
                h2, hs2 = res(blk, h, hs, want_stats=not final)
                give(h, hs)
                h, hs = h2, hs2
        y, _ = conv(self.convout, h, dst=out, fresh=out is None)
        give(h, hs)
        return y

    def _attention(self, att, x, pk, ws, res2=None, tile_stats=None, in_amax=None, out_amax=None):
        """TwoDimensionalAttention.forward (attention.py:67-72,82-90), channel-major throughout.  The block's three launches
        read raw tensors: x (in_amax: its producer's row, or None = reduced here), qkv and the attention output, whose
        exponents travel from epilogue to loader through rows of the forward's arena."""
        B, E, Hh, Ww = x.shape
        L = Hh * Ww
        m = att.mhattn
        in_bias = None if self.inhouse_attn else m.in_proj_bias    # the in-house attention has no biases
        out_bias = None if self.inhouse_attn else m.out_proj.bias
        am = self._am if self.conv_precision == "fp16x3" else None
        own = None
        if am is None and self.conv_precision == "fp16x3":           # called outside forward_with_shifts (the 3-D path sets its own)
            own = am = _AmaxArena(ws, B, x.device)
        split = 2 * E if E % 32 == 0 else 0                           # one exponent for q and k, one for v (E % 32: the fp16x3 attention)
        a_qkv = am.rows(2) if am is not None else None
        a_o = am.row() if am is not None else None
        if am is not None and in_amax is None:
            in_amax = am.of(x)
        direct = not self.cosine_attn and split > 0
        qkv = ops.conv(x, pk[(id(att), "in")], bias=in_bias, out=ws.take((B, 3 * E, Hh, Ww), x.device),
                       **self._amax_kw(in_amax=in_amax, out_amax=a_qkv if direct else None, amax_split=split if direct else 0))
        if self.cosine_attn:
            # cosine_similarity (attention.py:362-372): unit queries and keys, logits without 1/sqrt(E) -- the
            # attention kernels scale by 1/sqrt(E), which the queries' gain cancels
            ops.token_l2_normalize(qkv.view(B, 3 * E, L), 0, E, eps=1e-8, gain=math.sqrt(E))
            ops.token_l2_normalize(qkv.view(B, 3 * E, L), E, E, eps=1e-8, gain=1.0)
        if am is not None and not direct:                             # q and k were rewritten, or a head width the epilogue cannot split: measure
            ops.absmax_rows(qkv[:, :2 * E], out=a_qkv[:B])
            ops.absmax_rows(qkv[:, 2 * E:], out=a_qkv[B:])
        nws = ops.attention_workspace_floats(B, E, L, self.conv_precision)
        aws = ws.take((nws,), x.device) if nws else None
        o = ops.attention(qkv.view(B, 3 * E, L), E, out=ws.take((B, E, L), x.device),
                          precision=self.conv_precision, workspace=aws, **self._amax_kw(in_amax=a_qkv, out_amax=a_o))
        if aws is not None:
            ws.give(aws)
        res1 = x if self.config.attn_residual else None
        y = ops.conv(o.view(B, E, Hh, Ww), pk[(id(att), "out")], bias=out_bias,
                     res1=res1, res2=res2, tile_stats=tile_stats, out=ws.take(x.shape, x.device),
                     **self._amax_kw(in_amax=a_o, out_amax=out_amax))
        ws.give(qkv)
        ws.give(o)
        if own is not None:
            own.release()
        return y


class PUNetGCond(PUNetG):
    """PUNetG with channel-concatenated conditioning (punetg.py:706-735): the fields y[item] for item in
    ``channel_conditional_items`` are appended to x as input channels (config.input_channels counts them);
    the remaining entries of y go to the conditional embedding.  Like the reference it needs y on every call,
    so it cannot run the unconditional branch of classifier-free guidance (guidance must be 1)."""

    def __init__(self, config: PUNetGConfig, conditional_embedding: torch.nn.Module | None = None,
                 channel_conditional_items: list[str] | None = False, extra_residual: torch.nn.Module | None = None):
        super().__init__(config, conditional_embedding, extra_residual=extra_residual)
        self.channel_conditional_items = channel_conditional_items
        self._ycat = None
        self._ycat_static = {}       # (shape, device) -> buffer holding the concatenated channel fields

    def export_description(self) -> dict[str, Any]:
        args = super().export_description()
        args["channel_conditional_items"] = self.channel_conditional_items
        return args

    def _split_condition(self, y):
        if y is None:
            raise TypeError("PUNetGCond needs the condition dictionary y on every call (punetg.py:721-723)")
        fields = [y[item] for item in self.channel_conditional_items]
        rest = {k: v for k, v in y.items() if k not in self.channel_conditional_items}
        for f in fields:
            ops.require_device(f, "channel condition")
            if f.dim() < 3 or tuple(f.shape[2:]) != tuple(fields[0].shape[2:]) or f.shape[0] != fields[0].shape[0]:
                raise ValueError("channel condition fields must be [B or 1, C_i, *spatial] with equal batch and spatial sizes")
        # torch.cat([y[item] ...], dim=1) of punetg.py:724-727 INTO a buffer the network owns: a captured sampling
        # loop reads the fields at this address on every replay, so the caller's tensors (new ones on every call of
        # autoregressive_sample; temporaries of torch.cat) must never be what the graph points at.  One buffer per
        # (shape, device); every call -- eager or as the refresh before a replay -- rewrites it.
        shape = (fields[0].shape[0], sum(f.shape[1] for f in fields)) + tuple(fields[0].shape[2:])
        key = (shape, str(fields[0].device))
        buf = self._ycat_static.get(key)
        if buf is None:                  # never evicted: captured plans keep reading the buffer of their shape
            with torch.inference_mode(False):          # a normal tensor: written under inference_mode and outside it
                buf = torch.empty(shape, dtype=torch.float32, device=fields[0].device)
            self._ycat_static[key] = buf
        c0 = 0
        for f in fields:
            buf[:, c0:c0 + f.shape[1]].copy_(f)
            c0 += f.shape[1]
        return (rest if len(rest) else None), buf

    def _with_condition(self, x, ycat, ws):
        B = x.shape[0]
        yexp = None
        if ycat.shape[0] == 1 and B > 1:
            # broadcast into a WORKSPACE buffer: a temporary from torch's allocator would be freed right after the
            # capture while the graph keeps writing to its address on every replay
            yexp = ws.take((B,) + tuple(ycat.shape[1:]), x.device)
            yexp.copy_(ycat.expand(B, *ycat.shape[1:]))
            ycat = yexp
        elif ycat.shape[0] != B:
            raise ValueError("channel condition batch must be 1 or match x")
        out = ops.concat2(x, ycat, out=ws.take((B, x.shape[1] + ycat.shape[1]) + tuple(x.shape[2:]), x.device))
        if yexp is not None:
            ws.give(yexp)
        return out

    @ops.device_guard
    def forward(self, x, t, y=None):
        out = self.forward_unguarded(x, t, y)
        if precision.needs_escalation(self, out, x, self._ycat):
            precision.escalate(self)
            out = self.forward_unguarded(x, t, y)
        return out

    @ops.device_guard
    def forward_unguarded(self, x, t, y=None):
        ops.require_device(x, "x")
        rest, self._ycat = self._split_condition(y)
        te = self.embed_time(t.reshape(-1).to(x), PUNetG.embed_condition(self, rest))
        shifts = self.time_shifts(te)
        return self.forward_with_shifts(x.contiguous(), shifts, row=None)

    def embed_condition(self, y):
        """Planned sampler entry: remember the channel fields, embed what is left of y."""
        rest, self._ycat = self._split_condition(y)
        return PUNetG.embed_condition(self, rest)

    def forward_with_shifts(self, x, shifts, row=None, out=None):
        if self._ycat is None:
            raise TypeError("PUNetGCond needs the condition dictionary y on every call (punetg.py:721-723)")
        xc = self._with_condition(x, self._ycat, self._ws)
        try:
            return super().forward_with_shifts(xc, shifts, row=row, out=out)
        finally:
            self._ws.give(xc)
