"""ADM score network on HIP kernels (reference: diffsci/models/nets/adm.py).

Same constructor, ``net(x, t, y=None)`` protocol and state_dict key names as the reference's
``ADM`` / ``ADMConfig`` (adm.py:8-216), for the default family: 2-D fields, default convolutions,
GroupLN(1 group) + GroupRMS(1 group) norms, avg-pool down / nearest up inside the last block of a
layer, decoder_type 1 or 2, single-head attention in the middle block.  The torch.nn layers are
parameter containers only; every tensor operation is a launch into libdiffsci_hip.so:

  input/output layer, conv1 (+nearest-up load), conv2 (+residual)   ds_conv2d*
  convresidual (1x1, +nearest-up load)                              ds_conv2d
  GroupNorm(1,C)+SiLU [+AvgPool2d], GroupRMSNorm(1,C)+FiLM+SiLU      ds_gnorm1_stats, ds_gnorm1_apply
  AvgPool2d on the residual branch                                  ds_gnorm1_apply (kind 2)
  skip concat / add                                                 ds_concat2 / ds_add
  ADMTimeEmbedding, embed_linear                                    ds_fourier_features, ds_linear, ds_add_act
  attention                                                         ds_conv2d (1x1) + ds_attention*
"""
from typing import Any
import pathlib

import os

import torch
import yaml

from ... import ops
from ..._native import DS_LOAD_AVGPOOL2, DS_LOAD_PLAIN, DS_LOAD_UPSAMPLE2
from . import precision
from .punetg import _AffineHolder, _AmaxArena, _Attn, _CircConv, _Fourier, _Workspace, make_conv, require_eval

_FIELDS = dict(
    input_channels=1, output_channels=1, dimension=2, model_channels=64, time_embed_dim=64,
    output_embed_dim=256, channel_expansion=(2, 4),
    number_resnet_downward_block=2, number_resnet_upward_block=2, number_resnet_attn_block=2,
    number_resnet_before_attn_block=2, number_resnet_after_attn_block=2,
    kernel_size=3, time_projection_scale=30.0, transition_scale_factor=2, transition_kernel_size=3,
    dropout=0.0, cond_dropout=0.0, first_resblock_norm="GroupLN", second_resblock_norm="GroupRMS",
    affine_norm=True, convolution_type="default", num_groups=1, skip_integration_type="concat",
    attn_residual=True, decoder_type=1)


class ADMConfig(object):
    """adm.py:8-116 -- same arguments and defaults."""

    def __init__(self, input_channels=1, output_channels=1, dimension=2, model_channels=64, time_embed_dim=64, output_embed_dim=256,
                 channel_expansion=(2, 4), number_resnet_downward_block=2, number_resnet_upward_block=2,
                 number_resnet_attn_block=2, number_resnet_before_attn_block=2, number_resnet_after_attn_block=2, kernel_size=3,
                 time_projection_scale=30.0, transition_scale_factor=2, transition_kernel_size=3, dropout=0.0, cond_dropout=0.0,
                 first_resblock_norm="GroupLN", second_resblock_norm="GroupRMS", affine_norm=True, convolution_type="default",
                 num_groups=1, skip_integration_type="concat", attn_residual=True, decoder_type=1):
        given = locals()                                 # positional order and defaults of the reference's constructor (adm.py:8-40)
        for k in _FIELDS:
            v = given[k]
            if k == "channel_expansion":
                v = list(v)
            setattr(self, k, v)

    @property
    def middle_channel(self):
        return self.model_channels * self.channel_expansion[-1]

    @property
    def extended_channel_expansion(self):
        return [1] + list(self.channel_expansion)

    @property
    def middle_block_attn_config(self):
        return ([False] * self.number_resnet_before_attn_block +
                [True] * (self.number_resnet_attn_block - 1) + [False] +
                [False] * self.number_resnet_after_attn_block)

    @property
    def num_blocks_middle_block(self):
        return (self.number_resnet_before_attn_block + self.number_resnet_attn_block +
                self.number_resnet_after_attn_block)

    def export_description(self) -> dict[str, Any]:
        return {k: getattr(self, k) for k in _FIELDS}

    @classmethod
    def from_description(cls, description: dict):
        return cls(**description)

    @classmethod
    def from_config_file(cls, config_file: pathlib.Path | str):
        with open(config_file, "r") as f:
            return cls.from_description(yaml.safe_load(f))

    def unsupported_reason(self):
        checks = [
            (self.dimension == 2, "only 2-D fields (dimension=2): the reference's own ADM cannot run on volumes (its stem and "
                                  "output layers are Conv2d, adm.py:186-195); its 3-D capable part, the residual blocks, "
                                  "is ADMEncoderBlock / ADMDecoderBlock(dimension=3)"),
            (self.convolution_type in ("default", "circular"), "convolution_type 'default' or 'circular'"),
            (self.first_resblock_norm in ("GroupLN", "GroupRMS") and self.second_resblock_norm in ("GroupLN", "GroupRMS"),
             "first/second_resblock_norm 'GroupLN' or 'GroupRMS' (the reference raises on anything else, adm.py:395,406)"),
            (self.num_groups == 1, "num_groups=1"),
            (self.kernel_size == 3, "kernel_size=3"),
            (self.transition_scale_factor == 2, "transition_scale_factor=2"),
            (self.decoder_type in (1, 2), "decoder_type 1 or 2"),
            (self.skip_integration_type in ("concat", "add"), "skip_integration_type 'concat' or 'add'"),
            (self.number_resnet_downward_block >= 1 and self.number_resnet_upward_block >= 1,
             "at least one block per layer"),
        ]
        bad = [msg for ok, msg in checks if not ok]
        return None if not bad else "diffsci_amd ADM supports: " + "; ".join(bad)


class _Block(torch.nn.Module):
    """ADMBaseBlock parameters (adm.py:262-287); sample in {None, 'down', 'up'}."""

    def __init__(self, cin, cout, cembed, sample=None, has_attn=False, circular=False,
                 norms=("GroupLN", "GroupRMS"), affine=True):
        super().__init__()
        self.cin, self.cout, self.sample = cin, cout, sample
        # make_norm_layers, adm.py:385-406 (num_groups = 1): GroupNorm(1, C) or GroupRMSNorm(1, C) in either slot
        self.norm1 = torch.nn.GroupNorm(1, cin, affine=affine) if norms[0] == "GroupLN" else _AffineHolder(cin, affine)
        self.norm2 = torch.nn.GroupNorm(1, cout, affine=affine) if norms[1] == "GroupLN" else _AffineHolder(cout, affine)
        self.kinds = tuple(0 if n == "GroupLN" else 1 for n in norms)
        self.conv1 = make_conv(cin, cout, 3, circular)          # conv_fn, adm.py:427-443
        self.conv2 = make_conv(cout, cout, 3, circular)
        self.embed_linear = torch.nn.Linear(cembed, 2 * cout)
        self.convresidual = make_conv(cin, cout, 1, circular)
        if has_attn:
            self.attn = _Attn(cout)


class ADMBaseBlock(torch.nn.Module):
    """One ADM residual block as a module of its own -- the reference's ADMBaseBlock / ADMEncoderBlock / ADMDecoderBlock
    (adm.py:218-540), which its tests drive directly, on 2-D fields AND 3-D volumes (tests/test_adm.py:7-70):

        y = norm1(x) -> SiLU -> [AvgPool(2) | nearest x2] -> conv1 -> norm2 -> * te1 + te2 -> SiLU -> conv2
            [+ convresidual(resample(x))] [-> attention]              with (te1, te2) = chunk(embed_linear(te), 2)

    Same constructor arguments, defaults and state_dict keys.  Every tensor operation is a HIP launch (eager, standalone
    norm kernels); the whole-network class ``ADM`` folds the norms into the convolutions and is captured as a graph.
    Volumes: 3x3x3 convolutions as three 2-D matrix-core launches per convolution (ops.conv3d_mfma) or the direct
    kernel for thin layers, the group-1 statistics over (C, D, H, W), AvgPool3d / nearest resampling by ds_avgpool3d /
    the convolution loader / ds_upsample3d, attention over the flattened voxels.  (The reference's full ``ADM`` cannot
    run with dimension=3 -- its stem is a Conv2d, adm.py:190 -- so only the blocks exist in 3-D there too.)"""

    def __init__(self, channels_in: int, channels_out: int, channels_embed: int, channels_skip: int | None = None,
                 conv_type: str = 'default', image_sample: str | None = None, has_residual: bool = False,
                 has_attn: bool = False, first_norm: str = 'GroupLN', second_norm: str = 'GroupRMS',
                 affine_norm: bool = True, dimension: int = 2, num_groups: int = 1, pdrop: float = 0.0,
                 image_sample_type: str | None = None, image_sample_factor: int = 2, attn_type: str = 'default',
                 attn_heads: int = 1, attn_residual: bool = True, skip_integration_type: str = 'concat'):
        super().__init__()
        bad = []
        if dimension not in (2, 3):
            bad.append("dimension 2 or 3")
        if conv_type not in ("default", "circular"):
            bad.append("conv_type 'default' or 'circular'")
        if first_norm not in ("GroupLN", "GroupRMS") or second_norm not in ("GroupLN", "GroupRMS"):
            bad.append("norms 'GroupLN' or 'GroupRMS'")
        if num_groups != 1 or image_sample_factor != 2 or attn_heads != 1 or attn_type != "default":
            bad.append("num_groups=1, image_sample_factor=2, one default attention head")
        if image_sample not in (None, "downsample", "upsample"):
            bad.append("image_sample None, 'downsample' or 'upsample'")
        if image_sample == "downsample" and image_sample_type not in (None, "avg"):
            bad.append("average pooling")
        if image_sample == "upsample" and image_sample_type not in (None, "nearest"):
            bad.append("nearest upsampling")
        if skip_integration_type not in ("concat", "add"):
            bad.append("skip_integration_type 'concat' or 'add'")
        if bad:
            raise NotImplementedError("diffsci_amd ADM blocks support: " + "; ".join(bad))
        self.channels_in, self.channels_out, self.channels_embed = channels_in, channels_out, channels_embed
        self.channels_skip, self.dimension, self.image_sample = channels_skip, dimension, image_sample
        self.has_residual, self.has_attn, self.attn_residual = has_residual, has_attn, attn_residual
        self.skip_integration_type, self.pdrop = skip_integration_type, pdrop
        cin = channels_in + channels_skip if (channels_skip and skip_integration_type == "concat") else channels_in
        self.channels_in_modified = cin
        circ = conv_type == "circular"
        self.circular = circ
        self.norm1 = torch.nn.GroupNorm(1, cin, affine=affine_norm) if first_norm == "GroupLN" else _AffineHolder(cin, affine_norm)
        self.norm2 = (torch.nn.GroupNorm(1, channels_out, affine=affine_norm) if second_norm == "GroupLN"
                      else _AffineHolder(channels_out, affine_norm))
        self.kinds = (0 if first_norm == "GroupLN" else 1, 0 if second_norm == "GroupLN" else 1)
        self.conv1 = make_conv(cin, channels_out, 3, circ, True, dimension)
        self.conv2 = make_conv(channels_out, channels_out, 3, circ, True, dimension)
        self.embed_linear = torch.nn.Linear(channels_embed, 2 * channels_out)
        if has_residual:
            self.convresidual = make_conv(cin, channels_out, 1, circ, True, dimension)
        if has_attn:
            self.attn = _Attn(channels_out)
        self.conv_precision = "fp16x3"
        self._packed, self._packed_sig = None, None

    def _packs(self):
        convs = [self.conv1, self.conv2] + ([self.convresidual] if self.has_residual else [])
        tracked = [m.weight for m in convs] + ([self.attn.mhattn.in_proj_weight, self.attn.mhattn.out_proj.weight] if self.has_attn else [])
        sig = (self.conv_precision,) + tuple((t.data_ptr(), t._version) for t in tracked)
        if self._packed is not None and sig == self._packed_sig:
            return self._packed
        pk = {}
        with torch.no_grad():
            for m in (self.conv1, self.conv2):
                w = m.weight.detach()
                if self.dimension == 2:
                    pk[id(m)] = ops.pack_conv(w, self.conv_precision, upsampled=False)
                elif self.conv_precision == "fp16x3" and min(w.shape[0], w.shape[1]) > 4:
                    pk[id(m)] = ops.pack_conv3d(w)
            prec = "fp16x3" if self.conv_precision == "fp16x3" else "fp32"
            if self.has_residual:
                w = self.convresidual.weight.detach()
                pk[id(self.convresidual)] = ops.pack_conv(w.reshape(w.shape[0], w.shape[1], 1, 1).contiguous(), prec)
            if self.has_attn:
                m, E = self.attn.mhattn, self.attn.mhattn.embed_dim
                pk["in"] = ops.pack_conv(m.in_proj_weight.detach().reshape(3 * E, E, 1, 1), prec)
                pk["out"] = ops.pack_conv(m.out_proj.weight.detach().reshape(E, E, 1, 1), prec)
        self._packed, self._packed_sig = pk, sig
        return pk

    def _conv3(self, m, x, pk, up=False, res1=None):
        """3x3(x3) 'same' convolution of the block (with the nearest x2 upsampling in its loader when up)."""
        mode = DS_LOAD_UPSAMPLE2 if up else DS_LOAD_PLAIN
        if self.dimension == 2:
            return ops.conv(x, pk[id(m)], bias=m.bias, circular=self.circular, load_mode=mode, res1=res1)
        if id(m) in pk:
            return ops.conv3d_mfma(x, pk[id(m)], bias=m.bias, circular=self.circular, load_mode=mode, res1=res1)
        return ops.conv3d(x, m.weight, bias=m.bias, circular=self.circular, load_mode=mode, res1=res1)

    @ops.device_guard
    def forward(self, x, te, skip=None):
        """adm.py:292-313.  x [B, Cin, (D,) H, W]; te [B or 1, Cembed]; skip [B, Cskip, ...] when the block has one."""
        if self.training and self.pdrop:
            raise NotImplementedError("dropout in training mode is outside the HIP sampling path: call .eval()")
        ops.require_device(x, "x")
        if x.dim() != 2 + self.dimension:
            raise ValueError(f"a dimension={self.dimension} block takes {2 + self.dimension}-D tensors")
        x = x.contiguous()
        if self.channels_skip:
            if skip is None:
                raise ValueError("this block integrates a skip tensor")
            x = ops.concat2(x, skip.contiguous()) if self.skip_integration_type == "concat" else ops.add(x, skip.contiguous())
        B, Ci = x.shape[0], x.shape[1]
        if Ci != self.channels_in_modified:
            raise ValueError(f"expected {self.channels_in_modified} input channels, got {Ci}")
        down, up = self.image_sample == "downsample", self.image_sample == "upsample"
        pk = self._packs()
        k1, k2 = self.kinds
        Co = self.channels_out

        def v4(t):                                   # the (C, spatial...) kernels see volumes as [B, C, D*H, W]
            return t if t.dim() == 4 else t.view(t.shape[0], t.shape[1], -1, t.shape[-1])

        def pool(t):                                 # AvgPool(2) of a field or a volume
            if t.dim() == 4:
                return ops.gnorm1_apply(t, None, None, None, 2, pool=True)
            return ops.avgpool3d(t)

        film = ops.linear(te.to(x).contiguous(), self.embed_linear.weight, self.embed_linear.bias)      # [B or 1, 2*Cout]
        # first_block (adm.py:315-322): norm1 -> act -> resample -> conv1
        st = ops.gnorm1_stats(x, k1, eps=1e-5)
        fuse_pool = down and self.dimension == 2
        a = ops.gnorm1_apply(v4(x), st, self.norm1.weight, self.norm1.bias, k1, pool=fuse_pool)
        if down and self.dimension == 3:
            a = ops.avgpool3d(a.view(x.shape))
        elif self.dimension == 3:
            a = a.view(x.shape)
        y = self._conv3(self.conv1, a, pk, up=up)
        # norm2 -> FiLM -> act -> conv2 (adm.py:306-308,324-329)
        st2 = ops.gnorm1_stats(y, k2, eps=1e-5)
        a2 = ops.gnorm1_apply(v4(y), st2, self.norm2.weight, self.norm2.bias, k2, film=film).view(y.shape)
        r = None
        if self.has_residual:                        # convresidual(resample(x)), adm.py:345-349
            xr = pool(x) if down else x
            m = self.convresidual
            if up and self.dimension == 2:           # nearest x2 in the 1x1 convolution's loader
                r = ops.conv(xr, pk[id(m)], bias=m.bias, load_mode=DS_LOAD_UPSAMPLE2)
            else:
                r = ops.conv(v4(xr), pk[id(m)], bias=m.bias).view((B, Co) + tuple(xr.shape[2:]))
                if up:                               # a 1x1x1 convolution commutes with nearest upsampling: project at low resolution
                    r = ops.upsample3d(r)
        out = self._conv3(self.conv2, a2, pk, res1=r)
        if self.has_attn:                            # N-dimensional attention over the flattened positions, attention.py:67-102
            E, L = Co, out.numel() // (B * Co)
            mh = self.attn.mhattn
            qkv = ops.conv(v4(out), pk["in"], bias=mh.in_proj_bias)
            o = ops.attention(qkv.view(B, 3 * E, L), E, precision=self.conv_precision)
            out = ops.conv(o.view(v4(out).shape), pk["out"], bias=mh.out_proj.bias,
                           res1=v4(out) if self.attn_residual else None).view(out.shape)
        return out


class ADMEncoderBlock(ADMBaseBlock):
    """adm.py:455-498."""

    def __init__(self, channels_in: int, channels_out: int, channels_embed: int, conv_type: str = 'default',
                 has_downsample: bool = False, has_residual: bool = False, has_attn: bool = False,
                 first_norm: str = 'GroupLN', second_norm: str = 'GroupRMS', dimension: int = 2, num_groups: int = 1,
                 pdrop: float = 0.0, downsample_type: str = 'avg', downsample_factor: int = 2, attn_type: str = 'default',
                 attn_heads: int = 1, attn_residual: bool = True):
        super().__init__(channels_in, channels_out, channels_embed, channels_skip=None, conv_type=conv_type,
                         image_sample='downsample' if has_downsample else None, has_residual=has_residual,
                         has_attn=has_attn, first_norm=first_norm, second_norm=second_norm, dimension=dimension,
                         num_groups=num_groups, pdrop=pdrop, image_sample_type=downsample_type,
                         image_sample_factor=downsample_factor, attn_type=attn_type, attn_heads=attn_heads,
                         attn_residual=attn_residual, skip_integration_type='concat')


class ADMDecoderBlock(ADMBaseBlock):
    """adm.py:498-540."""

    def __init__(self, channels_in: int, channels_out: int, channels_embed: int, channels_skip: int | None = None,
                 conv_type: str = 'default', has_upsample: bool = False, has_residual: bool = False,
                 has_attn: bool = False, first_norm: str = 'GroupLN', second_norm: str = 'GroupRMS', dimension: int = 2,
                 num_groups: int = 1, pdrop: float = 0.0, upsample_type: str = 'nearest', upsample_factor: int = 2,
                 attn_type: str = 'default', attn_heads: int = 1, attn_residual: bool = True,
                 skip_integration_type: str = 'concat'):
        super().__init__(channels_in, channels_out, channels_embed, channels_skip=channels_skip, conv_type=conv_type,
                         image_sample='upsample' if has_upsample else None, has_residual=has_residual, has_attn=has_attn,
                         first_norm=first_norm, second_norm=second_norm, dimension=dimension, num_groups=num_groups,
                         pdrop=pdrop, image_sample_type=upsample_type, image_sample_factor=upsample_factor,
                         attn_type=attn_type, attn_heads=attn_heads, attn_residual=attn_residual,
                         skip_integration_type=skip_integration_type)


class ADMEncoderLayer(torch.nn.Module):
    """adm.py:526-598: nblocks encoder blocks, the last one widening and down-sampling; returns (x, skip)."""

    def __init__(self, channels_in: int, channels_out: int, channels_embed: int, nblocks: int = 2, conv_type: str = 'default',
                 has_residual: bool = True, has_attn: bool = False, first_norm: str = 'GroupLN', second_norm: str = 'GroupRMS',
                 dimension: int = 2, num_groups: int = 1, pdrop: float = 0.0, downsample_type: str = 'avg',
                 downsample_factor: int = 2, attn_type: str = 'default', attn_heads: int = 1, attn_residual: bool = True):
        super().__init__()
        self.channels_in, self.channels_out, self.channels_embed, self.nblocks = channels_in, channels_out, channels_embed, nblocks
        self.input_blocks = torch.nn.ModuleList([
            ADMEncoderBlock(channels_in, channels_in if i != nblocks - 1 else channels_out, channels_embed, conv_type=conv_type,
                            has_downsample=i == nblocks - 1, has_residual=has_residual, has_attn=has_attn, first_norm=first_norm,
                            second_norm=second_norm, dimension=dimension, num_groups=num_groups, pdrop=pdrop,
                            downsample_type=downsample_type, downsample_factor=downsample_factor, attn_type=attn_type,
                            attn_heads=attn_heads, attn_residual=attn_residual) for i in range(nblocks)])

    def forward(self, x, te):
        for block in self.input_blocks:
            x = block(x, te)
        return x, x                                      # the reference hands out a clone; nothing here writes in place


class ADMEncoder(torch.nn.Module):
    """adm.py:602-688."""

    def __init__(self, model_channels: int, channels_embed: int, channels_mult: list[int] = [1, 2, 4],
                 nblocks_per_layer: int | list[int] = 2, conv_type: str = 'default', has_residual: bool = True,
                 has_attn: bool | list[bool] = False, first_norm: str = 'GroupLN', second_norm: str = 'GroupRMS',
                 dimension: int = 2, num_groups: int = 1, pdrop: float = 0.0, downsample_type: str = 'avg',
                 downsample_factor: int | list[int] = 2, attn_type: str = 'default', attn_heads: int = 1,
                 attn_residual: bool = True):
        super().__init__()
        self.model_channels, self.channels_mult, self.channels_embed = model_channels, channels_mult, channels_embed
        n = self.nlayers
        nblocks_per_layer = nblocks_per_layer if isinstance(nblocks_per_layer, list) else [nblocks_per_layer] * n
        downsample_factor = downsample_factor if isinstance(downsample_factor, list) else [downsample_factor] * n
        has_attn = has_attn if isinstance(has_attn, list) else [has_attn] * n
        assert len(nblocks_per_layer) == n and len(downsample_factor) == n
        self.layers = torch.nn.ModuleList([
            ADMEncoderLayer(self.channels_in[i], self.channels_outs[i], channels_embed, nblocks=nblocks_per_layer[i],
                            conv_type=conv_type, has_residual=has_residual, has_attn=has_attn[i], first_norm=first_norm,
                            second_norm=second_norm, dimension=dimension, num_groups=num_groups, pdrop=pdrop,
                            downsample_type=downsample_type, downsample_factor=downsample_factor[i], attn_type=attn_type,
                            attn_heads=attn_heads, attn_residual=attn_residual) for i in range(n)])

    def forward(self, x, te):
        intermediate_outputs = [x]
        for layer in self.layers:
            x, xskip = layer(x, te)
            intermediate_outputs.append(xskip)
        return x, intermediate_outputs

    @property
    def channels_in(self):
        return [self.model_channels * i for i in self.channels_mult[:-1]]

    @property
    def channels_outs(self):
        return [self.model_channels * i for i in self.channels_mult[1:]]

    @property
    def nlayers(self):
        return len(self.channels_mult) - 1


class ADMDecoderLayer1(torch.nn.Module):
    """adm.py:690-776: the skip joins once, in front of the layer."""

    def __init__(self, channels_in: int, channels_out: int, channels_embed: int, channels_skip: int, nblocks: int = 2,
                 conv_type: str = 'default', has_residual: bool = True, has_attn: bool = False, first_norm: str = 'GroupLN',
                 second_norm: str = 'GroupRMS', dimension: int = 2, num_groups: int = 1, pdrop: float = 0.0,
                 upsample_factor: int = 2, attn_type: str = 'default', attn_heads: int = 1, attn_residual: bool = True,
                 skip_integration_type: str = 'concat'):
        super().__init__()
        self.skip_integration_type = skip_integration_type
        cin = channels_in + channels_skip if skip_integration_type == 'concat' else channels_in
        self.input_blocks = torch.nn.ModuleList([
            ADMDecoderBlock(cin, cin if i != nblocks - 1 else channels_out, channels_embed, channels_skip=None,
                            conv_type=conv_type, has_upsample=i == nblocks - 1, has_residual=has_residual, has_attn=has_attn,
                            first_norm=first_norm, second_norm=second_norm, dimension=dimension, num_groups=num_groups,
                            pdrop=pdrop, upsample_factor=upsample_factor, attn_type=attn_type, attn_heads=attn_heads,
                            attn_residual=attn_residual) for i in range(nblocks)])

    @ops.device_guard
    def forward(self, x, te, skip):
        if self.skip_integration_type == 'concat':
            xh = ops.concat2(x.contiguous(), skip.contiguous())
        elif self.skip_integration_type == 'add':
            xh = ops.add(x.contiguous(), skip.contiguous())
        else:
            raise ValueError(f"Invalid skip integration type {self.skip_integration_type}")
        for block in self.input_blocks:
            xh = block(xh, te)
        return xh


class ADMDecoderLayer2(torch.nn.Module):
    """adm.py:777-852: every block of the layer integrates the same skip again."""

    def __init__(self, channels_in: int, channels_out: int, channels_embed: int, channels_skip: int, nblocks: int = 2,
                 conv_type: str = 'default', has_residual: bool = True, has_attn: bool = False, first_norm: str = 'GroupLN',
                 second_norm: str = 'GroupRMS', dimension: int = 2, num_groups: int = 1, pdrop: float = 0.0,
                 upsample_factor: int = 2, attn_type: str = 'default', attn_heads: int = 1, attn_residual: bool = True,
                 skip_integration_type: str = 'concat'):
        super().__init__()
        self.input_blocks = torch.nn.ModuleList([
            ADMDecoderBlock(channels_in, channels_in if i != nblocks - 1 else channels_out, channels_embed,
                            channels_skip=channels_skip, conv_type=conv_type, has_upsample=i == nblocks - 1,
                            has_residual=has_residual, has_attn=has_attn, first_norm=first_norm, second_norm=second_norm,
                            dimension=dimension, num_groups=num_groups, pdrop=pdrop, upsample_factor=upsample_factor,
                            attn_type=attn_type, attn_heads=attn_heads, attn_residual=attn_residual,
                            skip_integration_type=skip_integration_type) for i in range(nblocks)])

    def forward(self, x, te, skip):
        for block in self.input_blocks:
            x = block(x, te, skip)
        return x


class ADMDecoder(torch.nn.Module):
    """adm.py:853-956."""

    def __init__(self, model_channels: int, channels_embed: int, channels_mult: list[int] = [4, 2, 1],
                 nblocks_per_layer: int | list[int] = 2, conv_type: str = 'default', has_residual: bool = True,
                 has_attn: bool | list[bool] = False, first_norm: str = 'GroupLN', second_norm: str = 'GroupRMS',
                 dimension: int = 2, num_groups: int = 1, pdrop: float = 0.0, upsample_factor: int | list[int] = 2,
                 attn_type: str = 'default', attn_heads: int = 1, attn_residual: bool = True,
                 skip_integration_type: str = 'concat', decoder_type: int = 1):
        super().__init__()
        self.model_channels, self.channels_mult, self.channels_embed = model_channels, channels_mult, channels_embed
        self.decoder_type = decoder_type
        n = self.nlayers
        nblocks_per_layer = nblocks_per_layer if isinstance(nblocks_per_layer, list) else [nblocks_per_layer] * n
        upsample_factor = upsample_factor if isinstance(upsample_factor, list) else [upsample_factor] * n
        has_attn = has_attn if isinstance(has_attn, list) else [has_attn] * n
        assert len(nblocks_per_layer) == n and len(upsample_factor) == n and len(has_attn) == n
        self.layers = torch.nn.ModuleList([
            self.decoder_fn(channels_in=self.channels_ins[i], channels_out=self.channels_outs[i], channels_embed=channels_embed,
                            channels_skip=self.channels_ins[i], nblocks=nblocks_per_layer[i], conv_type=conv_type,
                            has_residual=has_residual, has_attn=has_attn[i], first_norm=first_norm, second_norm=second_norm,
                            dimension=dimension, num_groups=num_groups, pdrop=pdrop, upsample_factor=upsample_factor[i],
                            attn_type=attn_type, attn_heads=attn_heads, attn_residual=attn_residual,
                            skip_integration_type=skip_integration_type) for i in range(n)])

    def forward(self, x, te, intermediate_outputs, pop=True):
        for i, layer in enumerate(self.layers):
            h = intermediate_outputs.pop() if pop else intermediate_outputs[-(i + 1)]
            x = layer(x, te, h)
        return x

    @property
    def decoder_fn(self):
        if self.decoder_type == 1:
            return ADMDecoderLayer1
        if self.decoder_type == 2:
            return ADMDecoderLayer2
        raise ValueError(f"Invalid decoder type {self.decoder_type}")

    @property
    def channels_ins(self):
        return [self.model_channels * i for i in self.channels_mult[:-1]]

    @property
    def channels_outs(self):
        return [self.model_channels * i for i in self.channels_mult[1:]]

    @property
    def nlayers(self):
        return len(self.channels_mult) - 1


class ADMMiddleBlock(torch.nn.Module):
    """adm.py:958-1011: nblocks same-width encoder blocks, attention in all but the last by default."""

    def __init__(self, channels: int, channels_embed: int, nblocks: int = 2, conv_type: str = 'default',
                 has_residual: bool = True, has_attn: bool | list[bool] | str = 'default', first_norm: str = 'GroupLN',
                 second_norm: str = 'GroupRMS', dimension: int = 2, num_groups: int = 1, pdrop: float = 0.0,
                 attn_type: str = 'default', attn_heads: int = 1, attn_residual: bool = True):
        super().__init__()
        if isinstance(has_attn, str):
            if has_attn != 'default':
                raise ValueError(f"Invalid has_attn {has_attn}")
            has_attn = [True] * (nblocks - 1) + [False]
        if not isinstance(has_attn, list):
            has_attn = [has_attn] * nblocks
        assert len(has_attn) == nblocks
        self.middle_blocks = torch.nn.ModuleList([
            ADMEncoderBlock(channels, channels, channels_embed, conv_type=conv_type, has_downsample=False,
                            has_residual=has_residual, has_attn=has_attn[i], first_norm=first_norm, second_norm=second_norm,
                            dimension=dimension, num_groups=num_groups, pdrop=pdrop, attn_type=attn_type,
                            attn_heads=attn_heads, attn_residual=attn_residual) for i in range(nblocks)])

    def forward(self, x, te):
        for block in self.middle_blocks:
            x = block(x, te)
        return x


class ADMTimeEmbedding(torch.nn.Module):
    """adm.py:1014-1053: SiLU(mlp(fourier(t)) + ye).  State-dict keys projection.W, mlp.{0,2}.{weight,bias}."""

    def __init__(self, embed_dim: int, output_dim: int, projection_scale: float = 30.0):
        super().__init__()
        self.projection = _Fourier(embed_dim, projection_scale)
        self.mlp = torch.nn.Sequential(torch.nn.Linear(embed_dim, output_dim), torch.nn.Identity(),
                                       torch.nn.Linear(output_dim, output_dim))

    @ops.device_guard
    def forward(self, t, ye=None):
        ops.require_device(t, "t")
        if ye is not None and ye.shape[0] not in (1, t.numel()):
            raise ValueError("conditional embedding batch must be 1 or match t")
        f = ops.fourier_features(t.contiguous(), self.projection.W)
        h = ops.linear(f, self.mlp[0].weight, self.mlp[0].bias, act=1)
        if ye is None:
            return ops.linear(h, self.mlp[2].weight, self.mlp[2].bias, act=1)
        h = ops.linear(h, self.mlp[2].weight, self.mlp[2].bias, act=0)
        return ops.add_act(h, ye.to(h).contiguous(), act=1)


class _Layer(torch.nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.input_blocks = torch.nn.ModuleList(blocks)


class _Layers(torch.nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = torch.nn.ModuleList(layers)


class _Middle(torch.nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.middle_blocks = torch.nn.ModuleList(blocks)


class ADM(torch.nn.Module):
    def __init__(self, config: ADMConfig, conditional_embedding: torch.nn.Module | None = None):
        super().__init__()
        why = config.unsupported_reason()
        if why:
            raise NotImplementedError(why)
        self.config = config
        self.conditional_embedding = conditional_embedding
        mc, ce = config.model_channels, config.output_embed_dim
        mult = config.extended_channel_expansion
        self.time_embedding = ADMTimeEmbedding(config.time_embed_dim, ce, config.time_projection_scale)
        circ = config.convolution_type == "circular"           # the blocks' convolutions; input/output layers stay zero-padded
        # ADMConfig.affine_norm never reaches the blocks in the reference (ADMEncoder / ADMMiddleBlock / ADMDecoder do
        # not forward it, adm.py:455-520,540-834): the norms are always affine, and checkpoints carry their weights
        nk = dict(norms=(config.first_resblock_norm, config.second_resblock_norm))
        nb = config.number_resnet_downward_block
        enc = []
        for i in range(len(mult) - 1):                                   # adm.py:566-592
            cin, cout = mc * mult[i], mc * mult[i + 1]
            enc.append(_Layer([_Block(cin, cin, ce, circular=circ, **nk) for _ in range(nb - 1)] +
                              [_Block(cin, cout, ce, "down", circular=circ, **nk)]))
        self.encoder = _Layers(enc)
        cm = config.middle_channel
        self.middle_block = _Middle([_Block(cm, cm, ce, None, a, circular=circ, **nk) for a in config.middle_block_attn_config])
        rmult = mult[::-1]
        nb = config.number_resnet_upward_block
        dec = []
        for i in range(len(mult) - 1):                                   # adm.py:731-762
            cin, cout = mc * rmult[i], mc * rmult[i + 1]
            cb = 2 * cin if config.skip_integration_type == "concat" else cin
            # decoder_type 1 (ADMDecoderLayer1, adm.py:690-776): the skip joins once, in front of the layer;
            # decoder_type 2 (ADMDecoderLayer2, :777-852): every block of the layer integrates the same skip again
            cmid = cb if config.decoder_type == 1 else cin
            dec.append(_Layer([_Block(cb, cmid, ce, circular=circ, **nk) for _ in range(nb - 1)] +
                              [_Block(cb, cout, ce, "up", circular=circ, **nk)]))
        self.decoder = _Layers(dec)
        self.input_layer = torch.nn.Conv2d(config.input_channels, mc, 3, padding="same")
        self.output_layer = torch.nn.Conv2d(mc, config.output_channels, 3, padding="same")
        self.conv_precision = "fp16x3"       # see PUNetG.conv_precision
        self.auto_precision = True           # see PUNetG.auto_precision
        # see PUNetG.fuse_norm / fuse_max_cot.  Measured on MI355X at config 3: folding the norms of the layers with
        # up to 256 channels (1 GiB .. 134 MB tensors: the standalone pass is HBM-bound) gives 8.28 samples/s against
        # 8.05 with standalone kernels everywhere and 8.22 with a 128-channel limit; folding every layer is slower
        # (the 512-1024-channel layers would redo the activation once per 64-channel tile).
        self.fuse_norm = True
        self.fuse_max_cot = 4
        self.norm_images = os.environ.get("DIFFSCI_NORM_IMAGES", "1") != "0"      # see PUNetG.norm_images
        # standalone norms take their statistics from the producer's tile statistics (as the folded ones do) instead of
        # a pass over the tensor, whenever the producer left them
        self.tile_stats_norms = os.environ.get("DIFFSCI_TILE_STATS_NORMS", "1") != "0"
        self._packed = None
        self._packed_sig = None
        self._ws = _Workspace()
        self._am = None              # the amax arena of the forward pass in flight (see PUNetG.forward_with_shifts)
        self._window_cache = {}
        self.exact_input_layer = False   # see PUNetG.exact_input_layer

    # ------------------------------------------------------------------ reference surface
    def export_description(self) -> dict[str, Any]:
        cemb = self.conditional_embedding
        cemb_args = cemb.export_description() if getattr(cemb, "export_description", None) else None
        return dict(config=self.config.export_description(), conditional_embedding_args=cemb_args,
                    has_conditional_embedding=cemb is not None)

    def set_conditional_embedding(self, conditional_embedding: torch.nn.Module | None = None):
        self.conditional_embedding = conditional_embedding

    @ops.device_guard
    def forward(self, x, t, y=None):
        """adm.py:199-216.  Top-level call: guarded (see PUNetG.forward)."""
        out = self.forward_unguarded(x, t, y)
        if precision.needs_escalation(self, out, x):
            precision.escalate(self)
            out = self.forward_unguarded(x, t, y)
        return out

    @ops.device_guard
    def forward_unguarded(self, x, t, y=None):
        ops.require_device(x, "x")
        te = self.embed_time(t.reshape(-1).to(x), self.embed_condition(y))
        shifts = self.time_shifts(te)
        return self.forward_with_shifts(x.contiguous(), shifts, row=None)

    # ------------------------------------------------------------------ conditioning
    def embed_condition(self, y):
        if y is None:
            return None
        if self.conditional_embedding is None:
            raise ValueError("y was given but the network has no conditional_embedding")
        ye = self.conditional_embedding(y)
        if ye.ndim != 2:
            raise NotImplementedError("spatial conditional embeddings are not implemented")
        return ye.to(torch.float32).contiguous()

    def embed_time(self, t, ye=None):
        """ADMTimeEmbedding.forward (adm.py:1047-1053) -> [M, output_embed_dim]."""
        return self.time_embedding(t, ye)

    def time_shifts(self, te):
        """Per-block embed_linear(te) (adm.py:333-334): list of [M, 2*C_out] FiLM rows."""
        return [ops.linear(te, b.embed_linear.weight, b.embed_linear.bias, act=0) for b in self._blocks()]

    def _blocks(self):
        for lay in self.encoder.layers:
            yield from lay.input_blocks
        yield from self.middle_block.middle_blocks
        for lay in self.decoder.layers:
            yield from lay.input_blocks

    # ------------------------------------------------------------------ weights
    def packed_weights(self):
        blocks = list(self._blocks())
        convs = [self.input_layer, self.output_layer]
        for b in blocks:
            convs += [b.conv1, b.conv2, b.convresidual]
        attns = [b.attn for b in blocks if hasattr(b, "attn")]
        sig = (self.conv_precision, getattr(self, "upsample_parity", True), self.exact_input_layer) + tuple((m.weight.data_ptr(), m.weight._version) for m in convs) + tuple(
            (a.mhattn.in_proj_weight.data_ptr(), a.mhattn.in_proj_weight._version) for a in attns)
        if self._packed is not None and sig == self._packed_sig:
            return self._packed
        pk = {}
        with torch.no_grad():
            ups = {id(b.conv1) for b in blocks if b.sample == "up"} if getattr(self, "upsample_parity", True) else set()
            for m in convs:
                pk[id(m)] = ops.pack_conv(m.weight.detach(), self.conv_precision, upsampled=id(m) in ups)
            if self.conv_precision == "fp16x3":
                pk[(id(self.input_layer), "wmax")] = self.input_layer.weight.detach().abs().amax(dim=(0, 2, 3)).contiguous()
                if self.exact_input_layer:
                    pk[(id(self.input_layer), "exact")] = ops.pack_conv(self.input_layer.weight.detach(), "fp32")
            for a in attns:
                E = a.mhattn.embed_dim
                prec = "fp16x3" if self.conv_precision == "fp16x3" else "fp32"
                pk[(id(a), "in")] = ops.pack_conv(a.mhattn.in_proj_weight.detach().reshape(3 * E, E, 1, 1), prec)
                pk[(id(a), "out")] = ops.pack_conv(a.mhattn.out_proj.weight.detach().reshape(E, E, 1, 1), prec)
        self._packed, self._packed_sig = pk, sig
        return pk

    # ------------------------------------------------------------------ the network
    def _amax_kw(self, m_or_pack, pk=None, **kw):
        """in_amax / out_amax are arguments of the fp16x3 kernels only."""
        p = m_or_pack if pk is None else pk[id(m_or_pack)]
        return kw if p.kind == "fp16x3" else {}

    def _conv(self, m, x, pk, in_amax=None, out_amax=None, **kw):
        return ops.conv(x, pk[id(m)], bias=m.bias, circular=isinstance(m, _CircConv),
                        **self._amax_kw(m, pk, in_amax=in_amax, out_amax=out_amax), **kw)

    def _raw_amax(self, x, xa):
        """in_amax of a launch that reads the raw tensor x: the row its producer left, else a reduction into a row."""
        if self._am is None:
            return None
        return xa if xa is not None else self._am.of(x)

    def _normed_amax(self, blk, a):
        """in_amax of a standalone norm (+ FiLM) + SiLU output: inside the fp16x3 window for affine parameters of ordinary size."""
        if self._am is None or precision.norms_in_window(self._window_cache, id(blk), (blk.norm1, blk.norm2)):
            return ops.NORMALISED
        return self._am.of(a)

    def _fused(self):
        return self.fuse_norm and self.conv_precision == "fp16x3"

    def _stats_buf(self, ws, B, C, H, W, dev):
        if not self._fused():
            return None
        return ws.take((B, C, ops.conv_tile_count(H, W), 4), dev)

    def _norm_images_ok(self, conv, pk, Cin):
        """A standalone norm may hand this convolution pre-split images: 3x3 fp16x3 packing, zero padding, an even number
        of 16-channel chunks."""
        p = pk[id(conv)]
        return (self.norm_images and not isinstance(conv, _CircConv) and p.kind == "fp16x3" and p.ks == 3 and p.subs is None
                and ((Cin + 15) // 16) % 2 == 0)

    def _block(self, blk, x, film, pk, ws, xs=None, want_stats=True, xa=None, out_amax=None):
        """ADMBaseBlock.forward (adm.py:292-349); returns (fresh buffer, its tile statistics); x untouched.
        xs: tile statistics of x -- one buffer, or a pair when x is the channel concatenation of two
        convolution outputs -- or None (then norm1 runs as standalone kernels).  xa: the amax row of x (convresidual reads
        the raw x), out_amax: a zeroed row for the result's (see PUNetG.forward_with_shifts)."""
        B, Ci, H, W = x.shape
        dev = x.device
        down, up = blk.sample == "down", blk.sample == "up"
        k1, k2 = blk.kinds                                     # 0 GroupNorm(1, C), 1 GroupRMSNorm(1, C)
        Ho, Wo = (H // 2, W // 2) if down else ((2 * H, 2 * W) if up else (H, W))
        mode = DS_LOAD_UPSAMPLE2 if up else DS_LOAD_PLAIN
        fused = self._fused()
        fuse1 = fused and (Ci + 63) // 64 <= self.fuse_max_cot            # per layer: see PUNetG.fuse_max_cot
        fuse2 = fused and (blk.cout + 63) // 64 <= self.fuse_max_cot
        ys = self._stats_buf(ws, B, blk.cout, Ho, Wo, dev)
        # first_block: norm1 -> act -> resample -> conv1                          (adm.py:312-323)
        # not for 'down' (pooling follows the activation) nor 'up' blocks (the loader would activate every source
        # pixel four times, once per upsampled copy)
        if fuse1 and xs is not None and not down and not up:
            sa, sb = xs if isinstance(xs, tuple) else (xs, None)
            tab = ws.take((B, ops.table_channels(Ci), 4), dev)
            ops.gnorm1_table(sa, blk.norm1.weight, blk.norm1.bias, k1, Ci * H * W, stats_b=sb, eps=1e-5, out=tab)   # + the activation's exponent
            y = self._conv(blk.conv1, x, pk, load_mode=mode, prenorm=tab, tile_stats=ys,
                           out=ws.take((B, blk.cout, Ho, Wo), dev))
            ws.give(tab)
        else:                                                                     # pooling follows the activation
            Hm, Wm = (Ho, Wo) if down else (H, W)
            stats = ws.take((B, 2), dev)
            scratch = ws.take((ops.N.lib().ds_gnorm1_workspace_bytes(B) // 4,), dev)
            if xs is not None and self.tile_stats_norms:   # the producers left tile statistics: no pass over x
                sa, sb = xs if isinstance(xs, tuple) else (xs, None)
                ops.gnorm1_stats_tiles(sa, k1, Ci * H * W, stats_b=sb, eps=1e-5, stats=stats)
            else:
                ops.gnorm1_stats(x, k1, eps=1e-5, stats=stats, workspace=scratch)
            if not up and self._norm_images_ok(blk.conv1, pk, Ci):
                # the standalone norm writes the convolution's pre-split fp16 images, staged there by LDS-DMA (see punetg._res)
                img = ops.gnorm1_apply_images(x, stats, blk.norm1.weight, blk.norm1.bias, k1, pool=down,
                                              out=ws.take((ops.conv_images_floats(B, Ci, Hm, Wm),), dev))
                y = ops.conv_img(img, pk[id(blk.conv1)], B, Ci, Hm, Wm, bias=blk.conv1.bias, tile_stats=ys,
                                 out=ws.take((B, blk.cout, Ho, Wo), dev))
                ws.give(img)
            elif up and self.norm_images and not isinstance(blk.conv1, _CircConv) \
                    and ops.conv_up_img_supported(pk[id(blk.conv1)], H, W):
                # the same for the parity kernels of conv1(nearest_x2(.)): images of the low-resolution activation
                img = ops.gnorm1_apply_images(x, stats, blk.norm1.weight, blk.norm1.bias, k1,
                                              out=ws.take((ops.conv_images_floats(B, Ci, H, W),), dev))
                y = ops.conv_up_img(img, pk[id(blk.conv1)], B, Ci, H, W, bias=blk.conv1.bias, tile_stats=ys,
                                    out=ws.take((B, blk.cout, Ho, Wo), dev))
                ws.give(img)
            else:
                a = ops.gnorm1_apply(x, stats, blk.norm1.weight, blk.norm1.bias, k1, pool=down,
                                     out=ws.take((B, Ci, Hm, Wm), dev))
                y = self._conv(blk.conv1, a, pk, load_mode=mode, tile_stats=ys, out=ws.take((B, blk.cout, Ho, Wo), dev),
                               in_amax=self._normed_amax(blk, a))
                ws.give(a)
            ws.give(stats)
            ws.give(scratch)
        # residual_block: convresidual(resample(x))                               (adm.py:345-349)
        r_up = False
        raw = self._raw_amax(x, xa) if pk[id(blk.convresidual)].kind == "fp16x3" else None       # pooling / upsampling keep max |x| a bound
        if down and pk[id(blk.convresidual)].kind == "fp16x3":
            r = self._conv(blk.convresidual, x, pk, load_mode=DS_LOAD_AVGPOOL2, out=ws.take((B, blk.cout, Ho, Wo), dev), in_amax=raw)
        elif down:
            a = ops.gnorm1_apply(x, None, None, None, 2, pool=True, out=ws.take((B, Ci, Ho, Wo), dev))
            r = self._conv(blk.convresidual, a, pk, out=ws.take((B, blk.cout, Ho, Wo), dev))
            ws.give(a)
        elif up and pk[id(blk.convresidual)].kind == "fp16x3" and pk[id(blk.conv2)].kind == "fp16x3":
            # a 1x1 convolution commutes with nearest upsampling: project at low resolution (a quarter of the
            # pixels) and let conv2's epilogue add the result upsampled
            r = self._conv(blk.convresidual, x, pk, out=ws.take((B, blk.cout, H, W), dev), in_amax=raw)
            r_up = True
        else:
            r = self._conv(blk.convresidual, x, pk, load_mode=mode, out=ws.take((B, blk.cout, Ho, Wo), dev), in_amax=raw)
        # norm2 -> FiLM -> act -> conv2, + residual                               (adm.py:325-337)
        has_attn = hasattr(blk, "attn")
        os_ = self._stats_buf(ws, B, blk.cout, Ho, Wo, dev) if (want_stats and not has_attn) else None
        h3 = self._am is not None
        oa = (self._am.row() if has_attn else out_amax) if h3 else None            # conv2's result feeds the attention, or is the block's
        if fuse2:
            tab = ws.take((B, ops.table_channels(blk.cout), 4), dev)
            ops.gnorm1_table(ys, blk.norm2.weight, blk.norm2.bias, k2, blk.cout * Ho * Wo, film=film, eps=1e-5, out=tab)
            out = self._conv(blk.conv2, y, pk, res1=r, res1_upsampled=r_up, prenorm=tab, tile_stats=os_,
                             out=ws.take((B, blk.cout, Ho, Wo), dev), out_amax=oa)
            ws.give(tab)
            ws.give(ys)
            ws.give(y)
        else:
            stats = ws.take((B, 2), dev)
            scratch = ws.take((ops.N.lib().ds_gnorm1_workspace_bytes(B) // 4,), dev)
            if ys is not None and self.tile_stats_norms:
                ops.gnorm1_stats_tiles(ys, k2, blk.cout * Ho * Wo, eps=1e-5, stats=stats)
            else:
                ops.gnorm1_stats(y, k2, eps=1e-5, stats=stats, workspace=scratch)
            if self._norm_images_ok(blk.conv2, pk, blk.cout):
                img = ops.gnorm1_apply_images(y, stats, blk.norm2.weight, blk.norm2.bias, k2, film=film,
                                              out=ws.take((ops.conv_images_floats(B, blk.cout, Ho, Wo),), dev))
                out = ops.conv_img(img, pk[id(blk.conv2)], B, blk.cout, Ho, Wo, bias=blk.conv2.bias, res1=r,
                                   res1_upsampled=r_up, tile_stats=os_, out=y, out_amax=oa)
                ws.give(img)
            else:
                a2 = ops.gnorm1_apply(y, stats, blk.norm2.weight, blk.norm2.bias, k2, film=film,
                                      out=ws.take((B, blk.cout, Ho, Wo), dev))
                out = self._conv(blk.conv2, a2, pk, res1=r, res1_upsampled=r_up, tile_stats=os_, out=y,
                                 in_amax=self._normed_amax(blk, a2), out_amax=oa)
                ws.give(a2)
            ws.give(stats)
            ws.give(scratch)
            if ys is not None:
                ws.give(ys)
        ws.give(r)
        if has_attn:
            os_ = self._stats_buf(ws, B, blk.cout, Ho, Wo, dev) if want_stats else None
            out2 = self._attention(blk.attn, out, pk, ws, tile_stats=os_, in_amax=oa, out_amax=out_amax)
            ws.give(out)
            out = out2
        return out, os_

    def _attention(self, att, x, pk, ws, tile_stats=None, in_amax=None, out_amax=None):
        """TwoDimensionalAttention.forward (attention.py:67-72,82-90), channel-major; amax rows as PUNetG._attention."""
        B, E, Hh, Ww = x.shape
        L = Hh * Ww
        m = att.mhattn
        am = self._am
        h3 = am is not None and pk[(id(att), "in")].kind == "fp16x3"
        a_qkv, a_o = (am.rows(2), am.row()) if h3 else (None, None)
        akw = (lambda **kw: kw) if h3 else (lambda **kw: {})
        split = 2 * E if E % 32 == 0 else 0                           # one exponent for q and k, one for v
        qkv = ops.conv(x, pk[(id(att), "in")], bias=m.in_proj_bias, out=ws.take((B, 3 * E, Hh, Ww), x.device),
                       **akw(in_amax=self._raw_amax(x, in_amax) if h3 else None, out_amax=a_qkv if split else None, amax_split=split))
        if h3 and not split:
            ops.absmax_rows(qkv[:, :2 * E], out=a_qkv[:B])
            ops.absmax_rows(qkv[:, 2 * E:], out=a_qkv[B:])
        nws = ops.attention_workspace_floats(B, E, L, self.conv_precision)
        aws = ws.take((nws,), x.device) if nws else None
        o = ops.attention(qkv.view(B, 3 * E, L), E, out=ws.take((B, E, L), x.device),
                          precision=self.conv_precision, workspace=aws, **akw(in_amax=a_qkv, out_amax=a_o))
        if aws is not None:
            ws.give(aws)
        y = ops.conv(o.view(B, E, Hh, Ww), pk[(id(att), "out")], bias=m.out_proj.bias,
                     res1=x if self.config.attn_residual else None, tile_stats=tile_stats,
                     out=ws.take(x.shape, x.device), **akw(in_amax=a_o, out_amax=out_amax))
        ws.give(qkv)
        ws.give(o)
        return y

    def forward_with_shifts(self, x, shifts, row=None, out=None):
        """UNet body given the per-block FiLM rows (see PUNetG.forward_with_shifts)."""
        require_eval(self, self.config.dropout, self.config.cond_dropout)
        pk = self.packed_weights()
        ws = self._ws
        cfg = self.config
        B = x.shape[0]
        it = iter(range(len(shifts)))

        def film():
            s = shifts[next(it)]
            if row is not None:
                if s.dim() == 3:                       # [n_evals, B, 2C]: per-sample conditions in the planned sampler
                    return s[row]
                return s[row:row + 1]
            if s.shape[0] not in (1, B):
                raise ValueError("time embedding batch does not match x")
            return s

        dev = x.device
        H, W = x.shape[2:]

        def give(t, ts):
            ws.give(t)
            for q in (ts if isinstance(ts, tuple) else (ts,)):
                if q is not None:
                    ws.give(q)

        # activation exponents of the raw-input launches (input layer, every block's convresidual, the attention, a wide
        # output layer): rows of one arena per forward, filled by the producers' epilogues -- see PUNetG.forward_with_shifts
        h3 = self.conv_precision == "fp16x3"
        am = self._am = _AmaxArena(ws, B, dev, zero=self.exact_input_layer) if h3 else None     # else zeroed by the input layer's reduction

        def slot():
            return am.row() if h3 else None

        try:
            x_amax = (am.of_input(x, precision.input_layer_flag(self, dev), pk[(id(self.input_layer), "wmax")])
                      if (h3 and not self.exact_input_layer) else None)             # first: this launch also zeroes the arena
            ha = slot()
            if h3 and self.exact_input_layer:                                       # see PUNetG.forward_with_shifts
                hs = None
                h = ops.conv(x, pk[(id(self.input_layer), "exact")], bias=self.input_layer.bias,
                             out=ws.take((B, cfg.model_channels, H, W), dev))
                ops.absmax_rows(h, out=ha)
            else:
                hs = self._stats_buf(ws, B, cfg.model_channels, H, W, dev)
                h = self._conv(self.input_layer, x, pk, tile_stats=hs, out=ws.take((B, cfg.model_channels, H, W), dev),
                               in_amax=x_amax, out_amax=ha)
            skips = [(h, hs, ha)]                                                   # adm.py:667-675
            for lay in self.encoder.layers:
                for blk in lay.input_blocks:
                    ha2 = slot()
                    h2, hs2 = self._block(blk, h, film(), pk, ws, xs=hs, xa=ha, out_amax=ha2)
                    if not any(h is s for s, _, _ in skips):
                        give(h, hs)
                    h, hs, ha = h2, hs2, ha2
                skips.append((h, hs, ha))
            for blk in self.middle_block.middle_blocks:
                ha2 = slot()
                h2, hs2 = self._block(blk, h, film(), pk, ws, xs=hs, xa=ha, out_amax=ha2)
                if not any(h is s for s, _, _ in skips):
                    give(h, hs)
                h, hs, ha = h2, hs2, ha2
            nl = len(self.decoder.layers)

            def join(h, hs, ha, skip, sks, ska):                                     # adm.py:297-304
                if cfg.skip_integration_type == "concat":
                    hc = ops.concat2(h, skip, out=ws.take((B, h.shape[1] + skip.shape[1]) + tuple(h.shape[2:]), dev))
                    hca = ops.amax_merge(slot(), ha, ska) if h3 else None           # max over the two halves
                    return hc, ((hs, sks) if (hs is not None and sks is not None) else None), hca   # statistics of a concat are additive
                hc = ops.add(h, skip, out=ws.take(h.shape, dev))
                return hc, None, (am.of(hc) if h3 else None)

            for li, lay in enumerate(self.decoder.layers):                          # adm.py:764-774, 927-934
                skip, sks, ska = skips.pop()
                nblk = len(lay.input_blocks)
                if cfg.decoder_type == 1:
                    hc, hcs, hca = join(h, hs, ha, skip, sks, ska)
                    pending = [(h, hs)] + ([(skip, sks)] if skip is not h else [])     # statistics are read by block 0's table
                    for j, blk in enumerate(lay.input_blocks):
                        final = li == nl - 1 and j == nblk - 1                        # feeds the output layer: no norm follows
                        ha2 = slot()
                        h2, hs2 = self._block(blk, hc, film(), pk, ws, xs=hcs, want_stats=not final, xa=hca, out_amax=ha2)
                        if j == 0:
                            ws.give(hc)
                            for t, ts in pending:
                                give(t, ts)
                        else:
                            give(hc, hcs)
                        hc, hcs, hca = h2, hs2, ha2
                    h, hs, ha = hc, hcs, hca
                else:                                                               # every block joins the skip (adm.py:848-851)
                    for j, blk in enumerate(lay.input_blocks):
                        final = li == nl - 1 and j == nblk - 1
                        hc, hcs, hca = join(h, hs, ha, skip, sks, ska)
                        ha2 = slot()
                        h2, hs2 = self._block(blk, hc, film(), pk, ws, xs=hcs, want_stats=not final, xa=hca, out_amax=ha2)
                        ws.give(hc)
                        if h is not skip:
                            give(h, hs)
                        h, hs, ha = h2, hs2, ha2
                    give(skip, sks)
            for s_, ss, _ in skips:                                                  # the stem copy is never consumed
                if s_ is not h:
                    give(s_, ss)
            m = self.output_layer
            if m.out_channels <= 4:                              # see PUNetG._out_conv
                y = ops.conv_direct(h, m.weight, m.bias, out=out)
            else:
                y = self._conv(m, h, pk, out=out, in_amax=ha)
            give(h, hs)
            return y
        finally:
            if am is not None:
                am.release()
            self._am = None
