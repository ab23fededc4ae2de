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

import torch
import yaml

from ... import ops
from ..._native import DS_LOAD_AVGPOOL2, DS_LOAD_PLAIN, DS_LOAD_UPSAMPLE2
from . import precision
from .punetg import _AffineHolder, _Attn, _CircConv, _Fourier, _Workspace, make_conv, require_eval

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

    def __init__(self, **kwargs):
        unknown = set(kwargs) - set(_FIELDS)
        if unknown:
            raise TypeError(f"ADMConfig got unexpected arguments {sorted(unknown)}")
        for k, default in _FIELDS.items():
            v = kwargs.get(k, default)
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
            (self.dimension == 2, "only 2-D fields (dimension=2)"),
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


class _TimeEmbedding(torch.nn.Module):
    """ADMTimeEmbedding parameters (adm.py:1014-1045)."""

    def __init__(self, embed_dim, output_dim, scale):
        super().__init__()
        self.projection = _Fourier(embed_dim, scale)
        self.mlp = torch.nn.Sequential(torch.nn.Linear(embed_dim, output_dim), torch.nn.Identity(),
                                       torch.nn.Linear(output_dim, output_dim))


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
        self.time_embedding = _TimeEmbedding(config.time_embed_dim, ce, config.time_projection_scale)
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
        self._packed = None
        self._packed_sig = None
        self._ws = _Workspace()

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
        """adm.py:199-216."""
        ops.require_device(x, "x")
        te = self.embed_time(t.reshape(-1).to(x), self.embed_condition(y))
        shifts = self.time_shifts(te)
        out = self.forward_with_shifts(x.contiguous(), shifts, row=None)
        if precision.needs_escalation(self, out, x, te):
            precision.escalate(self)
            out = self.forward_with_shifts(x.contiguous(), shifts, row=None)
        return out

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
        te = self.time_embedding
        if ye is not None and ye.shape[0] not in (1, t.numel()):
            raise ValueError("conditional embedding batch must be 1 or match t")
        f = ops.fourier_features(t.contiguous(), te.projection.W)
        h = ops.linear(f, te.mlp[0].weight, te.mlp[0].bias, act=1)
        if ye is None:
            return ops.linear(h, te.mlp[2].weight, te.mlp[2].bias, act=1)
        h = ops.linear(h, te.mlp[2].weight, te.mlp[2].bias, act=0)
        return ops.add_act(h, ye, act=1)

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
        sig = (self.conv_precision, getattr(self, "upsample_parity", True)) + tuple((m.weight.data_ptr(), m.weight._version) for m in convs) + tuple(
            (a.mhattn.in_proj_weight.data_ptr(), a.mhattn.in_proj_weight._version) for a in attns)
        if self._packed is not None and sig == self._packed_sig:
            return self._packed
        pk = {}
        with torch.no_grad():
            ups = {id(b.conv1) for b in blocks if b.sample == "up"} if getattr(self, "upsample_parity", True) else set()
            for m in convs:
                pk[id(m)] = ops.pack_conv(m.weight.detach(), self.conv_precision, upsampled=id(m) in ups)
            for a in attns:
                E = a.mhattn.embed_dim
                prec = "fp16x3" if self.conv_precision == "fp16x3" else "fp32"
                pk[(id(a), "in")] = ops.pack_conv(a.mhattn.in_proj_weight.detach().reshape(3 * E, E, 1, 1), prec)
                pk[(id(a), "out")] = ops.pack_conv(a.mhattn.out_proj.weight.detach().reshape(E, E, 1, 1), prec)
        self._packed, self._packed_sig = pk, sig
        return pk

    # ------------------------------------------------------------------ the network
    def _conv(self, m, x, pk, **kw):
        return ops.conv(x, pk[id(m)], bias=m.bias, circular=isinstance(m, _CircConv), **kw)

    def _fused(self):
        return self.fuse_norm and self.conv_precision == "fp16x3"

    def _stats_buf(self, ws, B, C, H, W, dev):
        if not self._fused():
            return None
        return ws.take((B, C, ops.conv_tile_count(H, W), 4), dev)

    def _block(self, blk, x, film, pk, ws, xs=None, want_stats=True):
        """ADMBaseBlock.forward (adm.py:292-349); returns (fresh buffer, its tile statistics); x untouched.
        xs: tile statistics of x -- one buffer, or a pair when x is the channel concatenation of two
        convolution outputs -- or None (then norm1 runs as standalone kernels)."""
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
            ops.gnorm1_table(sa, blk.norm1.weight, blk.norm1.bias, k1, Ci * H * W, stats_b=sb, eps=1e-5, out=tab)
            y = self._conv(blk.conv1, x, pk, load_mode=mode, prenorm=tab, tile_stats=ys,
                           out=ws.take((B, blk.cout, Ho, Wo), dev))
            ws.give(tab)
        else:                                                                     # pooling follows the activation
            Hm, Wm = (Ho, Wo) if down else (H, W)
            stats = ws.take((B, 2), dev)
            scratch = ws.take((ops.N.lib().ds_gnorm1_workspace_bytes(B) // 4,), dev)
            ops.gnorm1_stats(x, k1, eps=1e-5, stats=stats, workspace=scratch)
            a = ops.gnorm1_apply(x, stats, blk.norm1.weight, blk.norm1.bias, k1, pool=down,
                                 out=ws.take((B, Ci, Hm, Wm), dev))
            y = self._conv(blk.conv1, a, pk, load_mode=mode, tile_stats=ys, out=ws.take((B, blk.cout, Ho, Wo), dev))
            ws.give(a)
            ws.give(stats)
            ws.give(scratch)
        # residual_block: convresidual(resample(x))                               (adm.py:345-349)
        r_up = False
        if down and pk[id(blk.convresidual)].kind == "fp16x3":
            r = self._conv(blk.convresidual, x, pk, load_mode=DS_LOAD_AVGPOOL2, out=ws.take((B, blk.cout, Ho, Wo), dev))
        elif down:
            a = ops.gnorm1_apply(x, None, None, None, 2, pool=True, out=ws.take((B, Ci, Ho, Wo), dev))
            r = self._conv(blk.convresidual, a, pk, out=ws.take((B, blk.cout, Ho, Wo), dev))
            ws.give(a)
        elif up and pk[id(blk.convresidual)].kind == "fp16x3" and pk[id(blk.conv2)].kind == "fp16x3":
            # a 1x1 convolution commutes with nearest upsampling: project at low resolution (a quarter of the
            # pixels) and let conv2's epilogue add the result upsampled
            r = self._conv(blk.convresidual, x, pk, out=ws.take((B, blk.cout, H, W), dev))
            r_up = True
        else:
            r = self._conv(blk.convresidual, x, pk, load_mode=mode, out=ws.take((B, blk.cout, Ho, Wo), dev))
        # norm2 -> FiLM -> act -> conv2, + residual                               (adm.py:325-337)
        has_attn = hasattr(blk, "attn")
        os_ = self._stats_buf(ws, B, blk.cout, Ho, Wo, dev) if (want_stats and not has_attn) else None
        if fuse2:
            tab = ws.take((B, ops.table_channels(blk.cout), 4), dev)
            ops.gnorm1_table(ys, blk.norm2.weight, blk.norm2.bias, k2, blk.cout * Ho * Wo, film=film, eps=1e-5, out=tab)
            out = self._conv(blk.conv2, y, pk, res1=r, res1_upsampled=r_up, prenorm=tab, tile_stats=os_,
                             out=ws.take((B, blk.cout, Ho, Wo), dev))
            ws.give(tab)
            ws.give(ys)
            ws.give(y)
        else:
            stats = ws.take((B, 2), dev)
            scratch = ws.take((ops.N.lib().ds_gnorm1_workspace_bytes(B) // 4,), dev)
            ops.gnorm1_stats(y, k2, eps=1e-5, stats=stats, workspace=scratch)
            a2 = ops.gnorm1_apply(y, stats, blk.norm2.weight, blk.norm2.bias, k2, film=film,
                                  out=ws.take((B, blk.cout, Ho, Wo), dev))
            out = self._conv(blk.conv2, a2, pk, res1=r, res1_upsampled=r_up, tile_stats=os_, out=y)
            ws.give(a2)
            ws.give(stats)
            ws.give(scratch)
            if ys is not None:
                ws.give(ys)
        ws.give(r)
        if has_attn:
            os_ = self._stats_buf(ws, B, blk.cout, Ho, Wo, dev) if want_stats else None
            out2 = self._attention(blk.attn, out, pk, ws, tile_stats=os_)
            ws.give(out)
            out = out2
        return out, os_

    def _attention(self, att, x, pk, ws, tile_stats=None):
        """TwoDimensionalAttention.forward (attention.py:67-72,82-90), channel-major."""
        B, E, Hh, Ww = x.shape
        L = Hh * Ww
        m = att.mhattn
        qkv = ops.conv(x, pk[(id(att), "in")], bias=m.in_proj_bias, out=ws.take((B, 3 * E, Hh, Ww), x.device))
        o = ops.attention(qkv.view(B, 3 * E, L), E, out=ws.take((B, E, L), x.device),
                          precision=self.conv_precision)
        y = ops.conv(o.view(B, E, Hh, Ww), pk[(id(att), "out")], bias=m.out_proj.bias,
                     res1=x if self.config.attn_residual else None, tile_stats=tile_stats,
                     out=ws.take(x.shape, x.device))
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

        hs = self._stats_buf(ws, B, cfg.model_channels, H, W, dev)
        h = self._conv(self.input_layer, x, pk, tile_stats=hs, out=ws.take((B, cfg.model_channels, H, W), dev))
        skips = [(h, hs)]                                                       # adm.py:667-675
        for lay in self.encoder.layers:
            for blk in lay.input_blocks:
                h2, hs2 = self._block(blk, h, film(), pk, ws, xs=hs)
                if not any(h is s for s, _ in skips):
                    give(h, hs)
                h, hs = h2, hs2
            skips.append((h, hs))
        for blk in self.middle_block.middle_blocks:
            h2, hs2 = self._block(blk, h, film(), pk, ws, xs=hs)
            if not any(h is s for s, _ in skips):
                give(h, hs)
            h, hs = h2, hs2
        nl = len(self.decoder.layers)

        def join(h, hs, skip, sks):                                              # adm.py:297-304
            if cfg.skip_integration_type == "concat":
                hc = ops.concat2(h, skip, out=ws.take((B, h.shape[1] + skip.shape[1]) + tuple(h.shape[2:]), dev))
                return hc, ((hs, sks) if (hs is not None and sks is not None) else None)   # statistics of a concat are additive
            return ops.add(h, skip, out=ws.take(h.shape, dev)), None

        for li, lay in enumerate(self.decoder.layers):                          # adm.py:764-774, 927-934
            skip, sks = skips.pop()
            nblk = len(lay.input_blocks)
            if cfg.decoder_type == 1:
                hc, hcs = join(h, hs, skip, sks)
                pending = [(h, hs)] + ([(skip, sks)] if skip is not h else [])     # statistics are read by block 0's table
                for j, blk in enumerate(lay.input_blocks):
                    final = li == nl - 1 and j == nblk - 1                        # feeds the output layer: no norm follows
                    h2, hs2 = self._block(blk, hc, film(), pk, ws, xs=hcs, want_stats=not final)
                    if j == 0:
                        ws.give(hc)
                        for t, ts in pending:
                            give(t, ts)
                    else:
                        give(hc, hcs)
                    hc, hcs = h2, hs2
                h, hs = hc, hcs
            else:                                                               # every block joins the skip (adm.py:848-851)
                for j, blk in enumerate(lay.input_blocks):
                    final = li == nl - 1 and j == nblk - 1
                    hc, hcs = join(h, hs, skip, sks)
                    h2, hs2 = self._block(blk, hc, film(), pk, ws, xs=hcs, want_stats=not final)
                    ws.give(hc)
                    if h is not skip:
                        give(h, hs)
                    h, hs = h2, hs2
                give(skip, sks)
        for s_, ss in skips:                                                     # the stem copy is never consumed
            if s_ is not h:
                give(s_, ss)
        m = self.output_layer
        if m.out_channels <= 4:                              # see PUNetG._out_conv
            y = ops.conv_direct(h, m.weight, m.bias, out=out)
        else:
            y = self._conv(m, h, pk, out=out)
        give(h, hs)
        return y
