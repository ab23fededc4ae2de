"""CPU oracle for the ADM score network (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Functional forward pass driven by a ``state_dict`` with the reference's key names, restating
  diffsci/models/nets/adm.py:199-216     ADM.forward
  diffsci/models/nets/adm.py:292-349     ADMBaseBlock.forward / first_block / second_block /
                                         embed_block (FiLM: x1*te1 + te2, no "1 +") / residual_block
  diffsci/models/nets/adm.py:385-406     norms: GroupNorm(num_groups=1), GroupRMSNorm(1, C)
  diffsci/models/nets/adm.py:590-599,667-675   encoder layers / skips (stem output + one per layer)
  diffsci/models/nets/adm.py:744-774,893-900   decoder type 1: cat/add the skip once per layer
  diffsci/models/nets/adm.py:1014-1053   middle block, ADMTimeEmbedding
for the default family: 2-D, default convolutions, GroupLN + GroupRMS, avg-pool down / nearest up
inside the last block of each layer, decoder_type 1 or 2, attention only in the middle block.
"""
import torch
import torch.nn.functional as F

from .punetg_ref import attention_2d, conv3x3, fourier_features


def default_config(**over):
    cfg = dict(input_channels=1, output_channels=1, model_channels=64, time_embed_dim=64,
               output_embed_dim=256, channel_expansion=[2, 4],
               number_resnet_downward_block=2, number_resnet_upward_block=2,
               number_resnet_attn_block=2, number_resnet_before_attn_block=2,
               number_resnet_after_attn_block=2, skip_integration_type="concat", attn_residual=True)
    cfg.update(over)
    return cfg


def group1_rms_norm(x, weight, bias, eps=1e-5):
    """commonlayers.py:372-384 with num_groups = 1: RMS over (C, H, W) of each sample."""
    B, C = x.shape[:2]
    xg = x.view(B, 1, C, *x.shape[2:])
    xg = xg / torch.sqrt(xg.pow(2).mean(dim=tuple(range(2, xg.dim())), keepdim=True) + eps)
    x = xg.view(B, C, *xg.shape[3:])
    shape = (1, C) + (1,) * (x.dim() - 2)
    return x * weight.view(shape) + bias.view(shape)


def time_embedding(sd, t, ye=None):
    """ADMTimeEmbedding.forward, adm.py:1047-1053."""
    te = fourier_features(t, sd["time_embedding.projection.W"])
    te = F.linear(te, sd["time_embedding.mlp.0.weight"], sd["time_embedding.mlp.0.bias"])
    te = F.linear(F.silu(te), sd["time_embedding.mlp.2.weight"], sd["time_embedding.mlp.2.bias"])
    if ye is not None:
        te = te + ye
    return F.silu(te)


def block_norm(kind, sd, p, x):
    """make_norm_layers (adm.py:385-406), num_groups = 1; affine_norm=False leaves no weight / bias keys."""
    C = x.shape[1]
    w, b = sd.get(p + "weight"), sd.get(p + "bias")
    if kind == "GroupLN":
        return F.group_norm(x, 1, w, b, 1e-5)
    if w is None:
        w, b = torch.ones(C).to(x), torch.zeros(C).to(x)
    return group1_rms_norm(x, w, b)


def block(sd, p, x, te, sample=None, has_attn=False, attn_residual=True, circular=False, norms=("GroupLN", "GroupRMS"),
          skip=None, skip_integration_type="concat", has_residual=True):
    """ADMBaseBlock.forward (adm.py:292-349) on fields [B, C, H, W] or volumes [B, C, D, H, W] (AvgPool3d / Conv3d /
    attention over the flattened voxels).  circular: the block's convolutions are CircularConv2d / 3d (conv_fn,
    adm.py:427-443; parameters under `.conv`).  skip: the block's own skip input (decoder blocks), joined first."""
    if skip is not None:                                             # adm.py:297-304
        x = torch.cat([x, skip], dim=1) if skip_integration_type == "concat" else x + skip
    vol = x.dim() == 5

    def resample(v):
        if sample == "down":
            return (F.avg_pool3d if vol else F.avg_pool2d)(v, 2)
        if sample == "up":
            return F.interpolate(v, scale_factor=2.0, mode="nearest")
        return v
    one = (1,) * (x.dim() - 2)
    y = F.silu(block_norm(norms[0], sd, p + "norm1.", x))
    y = conv3x3(sd, p + "conv1", resample(y), circular)
    y = block_norm(norms[1], sd, p + "norm2.", y)
    e = F.linear(te, sd[p + "embed_linear.weight"], sd[p + "embed_linear.bias"])
    te1, te2 = torch.chunk(e, 2, dim=-1)
    y = y * te1.view(*te1.shape, *one) + te2.view(*te2.shape, *one)
    y = conv3x3(sd, p + "conv2", F.silu(y), circular)
    if has_residual:
        rk = p + ("convresidual.conv." if circular else "convresidual.")
        y = y + (F.conv3d if vol else F.conv2d)(resample(x), sd[rk + "weight"], sd[rk + "bias"])
    if has_attn:
        y = attention_2d(sd, p + "attn.", y, attn_residual)
    return y


def adm_forward(sd, cfg, x, t, ye=None):
    """ADM.forward, adm.py:199-216 (ye = conditional_embedding(y) or None)."""
    nl = len(cfg["channel_expansion"])
    circ = cfg.get("convolution_type", "default") == "circular"
    norms = (cfg.get("first_resblock_norm", "GroupLN"), cfg.get("second_resblock_norm", "GroupRMS"))
    te = time_embedding(sd, t, ye)
    x = F.conv2d(x, sd["input_layer.weight"], sd["input_layer.bias"], padding="same")
    skips = [x]
    for i in range(nl):
        nb = cfg["number_resnet_downward_block"]
        for j in range(nb):
            x = block(sd, f"encoder.layers.{i}.input_blocks.{j}.", x, te, sample="down" if j == nb - 1 else None, circular=circ, norms=norms)
        skips.append(x)
    nmid = (cfg["number_resnet_before_attn_block"] + cfg["number_resnet_attn_block"] +
            cfg["number_resnet_after_attn_block"])
    flags = ([False] * cfg["number_resnet_before_attn_block"] +
             [True] * (cfg["number_resnet_attn_block"] - 1) + [False] +
             [False] * cfg["number_resnet_after_attn_block"])
    for j in range(nmid):
        x = block(sd, f"middle_block.middle_blocks.{j}.", x, te, has_attn=flags[j],
                  attn_residual=cfg["attn_residual"], circular=circ, norms=norms)
    def join(a, h):                                                  # adm.py:297-304
        return torch.cat([a, h], dim=1) if cfg["skip_integration_type"] == "concat" else a + h

    dtype2 = cfg.get("decoder_type", 1) == 2
    for i in range(nl):
        h = skips.pop()
        if not dtype2:                                               # ADMDecoderLayer1.forward, adm.py:764-774
            x = join(x, h)
        nb = cfg["number_resnet_upward_block"]
        for j in range(nb):
            if dtype2:                                               # ADMDecoderLayer2: every block (adm.py:848-851)
                x = join(x, h)
            x = block(sd, f"decoder.layers.{i}.input_blocks.{j}.", x, te, sample="up" if j == nb - 1 else None, circular=circ, norms=norms)
    return F.conv2d(x, sd["output_layer.weight"], sd["output_layer.bias"], padding="same")


def make_net(sd, cfg, embed=None):
    def net(x, t, y=None):
        ye = None
        if y is not None:
            ye = y if embed is None else embed(y)
        return adm_forward(sd, cfg, x, t, ye)
    return net
