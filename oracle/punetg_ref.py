"""CPU oracle for the PUNetG score network (TEST INFRASTRUCTURE -- see oracle/__init__.py).

A functional forward pass driven by a PyTorch ``state_dict`` with the reference's
key names, restating
  diffsci/models/nets/punetg.py:356-416          (forward / encode / decode / bottom_forward)
  diffsci/models/nets/commonlayers.py:809-836    (ResnetBlockC.forward)
  diffsci/models/nets/commonlayers.py:372-384    (GroupRMSNorm.forward)
  diffsci/models/nets/commonlayers.py:185-190    (GaussianFourierProjection.forward)
  diffsci/models/nets/commonlayers.py:516-549    (ResnetTimeBlock)
  diffsci/models/nets/commonlayers.py:81,145     (DownSampler / UpSampler forward)
  diffsci/models/nets/attention.py:54-90         (TwoDimensionalAttention, nn.MultiheadAttention, 1 head)
  diffsci/models/nets/normedlayers.py:6-99       (magnitude-preserving conv / linear, convolution_type="mp")
  diffsci/models/nets/attention.py:110-247       (the in-house one-head attention "mp" substitutes)
  diffsci/models/nets/commonlayers.py:387-440    (GroupPixNorm), :882-899 (norm selection)
for the 2-D configuration family ("default" | "circular" | "mp" convolutions, any of GroupLN / GroupRMS /
GroupPix / none in either norm slot, affine or not, bias on / off, dropout 0).  Convolution / GroupNorm /
attention arithmetic is torch's, as in the reference.
"""
import math

import torch
import torch.nn.functional as F


def default_config(**over):
    """PUNetGConfig defaults, punetg_config.py:8-38 (only the fields this oracle reads)."""
    cfg = dict(input_channels=1, output_channels=1, model_channels=64,
               channel_expansion=[2, 4],
               number_resnet_downward_block=2, number_resnet_upward_block=2,
               number_resnet_attn_block=2, number_resnet_before_attn_block=2,
               number_resnet_after_attn_block=2, attn_residual=False)
    cfg.update(over)
    return cfg


def fourier_features(t, W):
    """commonlayers.py:185-190: [sin(2*pi*t*W), cos(2*pi*t*W)]."""
    proj = 2 * math.pi * t[..., None] * W
    return torch.cat([torch.sin(proj), torch.cos(proj)], dim=-1)


def group_rms_norm(x, weight, bias, eps=1e-5):
    """commonlayers.py:372-384 with num_groups == num_channels (per-(b,c) RMS over H*W)."""
    B, C = x.shape[:2]
    xg = x.view(B, C, 1, *x.shape[2:])
    dims = tuple(range(2, xg.dim()))
    xg = xg / (torch.sqrt(xg.pow(2).mean(dim=dims, keepdim=True) + eps))
    x = xg.view(B, C, *xg.shape[3:])
    w = weight.view(1, C, *([1] * (x.dim() - 2)))
    b = bias.view(1, C, *([1] * (x.dim() - 2)))
    return x * w + b


def mp_normalize(x, eps=1e-4):
    """normedlayers.py:95-99."""
    dim = list(range(1, x.ndim))
    n = torch.linalg.vector_norm(x, dim=dim, keepdim=True)
    alpha = math.sqrt(n.numel() / x.numel())
    return x / torch.add(eps, n, alpha=alpha)


def mp_effective(w):
    """Eval-mode weight of MagnitudePreservingConv2d / Linear (normedlayers.py:17-22, 46-55)."""
    return mp_normalize(w) / math.sqrt(w[0].numel())


def conv3x3(sd, name, x, circular=False):
    """torch.nn.Conv2d(padding='same'), or CircularConv2d (commonlayers.py:918-971: F.pad circular in W,
    then in H, then an unpadded convolution; parameters one level down, in `<name>.conv`), or -- circular ==
    "mp" -- MagnitudePreservingConv2d (normedlayers.py:26-55)."""
    conv = F.conv3d if x.dim() == 5 else F.conv2d            # dimension = 3: Conv3d / CircularConv3d (:973-1034)
    if circular == "mp":
        return conv(x, mp_effective(sd[name + ".weight"]), sd.get(name + ".bias"), padding="same")
    if circular:
        p = sd[name + ".conv.weight"].shape[-1] // 2            # self.padding = kernel_size // 2, commonlayers.py:941
        if x.dim() == 5:                                     # W, then H, then D
            x = F.pad(x, (p, p, 0, 0, 0, 0), mode="circular")
            x = F.pad(x, (0, 0, p, p, 0, 0), mode="circular")
            x = F.pad(x, (0, 0, 0, 0, p, p), mode="circular")
        elif p:
            x = F.pad(x, (p, p, 0, 0), mode="circular")
            x = F.pad(x, (0, 0, p, p), mode="circular")
        return conv(x, sd[name + ".conv.weight"], sd.get(name + ".conv.bias"))
    return conv(x, sd[name + ".weight"], sd.get(name + ".bias"), padding="same")      # bias=False: no bias keys


def time_shift(sd, prefix, te, mp=False, ndim=2):
    """ResnetTimeBlock: Linear-SiLU-Linear-SiLU-Linear, commonlayers.py:512-522,546-549 (magnitude-preserving
    linears when mp)."""
    def w(i):
        wt = sd[prefix + f"net.{i}.weight"]
        return mp_effective(wt) if mp else wt
    spatial = te.dim() - 2 == ndim                            # commonlayers.py:537-546: a field of embeddings, one MLP per pixel
    if spatial:
        shape = te.shape
        te = te.movedim(1, -1).reshape(-1, shape[1])          # 'nbatch embed s.. -> (nbatch s..) embed'
    h = F.linear(te, w(0), sd[prefix + "net.0.bias"])
    h = F.silu(h)
    h = F.linear(h, w(2), sd[prefix + "net.2.bias"])
    h = F.silu(h)
    h = F.linear(h, w(4), sd[prefix + "net.4.bias"])
    if spatial:
        return h.reshape(shape[0], *shape[2:], h.shape[1]).movedim(-1, 1)
    return h.view(*h.shape, *([1] * ndim))


def rescale_yt(yt, y):
    """ResnetBlockC.rescale_yt, commonlayers.py:838-869: a field-valued time shift follows the block's resolution by
    taking the top-left corner of every window (CornerPool, :1035-1098).  The upscaling branch of the reference builds
    torch.nn.Upsample(shape_factor), whose first argument is the output SIZE, so it only works when the block's
    resolution equals the factor; it is restated as written."""
    yt_dims, y_dims = tuple(yt.shape[2:]), tuple(y.shape[2:])
    if yt_dims == (1,) * len(y_dims) or yt_dims == y_dims:
        return yt
    factor = yt_dims[0] / y_dims[0]
    if factor > 1:
        f = int(factor)
        if any(dy * f != dyt for dy, dyt in zip(y_dims, yt_dims)):
            raise ValueError(f"yt_dims {yt_dims} and y_dims {y_dims} are not compatible")
        return yt[(Ellipsis,) + (slice(None, None, f),) * len(y_dims)]
    f = int(1 / factor)
    if any(dyt * f != dy for dy, dyt in zip(y_dims, yt_dims)):
        raise ValueError(f"yt_dims {yt_dims} and y_dims {y_dims} are not compatible")
    return F.interpolate(yt, size=f, mode="nearest")


def block_norm(kind, sd, prefix, x):
    """One norm slot of ResnetBlockC with num_groups = C (commonlayers.py:766-774, 882-899); affine_norm=False
    leaves no weight / bias keys."""
    C = x.shape[1]
    w, b = sd.get(prefix + "weight"), sd.get(prefix + "bias")
    if kind == "GroupLN":
        return F.group_norm(x, C, w, b, 1e-5)
    if kind == "GroupRMS":
        if w is None:
            w, b = torch.ones(C).to(x), torch.zeros(C).to(x)
        return group_rms_norm(x, w, b)
    if kind == "GroupPix":                                   # commonlayers.py:425-440, one channel per group
        y = x / torch.sqrt(x.pow(2) + 1e-5)
        if w is not None:
            shape = (1, C) + (1,) * (x.dim() - 2)
            y = y * w.view(shape) + b.view(shape)
        return y
    return x                                                 # torch.nn.Identity


def resnet_block(sd, prefix, x, te, circular=False, norms=("GroupLN", "GroupRMS"), extra_residual=None):
    """ResnetBlockC.forward, commonlayers.py:824-833; extra_residual: the user module of punetg.py:83-92 (a callable
    here), added after the residual connection."""
    h = block_norm(norms[0], sd, prefix + "gnorm1.", x)
    y = conv3x3(sd, prefix + "conv1", F.silu(h), circular)
    y = y + rescale_yt(time_shift(sd, prefix + "timeblock.", te, mp=circular == "mp", ndim=x.dim() - 2), y)   # [B, C, 1, 1(, 1)] or a field
    h = block_norm(norms[1], sd, prefix + "gnorm2.", y)
    y = conv3x3(sd, prefix + "conv2", F.silu(h), circular)
    y = y + x
    if extra_residual is not None:
        y = y + extra_residual(x)
    return y


def fourier_input(sd, x):
    """ConvolutionalFourierProjection.forward with bias=False (commonlayers.py:246-255): PUNetG's convin when
    in_embedding=True (punetg.py:194-202)."""
    xc = torch.einsum('bc...,cd->bd...', x, 2 * math.pi * sd["convin.W"])
    return torch.cat([torch.sin(xc), torch.cos(xc)], dim=1)


def mp_attention_2d(sd, prefix, x, attn_residual=False, magnitude_preserving=True, cosine=False):
    """TwoDimensionalAttention around the in-house MultiHeadAttention(1 head, dk = dv = C), attention.py:29-52,
    156-247: 'dot' (250-296) or 'cosine' (300-372) logits; weights renormalised only when magnitude preserving,
    always divided by sqrt(fan_in)."""
    if x.dim() == 5:
        Bv, Cv, Dv, Hv, Wv = x.shape
        return mp_attention_2d(sd, prefix, x.reshape(Bv, Cv, Dv * Hv, Wv), attn_residual, magnitude_preserving,
                               cosine).reshape(x.shape)
    B, C, Hh, Ww = x.shape
    xr = x.permute(0, 2, 3, 1).reshape(B, Hh * Ww, C)
    ws = []
    for kind in ("q", "k", "v", "o"):
        weight = sd[prefix + f"mhattn.{kind}_proj_matrix"]
        fan_in = weight.shape[0] * weight.shape[2] if kind == "o" else weight.shape[1]
        if magnitude_preserving:
            norm = torch.linalg.vector_norm(weight, dim=[0, 2] if kind == "o" else 1, keepdim=True)
            alpha = math.sqrt(norm.numel() / weight.numel())
            weight = weight / (alpha * norm + 1e-4)
        ws.append(weight / math.sqrt(fan_in))
    wq, wk, wv, wo = ws
    q = torch.einsum('...ij, kjm -> ...kim', xr, wq)
    k = torch.einsum('...ij, kjm -> ...kim', xr, wk)
    v = torch.einsum('...ij, kjm -> ...kim', xr, wv)
    if cosine:                                                # cosine_similarity, attention.py:362-372
        qn = q / (torch.linalg.vector_norm(q, dim=-1, keepdim=True) + 1e-8)
        kn = k / (torch.linalg.vector_norm(k, dim=-1, keepdim=True) + 1e-8)
        inner = torch.einsum('...nd,...md->...nm', qn, kn)
    else:
        inner = torch.einsum('...ij, ...kj -> ...ik', q, k)
        inner = inner / math.sqrt(q.shape[-1])
    a = torch.einsum('...ij, ...jk -> ...ik', torch.softmax(inner, dim=-1), v)
    out = torch.einsum('...ijk, ilk -> ...jl', a, wo)
    out = out.reshape(B, Hh, Ww, C).permute(0, 3, 1, 2)
    return x + out if attn_residual else out


def attention_2d(sd, prefix, x, attn_residual=False):
    """TwoDimensionalAttention (attention.py:67-90) around nn.MultiheadAttention(E, 1 head)."""
    if x.dim() == 5:                                     # ThreeDimensionalAttention (attention.py:93-102): flatten the voxels
        Bv, Cv, Dv, Hv, Wv = x.shape
        return attention_2d(sd, prefix, x.reshape(Bv, Cv, Dv * Hv, Wv), attn_residual).reshape(x.shape)
    B, C, Hh, Ww = x.shape
    xr = x.permute(0, 2, 3, 1).reshape(B, Hh * Ww, C)   # 'b c w h -> b (w h) c'
    out, _ = F.multi_head_attention_forward(
        xr.transpose(0, 1), xr.transpose(0, 1), xr.transpose(0, 1),
        C, 1,
        sd[prefix + "mhattn.in_proj_weight"], sd[prefix + "mhattn.in_proj_bias"],
        None, None, False, 0.0,
        sd[prefix + "mhattn.out_proj.weight"], sd[prefix + "mhattn.out_proj.bias"],
        training=False, need_weights=False)
    out = out.transpose(0, 1).reshape(B, Hh, Ww, C).permute(0, 3, 1, 2)
    return x + out if attn_residual else out


def punetg_forward(sd, cfg, x, t, ye=None):
    """PUNetG.forward, punetg.py:389-416.  ``ye`` is the already-embedded condition
    (conditional_embedding(y), shape [B or 1, model_channels]) or None."""
    nlev = len(cfg["channel_expansion"])
    ctype = cfg.get("convolution_type", "default")
    circ = "mp" if ctype == "mp" else ctype == "circular"
    norms = (cfg.get("first_resblock_norm", "GroupLN"), cfg.get("second_resblock_norm", "GroupRMS"))
    if not cfg.get("bias", True):                                    # punetg.py:390-394: constant-one input channel
        xe_shape = list(x.shape)
        xe_shape[1] = 1
        x = torch.cat([x, torch.ones(xe_shape).to(x)], dim=1)
    er = cfg.get("extra_residual")                                   # a callable (test configurations only)
    x = fourier_input(sd, x) if cfg.get("in_embedding", False) else conv3x3(sd, "convin", x, circ)
    if t is None:                                                    # punetg.py:396-399: no time input
        te = torch.zeros(x.shape[0], cfg["model_channels"]).to(x)
    else:
        te = fourier_features(t, sd["time_projection.W"])
    if ye is not None:
        if ye.dim() > te.dim():                                      # punetg.py:405-407: a field of embeddings
            te = te.reshape(list(te.shape) + [1] * (ye.dim() - te.dim()))
        te = te + ye
    skips = []
    for lv in range(nlev):                                           # encode, punetg.py:356-365
        for r in range(cfg["number_resnet_downward_block"]):
            x = resnet_block(sd, f"downward_blocks.{lv}.{r}.", x, te, circ, norms, er)
        skips.append(x)
        x = conv3x3(sd, f"downsamplers.{lv}.conv", (F.max_pool3d if x.dim() == 5 else F.max_pool2d)(x, 2), circ)
    for r in range(cfg["number_resnet_before_attn_block"]):         # bottom, punetg.py:378-387
        x = resnet_block(sd, f"before_block.{r}.", x, te, circ, norms, er)
    xa = x
    nattn = cfg["number_resnet_attn_block"]
    for r in range(nattn):                                           # punetg.py:344-354
        xa = resnet_block(sd, f"attn_resnet_block.{r}.", xa, te, circ, norms, er)
        if r < nattn - 1:
            cosine = cfg.get("attn_type", "default") == "cosine"
            if ctype == "mp" or cosine:
                xa = mp_attention_2d(sd, f"attn_block.{r}.", xa, cfg["attn_residual"], ctype == "mp", cosine)
            else:
                xa = attention_2d(sd, f"attn_block.{r}.", xa, cfg["attn_residual"])
    x = x + xa
    for r in range(cfg["number_resnet_after_attn_block"]):
        x = resnet_block(sd, f"after_block.{r}.", x, te, circ, norms, er)
    for lv in range(nlev):                                           # decode, punetg.py:367-376
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        x = conv3x3(sd, f"upsamplers.{lv}.conv", x, circ)
        x = x + skips.pop()
        for r in range(cfg["number_resnet_upward_block"]):
            x = resnet_block(sd, f"upward_blocks.{lv}.{r}.", x, te, circ, norms, er)
    return conv3x3(sd, "convout", x, circ)


def make_net(sd, cfg, embed=None):
    """Return net(x, c_noise[, y]) with the reference model protocol (karrasmodule.py:706-716).
    embed: optional callable y -> [*, model_channels] (conditional_embedding)."""
    def net(x, t, y=None):
        ye = None
        if y is not None:
            ye = y if embed is None else embed(y)
        return punetg_forward(sd, cfg, x, t, ye)
    return net


def random_state_dict(cfg, seed=0, dtype=torch.float32):
    """Synthetic weights with the reference's default initialisers:
    Conv/Linear U(+-1/sqrt(fan_in)) for weight and bias, norm w=1 b=0, Fourier W~N(0,30^2),
    MultiheadAttention xavier-uniform in_proj and zero biases.  Used by bench/smoke only;
    parity fixtures carry the reference's own state_dict."""
    g = torch.Generator().manual_seed(seed)
    mc = cfg["model_channels"]
    mult = [1] + list(cfg["channel_expansion"])
    sd = {}

    def uni(shape, bound):
        return (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1).mul(bound).to(dtype)

    def conv(name, co, ci, k=3):
        b = 1.0 / math.sqrt(ci * k * k)
        sd[name + ".weight"] = uni((co, ci, k, k), b)
        sd[name + ".bias"] = uni((co,), b)

    def lin(name, co, ci):
        b = 1.0 / math.sqrt(ci)
        sd[name + ".weight"] = uni((co, ci), b)
        sd[name + ".bias"] = uni((co,), b)

    def res(prefix, C):
        for n in ("gnorm1", "gnorm2"):
            sd[prefix + n + ".weight"] = torch.ones(C, dtype=dtype)
            sd[prefix + n + ".bias"] = torch.zeros(C, dtype=dtype)
        conv(prefix + "conv1", C, C)
        conv(prefix + "conv2", C, C)
        lin(prefix + "timeblock.net.0", 4 * mc, mc)
        lin(prefix + "timeblock.net.2", 4 * mc, 4 * mc)
        lin(prefix + "timeblock.net.4", C, 4 * mc)

    sd["time_projection.W"] = (torch.randn(mc // 2, generator=g, dtype=torch.float64) * 30.0).to(dtype)
    conv("convin", mc, cfg["input_channels"])
    conv("convout", cfg["output_channels"], mc)
    nlev = len(cfg["channel_expansion"])
    for lv in range(nlev):
        for r in range(cfg["number_resnet_downward_block"]):
            res(f"downward_blocks.{lv}.{r}.", mult[lv] * mc)
        conv(f"downsamplers.{lv}.conv", mult[lv + 1] * mc, mult[lv] * mc)
    Cb = mult[-1] * mc
    for r in range(cfg["number_resnet_before_attn_block"]):
        res(f"before_block.{r}.", Cb)
    for r in range(cfg["number_resnet_attn_block"]):
        res(f"attn_resnet_block.{r}.", Cb)
    for r in range(cfg["number_resnet_attn_block"] - 1):
        bound = math.sqrt(6.0 / (3 * Cb + Cb))
        sd[f"attn_block.{r}.mhattn.in_proj_weight"] = uni((3 * Cb, Cb), bound)
        sd[f"attn_block.{r}.mhattn.in_proj_bias"] = torch.zeros(3 * Cb, dtype=dtype)
        sd[f"attn_block.{r}.mhattn.out_proj.weight"] = uni((Cb, Cb), 1.0 / math.sqrt(Cb))
        sd[f"attn_block.{r}.mhattn.out_proj.bias"] = torch.zeros(Cb, dtype=dtype)
    for r in range(cfg["number_resnet_after_attn_block"]):
        res(f"after_block.{r}.", Cb)
    rmult = list(reversed(mult))
    for lv in range(nlev):
        conv(f"upsamplers.{lv}.conv", rmult[lv + 1] * mc, rmult[lv] * mc)
        for r in range(cfg["number_resnet_upward_block"]):
            res(f"upward_blocks.{lv}.{r}.", rmult[lv + 1] * mc)
    return sd
