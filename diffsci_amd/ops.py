"""Tensor-level wrappers over the C ABI: shape / dtype / device checks on the host, then a raw
pointer + stream call.  PyTorch is used only as the owner of device memory and of the stream."""
import ctypes

import torch

from . import _native as N
from ._native import EvalCoef  # noqa: F401  (re-export)


def require_device(t, what="tensor"):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{what} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(
            f"{what} lives on {t.device}: diffsci_amd computes only on an AMD GPU through "
            "libdiffsci_hip.so (there is no CPU path; move the module and inputs to 'cuda').")
    if t.dtype != torch.float32:
        raise TypeError(f"{what} has dtype {t.dtype}; the HIP path is fp32 only")
    if t.device.index != torch.cuda.current_device():
        # the C ABI launches on the stream it is handed and never switches devices: a launch on cuda:0's stream
        # with cuda:1 pointers would fault (or silently compute on the wrong GPU's copy of a kernel attribute)
        raise RuntimeError(f"{what} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}: "
                           "wrap the call in `with torch.cuda.device(tensor.device):` (KarrasModule / SIModule / the "
                           "networks do this for their own entry points)")


def on_device_of(t):
    """Context manager: make t's GPU the current device (and its current stream the launch stream)."""
    return torch.cuda.device(t.device)


def device_guard(fn):
    """Decorator for the public entry points of the modules: run with the GPU of the first CUDA tensor argument as
    the current device, so a module on cuda:1 works while the caller's current device is cuda:0 (the C ABI launches
    on the stream it is handed and never switches devices; every op checks its tensors against the current device)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(*args, **kwargs)
                break
        return fn(*args, **kwargs)
    return wrapped


def _p(t, what="tensor"):
    if t is None:
        return None
    require_device(t, what)
    if not t.is_contiguous():
        raise ValueError(f"{what} must be contiguous")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ---------------------------------------------------------------- per-sample activation exponents of the fp16x3 kernels
# fp16 has 5 exponent bits; the reference's fp32 convolutions take raw user fields (punetg.py:719-735) and c_in = 1
# parameterisations (preconditioners.py:139-161) at any magnitude.  A launch whose input is not normalised by construction
# scales sample b by a power of two taken from max |x_b| (include/diffsci_hip.h: in_amax / out_amax).  "amax" tensors are
# int32 [rows] holding float bits; they are MERGED into (atomicMax), so they start from zero.
class _Normalised:
    """in_amax=NORMALISED: the input is normalised by construction (a norm + SiLU output): no scaling, no reduction."""

    def __repr__(self):
        return "ops.NORMALISED"


NORMALISED = _Normalised()


def _pi(t, n, what="amax"):
    if t is None:
        return None
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.int32 and t.is_contiguous() and t.numel() == n):
        raise TypeError(f"{what} must be a contiguous int32 device tensor of {n} entries (float bits of per-sample max |x|)")
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"{what} lives on another device than the current one")
    return t.data_ptr()


def amax_zero(t):
    """Zero amax slots (a one-line kernel: hipGraph memset nodes replay unreliably on this ROCm, see ds_amax.hip)."""
    N.check(N.lib().ds_fill_u32(_pi(t, t.numel()), 0, t.numel(), _stream()), "ds_fill_u32")
    return t


def amax_new(rows, device):
    return torch.zeros(int(rows), dtype=torch.int32, device=device)


def absmax_rows(x, rows=None, out=None):
    """out[r] = max(out[r], float bits of max |x[r]|) over the rows of x viewed as [rows, -1] (default: the batch dimension).
    x: contiguous, or a channel slice x_full[:, c0:c1] of a contiguous tensor (rows = samples, dense inside a row).
    out=None: a fresh zeroed tensor (an allocation: captured code passes its own, zeroed, slots)."""
    require_device(x, "x")
    if x.is_contiguous():
        rows = x.shape[0] if rows is None else int(rows)
        n = x.numel() // max(rows, 1)
        stride = n
    else:
        if rows not in (None, x.shape[0]) or x.dim() < 2 or not x[0].is_contiguous():
            raise ValueError("absmax_rows: x must be contiguous or a channel slice of a contiguous tensor")
        rows, n, stride = x.shape[0], x[0].numel(), x.stride(0)
    if out is None:
        out = amax_new(rows, x.device)
    N.check(N.lib().ds_absmax_rows(_pi(out, rows), x.data_ptr(), rows, n, stride, _stream()), "ds_absmax_rows")
    return out


CHANNEL_GAP = 14     # binades: see ds_absmax_channels


def absmax_channels(x, out, scratch, flag=None, wmax=None, gap=CHANNEL_GAP):
    """absmax_rows for an input layer's x [B, C, *spatial]: out [B] (zeroed slots) <- per-sample maxima; scratch: zeroed int32
    [B*C]; flag (int32 [1] or None) is OR-ed with 1 when one exponent per sample cannot serve the layer given its weights (wmax
    [C]: largest |weight| per input channel) -- the caller's signal to run that layer on the exact-fp32 kernel (nets/precision.py)."""
    require_device(x, "x")
    if not x.is_contiguous():
        raise ValueError("absmax_channels: x must be contiguous")
    B, C = x.shape[0], x.shape[1]
    if wmax is not None and wmax.numel() != C:
        raise ValueError("absmax_channels: wmax must hold one entry per input channel")
    N.check(N.lib().ds_absmax_channels(_pi(out, B), _pi(flag, 1, "flag"), _pi(scratch, B * C, "scratch"), x.data_ptr(), _p(wmax, "wmax"),
                                       B, C, x.numel() // max(B * C, 1), int(gap), _stream()), "ds_absmax_channels")
    return out


INPUT_AMAX_MAX_FLOATS = 1 << 20      # per sample: above this one workgroup per sample would be the slow way


def input_amax(arena, out_row, x, flag=None, wmax=None, gap=CHANNEL_GAP):
    """One launch at the head of a network evaluation: zero the amax arena (int32 [rows, B]) and fill its row `out_row` with the
    per-sample maxima of the input x [B, C, *spatial] (C <= 64, C * spatial <= INPUT_AMAX_MAX_FLOATS), with absmax_channels'
    channel criterion.  Returns the row."""
    require_device(x, "x")
    rows, B = arena.shape
    C = x.shape[1]
    if x.shape[0] != B or not x.is_contiguous() or not arena.is_contiguous() or arena.dtype != torch.int32:
        raise ValueError("input_amax: arena int32 [rows, B] and a contiguous x [B, C, ...]")
    N.check(N.lib().ds_input_amax(arena.data_ptr(), rows, int(out_row), _pi(flag, 1, "flag"), x.data_ptr(), _p(wmax, "wmax"), B, C,
                                  x.numel() // max(B * C, 1), int(gap), _stream()), "ds_input_amax")
    return arena[out_row]


def amax_merge(out, a, b=None):
    """out[i] = max(out[i], a[i], b[i]): the amax of a channel concatenation from those of its parts."""
    n = out.numel()
    N.check(N.lib().ds_amax_merge(_pi(out, n), _pi(a, n), _pi(b, n), n, _stream()), "ds_amax_merge")
    return out


def _in_amax(x, in_amax, rows, raw):
    """Pointer for a kernel's in_amax argument.  raw: the launch reads x without a normalising loader (with one, the table's
    fourth column carries the exponent)."""
    if in_amax is NORMALISED or not raw:
        return None
    if in_amax is None:
        in_amax = absmax_rows(x, rows)
    return _pi(in_amax, rows, "in_amax")


def _xin_numel(x, xin_out, copies):
    """xin_out of the step kernels: numel of x, or twice that when the kernel writes two copies (batched guidance)."""
    if xin_out is not None and xin_out.numel() != x.numel() * (2 if copies == 2 else 1):
        raise ValueError(f"xin_out holds {xin_out.numel()} elements; expected {x.numel() * (2 if copies == 2 else 1)}")


def _same_numel(*ts):
    n = None
    for t in ts:
        if t is None:
            continue
        if n is None:
            n = t.numel()
        elif t.numel() != n:
            raise ValueError(f"size mismatch: {t.numel()} vs {n} elements")
    return n


def scale(x, s, out=None):
    out = torch.empty_like(x) if out is None else out
    n = _same_numel(x, out)
    N.check(N.lib().ds_karras_scale(_p(out, "out"), _p(x, "x"), float(s), n, _stream()), "ds_karras_scale")
    return out


def add(a, b, out=None):
    out = torch.empty_like(a) if out is None else out
    n = _same_numel(a, b, out)
    N.check(N.lib().ds_add(_p(out), _p(a), _p(b), n, _stream()), "ds_add")
    return out


def mask_blend(x, y, mask, out=None):
    """x*(1-mask) + y*mask with mask [*shape] broadcast over the batch."""
    B = x.shape[0]
    nps = x.numel() // max(B, 1)
    if y.shape != x.shape or mask.numel() != nps:
        raise ValueError(f"mask_blend: x {tuple(x.shape)}, y {tuple(y.shape)}, mask {tuple(mask.shape)}")
    out = torch.empty_like(x) if out is None else out
    N.check(N.lib().ds_mask_blend(_p(out), _p(x), _p(y), _p(mask), nps, B, _stream()), "ds_mask_blend")
    return out


def axpby(x, a, y=None, b=0.0, out=None):
    """a*x + b*y (y optional)."""
    out = torch.empty_like(x) if out is None else out
    _same_numel(x, y, out)
    N.check(N.lib().ds_axpby(_p(out), _p(x), float(a), _p(y), float(b), x.numel(), _stream()), "ds_axpby")
    return out


def div_scalar(x, s, out=None):
    """x / s."""
    out = torch.empty_like(x) if out is None else out
    N.check(N.lib().ds_div_scalar(_p(out), _p(x), float(s), x.numel(), _stream()), "ds_div_scalar")
    return out


def batchnorm_eval(x, mean, var, weight=None, bias=None, eps=1e-5, sigma=1.0, inverse=False, out=None):
    """DimensionAgnosticBatchNorm.forward / .unnorm with running statistics; x [B, C, *spatial]."""
    require_device(x, "x")
    out = torch.empty_like(x) if out is None else out
    _same_numel(x, out)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // max(B * C, 1)
    nc = mean.numel()
    if var.numel() != nc or nc not in (1, C) or (weight is not None and (weight.numel() != nc or bias.numel() != nc)):
        raise ValueError("batch-norm statistics / affine must hold 1 or C entries")
    N.check(N.lib().ds_batchnorm_eval(_p(out), _p(x), _p(mean), _p(var), _p(weight), _p(bias), float(eps), float(sigma),
                                      1 if inverse else 0, B, C, nc, HW, _stream()), "ds_batchnorm_eval")
    return out


def lerp_stack(x1, x2, n):
    """stack([x1 + (x2 - x1)*i/(n-1) for i in range(n)])."""
    out = torch.empty((n,) + tuple(x1.shape), dtype=torch.float32, device=x1.device)
    N.check(N.lib().ds_lerp_stack(_p(out), _p(x1), _p(x2), int(n), x1.numel(), _stream()), "ds_lerp_stack")
    return out


def drift(x, f, k, fu=None, out=None):
    out = torch.empty_like(f) if out is None else out
    n = _same_numel(x, f, fu, out)
    N.check(N.lib().ds_karras_drift(_p(out), _p(x), _p(f), _p(fu), ctypes.byref(k), n, _stream()),
            "ds_karras_drift")
    return out


def score(x, f, k, fu=None, out=None):
    out = torch.empty_like(f) if out is None else out
    n = _same_numel(x, f, fu, out)
    N.check(N.lib().ds_karras_score(_p(out), _p(x), _p(f), _p(fu), ctypes.byref(k), n, _stream()),
            "ds_karras_score")
    return out


def _philox(philox):
    """(state tensor int64[2] on the device, offset) -> (pointer, offset) or (None, 0)."""
    if philox is None:
        return None, 0
    state, offset = philox
    if not (isinstance(state, torch.Tensor) and state.is_cuda and state.dtype == torch.int64 and state.numel() == 2
            and state.is_contiguous()):
        raise TypeError("philox state must be a contiguous int64[2] device tensor (seed, base offset)")
    if state.device.index != torch.cuda.current_device():
        raise RuntimeError("philox state lives on another device than the current one")
    return state.data_ptr(), int(offset)


def philox_counters(n):
    """Philox counters one noise tensor of n elements consumes (4 normals per counter)."""
    return (int(n) + 3) // 4


def philox_normal(state, offset, shape, out=None):
    """The standard-normal stream the stepper kernels generate in place for (state, offset): element e <- counter
    state[1] + offset + e/4, output e%4."""
    if out is None:
        out = torch.empty(tuple(shape), dtype=torch.float32, device=state.device)
    ps, po = _philox((state, offset))
    N.check(N.lib().ds_philox_normal(_p(out, "out"), ps, po, out.numel(), _stream()), "ds_philox_normal")
    return out


def euler(x, f, k, dt, fu=None, x_out=None, xin_out=None, c_in_next=1.0, eps=None, noise_coef=0.0,
          sqrt_abs_dt=0.0, philox=None):
    n = _same_numel(x, f, fu, x_out, eps)
    _xin_numel(x, xin_out, k.xin_copies)
    ps, po = _philox(philox)
    N.check(N.lib().ds_karras_euler(_p(x_out), _p(xin_out), _p(x), _p(f), _p(fu), ctypes.byref(k),
                                    float(dt), float(c_in_next), _p(eps), ps, po, float(noise_coef),
                                    float(sqrt_abs_dt), n, _stream()), "ds_karras_euler")
    return x_out


def heun(x, f1, k1, f2, k2, dt, f1u=None, f2u=None, x_out=None, xin_out=None, c_in_next=1.0):
    n = _same_numel(x, f1, f2, f1u, f2u, x_out)
    _xin_numel(x, xin_out, k2.xin_copies)
    N.check(N.lib().ds_karras_heun(_p(x_out), _p(xin_out), _p(x), _p(f1), _p(f1u), ctypes.byref(k1),
                                   _p(f2), _p(f2u), ctypes.byref(k2), float(dt), float(c_in_next), n,
                                   _stream()), "ds_karras_heun")
    return x_out


def churn(x, eps, coef, xhat_out, xin_out=None, c_in=1.0, philox=None, ratio=1.0, scale=1.0, xin_copies=1):
    """x_hat = ratio*x + coef*eps, with eps injected (a tensor) or, eps=None, generated in the kernel from
    philox = (state, offset); xin_out = c_in * (x_hat / scale), written xin_copies (1 or 2) times back to back."""
    n = _same_numel(x, eps, xhat_out)
    _xin_numel(x, xin_out, xin_copies)
    ps, po = _philox(philox)
    N.check(N.lib().ds_karras_churn(_p(xhat_out), _p(xin_out), _p(x), _p(eps), ps, po, float(coef), float(c_in), float(ratio),
                                    float(scale), int(xin_copies), n, _stream()), "ds_karras_churn")
    return xhat_out


def denoiser(x, f, c_out, c_skip, fu=None, guidance=1.0, out=None):
    out = torch.empty_like(x) if out is None else out
    B = x.shape[0]
    if c_out.numel() != B or c_skip.numel() != B:
        raise ValueError("c_out / c_skip must have one entry per sample")
    _same_numel(x, f, fu, out)
    N.check(N.lib().ds_karras_denoiser(_p(out), _p(x), _p(f), _p(fu), float(guidance), float(1 - guidance),
                                       _p(c_out), _p(c_skip), B, x.numel() // max(B, 1), _stream()),
            "ds_karras_denoiser")
    return out


def inorm_silu(x, w, b, kind, eps=1e-5, out=None):
    """kind 0: GroupNorm(C, C)+SiLU, kind 1: GroupRMSNorm(C, C)+SiLU; x [B, C, *spatial]."""
    out = torch.empty_like(x) if out is None else out
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // max(B * C, 1)
    if w is not None and (w.numel() != C or b.numel() != C):
        raise ValueError("norm affine parameters must have C entries")
    _same_numel(x, out)
    N.check(N.lib().ds_inorm_silu(_p(out), _p(x), _p(w), _p(b), B, C, HW, float(eps), int(kind), _stream()),
            "ds_inorm_silu")
    return out


def conv3d(x, w, bias=None, shift=None, res1=None, res2=None, load_mode=N.DS_LOAD_PLAIN, circular=False, out=None):
    """3x3x3 'same' convolution of a volume [B, Cin, Di, Hi, Wi] with raw torch weights [Cout, Cin, 3, 3, 3], exact
    fp32; MaxPool3d(2) / nearest x2 upsampling fused in the loader; shift [1 or B, Cout]."""
    require_device(x, "x")
    B, Cin, Di, Hi, Wi = x.shape
    Cout = w.shape[0]
    if tuple(w.shape) != (Cout, Cin, 3, 3, 3):
        raise ValueError(f"conv3d: weight must be [Cout, {Cin}, 3, 3, 3]; got {tuple(w.shape)}")
    if load_mode == N.DS_LOAD_MAXPOOL2:
        if Di % 2 or Hi % 2 or Wi % 2:
            raise ValueError("pooling load needs an even input volume")
        D, H, W = Di // 2, Hi // 2, Wi // 2
    elif load_mode == N.DS_LOAD_UPSAMPLE2:
        D, H, W = 2 * Di, 2 * Hi, 2 * Wi
    else:
        D, H, W = Di, Hi, Wi
    if out is None:
        out = torch.empty((B, Cout, D, H, W), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (B, Cout, D, H, W):
        raise ValueError(f"out has shape {tuple(out.shape)}, expected {(B, Cout, D, H, W)}")
    stride = 0
    if shift is not None:
        if shift.dim() != 2 or shift.shape[1] != Cout or shift.shape[0] not in (1, B):
            raise ValueError(f"shift must be [1 or B, Cout]; got {tuple(shift.shape)}")
        stride = 0 if shift.shape[0] == 1 else Cout
    for r in (res1, res2):
        if r is not None and tuple(r.shape) != (B, Cout, D, H, W):
            raise ValueError("residual shape mismatch")
    if bias is not None and bias.numel() != Cout:
        raise ValueError("bias must have Cout entries")
    N.check(N.lib().ds_conv3d_direct(_p(out), _p(x.contiguous()), _p(w.contiguous()), _p(bias), _p(shift), stride, _p(res1),
                                     _p(res2), B, Cin, Cout, D, H, W,
                                     load_mode | (N.DS_PAD_CIRCULAR if circular else 0), _stream()), "ds_conv3d_direct")
    return out


def pack_conv3d(w, upsampled=False):
    """A k x k x k weight [Cout, Cin, k, k, k] (k = 1, 3, 5, 7) as k fp16x3-packed k x k weights, one per depth tap (see
    conv3d_mfma); the parity kernels of a nearest-x2 upsampled input (upsampled) exist for k = 3."""
    k = w.shape[2] if w.dim() == 5 else 0
    if w.dim() != 5 or tuple(w.shape[2:]) != (k, k, k) or k not in (1, 3, 5, 7):
        raise ValueError("pack_conv3d: weight must be [Cout, Cin, k, k, k] with k in (1, 3, 5, 7)")
    return [pack_conv(w[:, :, kz].contiguous(), "fp16x3", upsampled=upsampled and k == 3) for kz in range(k)]


def volume_stat_tiles(D, HW):
    """Entries per (sample, channel) of the statistics ds_slices_to_volume_stats leaves for a [.., D, H, W] volume."""
    return N.lib().ds_volume_stat_tiles(int(D), int(HW))


def _slice_rows(shift, B, D, Cout, ws=None, pad=1):
    """Per-slice rows of a per-sample time shift for the 2-D batch of all slices but the outermost `pad` on each end -> (rows,
    buffer to give back to ws or None).  With a pool the expansion lands in a pool buffer (a captured loop must not allocate)."""
    if shift is None:
        return None, None
    if shift.dim() != 2 or shift.shape[1] != Cout or shift.shape[0] not in (1, B):
        raise ValueError(f"shift must be [1 or B, Cout]; got {tuple(shift.shape)}")
    if shift.shape[0] == 1:
        return shift, None
    DP = D + 2 * pad
    ns = B * DP
    if ws is None:
        return shift.repeat_interleave(DP, dim=0)[pad:ns - pad].contiguous(), None
    buf = ws.take((B, DP, Cout), shift.device)
    buf.copy_(shift[:, None, :].expand(B, DP, Cout))
    return buf.view(ns, Cout)[pad:ns - pad], buf


def _depth_taps(s_in, s_out, packs, bias, rows, load_mode, circular, prenorm=None, tile_stats=None, in_amax=None):
    """The depth-tap launches of a k x k x k convolution over slice-major volumes (k = len(packs); three for 3x3x3): the centre
    tap initialises the accumulator (all slices of s_out but the outermost k/2 on each end), the others add to it; prenorm:
    per-SLICE table [B*(D+2), ceil16(Cin), 4] (ds_slice_tables) for the fused norm + SiLU loader; tile_stats: filled by the last
    launch.  in_amax: per-SLICE max |s_in| [B*(D+2 pad)] (every slice is a 2-D sample with its own exponent), NORMALISED, or
    None = computed here."""
    ns = s_in.shape[0]
    P = len(packs) // 2
    acc = s_out[P:ns - P]
    if prenorm is not None:
        in_amax = NORMALISED
    elif in_amax is None:
        in_amax = absmax_rows(s_in)
    order = [0] + [d for q in range(1, P + 1) for d in (-q, q)]
    for n, dz in enumerate(order):
        conv(s_in[P + dz:ns - P + dz], packs[dz + P], bias=bias if n == 0 else None, shift=rows if n == 0 else None,
             res1=None if n == 0 else acc, load_mode=load_mode, circular=circular, out=acc,
             prenorm=None if prenorm is None else prenorm[P + dz:ns - P + dz], tile_stats=tile_stats if n == len(order) - 1 else None,
             in_amax=in_amax if in_amax is NORMALISED else in_amax[P + dz:ns - P + dz])
    return acc


def _from_slices(out, s_out, res1, res2, B, C, D, HW, out_stats=None, pad=1):
    if out_stats is None:
        N.check(N.lib().ds_slices_to_volume(_p(out), _p(s_out), _p(res1), _p(res2), B, C, D, HW, int(pad), _stream()),
                "ds_slices_to_volume")
    else:
        if tuple(out_stats.shape) != (B, C, volume_stat_tiles(D, HW), 4):
            raise ValueError(f"out_stats must be {(B, C, volume_stat_tiles(D, HW), 4)}")
        N.check(N.lib().ds_slices_to_volume_stats(_p(out), _p(s_out), _p(res1), _p(res2), _p(out_stats), B, C, D, HW, int(pad),
                                                  _stream()), "ds_slices_to_volume_stats")
    return out


def resblock3d_fused(h, tab1, packs1, bias1, shift, packs2, bias2, w2, b2, kind2, res2=None, out=None, out_stats=None,
                     ws=None, eps=1e-5, circular=False):
    """ResnetBlockC on a volume (commonlayers.py:824-833) with both norms folded and the intermediate kept slice-major:
        S1 = SiLU(norm1(h))         by the volume -> slice copy (tab1 = ds_inorm_table rows of h's statistics)
        S2 = conv1(S1) + shift      three depth-tap launches; the last one leaves S2's tile statistics
        T  = per-slice table of norm2 over the sample's real slices, zero rows for the pad slices (ds_slice_tables)
        S3 = conv2(SiLU(norm2(S2))) three launches with the fused loader reading S2 in place
        out = S3 + h [+ res2]       by the slice -> volume copy, which also leaves out's statistics (out_stats)
    against norm, copy, 3 launches, copy, norm, copy, 3 launches, copy.  Plain loads, fp16x3 packings.  circular: periodic
    padding on all three axes (commonlayers.py:918-971) -- in the plane by the convolution's loader, along the depth by pad
    slices that hold wrapped copies: S1's from the volume -> slice copy, S2's from one small copy after conv1 (the pad rows of T
    are then the sample's row, not zeros)."""
    require_device(h, "h")
    B, C, D, H, W = h.shape
    if packs1[0].Cout != C or packs2[0].Cout != C:
        raise ValueError("resblock3d_fused keeps the channel count")
    ns, dev = B * (D + 2), h.device

    def take(shape):
        return torch.empty(shape, dtype=torch.float32, device=dev) if ws is None else ws.take(shape, dev)
    if out is None:
        out = torch.empty_like(h)
    s1 = take((ns, C, H, W))
    circ = 1 if circular else 0
    N.check(N.lib().ds_volume_to_slices_act(_p(s1), _p(h.contiguous()), _p(tab1), B, C, D, H * W, circ, _stream()),
            "ds_volume_to_slices_act")
    s2 = take((ns, C, H, W))
    if not circular:
        s2[0].zero_()                                       # the outermost pad slices are never written by the launches
        s2[ns - 1].zero_()
    ts = take((ns - 2, C, conv_tile_count(H, W), 4))
    rows, rows_buf = _slice_rows(shift, B, D, C, ws)
    _depth_taps(s1, s2, packs1, bias1, rows, N.DS_LOAD_PLAIN, bool(circular), tile_stats=ts, in_amax=NORMALISED)   # S1 = SiLU(norm1(h))
    if circular:
        N.check(N.lib().ds_wrap_pad_slices(_p(s2), B, C, D, H * W, _stream()), "ds_wrap_pad_slices")
    tab2 = take((ns, table_channels(C), 4))
    N.check(N.lib().ds_slice_tables(_p(tab2), _p(ts), _p(w2), _p(b2), B, C, D, ts.shape[2], D * H * W, float(eps), int(kind2),
                                    circ, _stream()), "ds_slice_tables")
    _depth_taps(s2, s1, packs2, bias2, None, N.DS_LOAD_PLAIN, bool(circular), prenorm=tab2)  # S1 is dead: reuse it for S3
    _from_slices(out, s1, h, res2, B, C, D, H * W, out_stats)
    if ws is not None:
        for t in (s1, s2, ts, tab2, rows_buf):
            if t is not None:
                ws.give(t)
    return out


def conv3d_mfma(x, packs, bias=None, shift=None, res1=None, res2=None, load_mode=N.DS_LOAD_PLAIN, circular=False, out=None,
                ws=None, out_stats=None, in_amax=None):
    """k x k x k 'same' convolution of a volume on the matrix cores (k = len(packs): 1, 3, 5, 7): k 2-D fp16x3 convolutions
    (one per depth tap; each a sum of shifted 3 x 3 blocks when k > 3) over a slice-major copy of the volume padded by k/2
    slices in depth (ds_volume_to_slices / ds_slices_to_volume).  Same arguments and
    fusions as conv3d; packs = pack_conv3d(weight).  ws: an optional buffer pool (take(shape, device) / give(tensor)) for
    the two slice copies, so that a captured loop allocates nothing.  out_stats: [B, Cout, volume_stat_tiles(D, H*W), 4],
    filled with the result's shifted partial sums (the consumer's norm table, ds_inorm_table with count D*H*W).
    in_amax: NORMALISED for a norm + SiLU output; otherwise the per-slice activation exponents are taken from a reduction over the
    slice copy (in a pool buffer when ws is given)."""
    require_device(x, "x")
    B, Cin, Din, Hi, Wi = x.shape
    Cout = packs[0].Cout
    if load_mode == N.DS_LOAD_MAXPOOL2:
        if Din % 2 or Hi % 2 or Wi % 2:
            raise ValueError("pooling load needs an even input volume")
        D, H, W, depth_mode = Din // 2, Hi // 2, Wi // 2, 1
    elif load_mode == N.DS_LOAD_UPSAMPLE2:
        D, H, W, depth_mode = 2 * Din, 2 * Hi, 2 * Wi, 2
    else:
        D, H, W, depth_mode = Din, Hi, Wi, 0
    if out is None:
        out = torch.empty((B, Cout, D, H, W), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (B, Cout, D, H, W):
        raise ValueError(f"out has shape {tuple(out.shape)}, expected {(B, Cout, D, H, W)}")
    for r in (res1, res2):
        if r is not None and tuple(r.shape) != (B, Cout, D, H, W):
            raise ValueError("residual shape mismatch")
    P = len(packs) // 2
    if circular and P > D:
        raise ValueError(f"periodic padding of {P} slices needs a depth of at least {P}; got {D}")
    ns = B * (D + 2 * P)

    def take(shape):
        return torch.empty(shape, dtype=torch.float32, device=x.device) if ws is None else ws.take(shape, x.device)
    s_in = take((ns, Cin, Hi, Wi))
    N.check(N.lib().ds_volume_to_slices(_p(s_in), _p(x.contiguous()), B, Cin, D, Hi * Wi, depth_mode, 1 if circular else 0, P,
                                        _stream()), "ds_volume_to_slices")
    s_out = take((ns, Cout, H, W))
    rows, rows_buf = _slice_rows(shift, B, D, Cout, ws, pad=P)
    am_buf = None
    if in_amax is not NORMALISED:
        if ws is None:
            in_amax = absmax_rows(s_in)
        else:
            am_buf = ws.take((ns,), x.device)
            in_amax = absmax_rows(s_in, out=amax_zero(am_buf.view(torch.int32)))
    _depth_taps(s_in, s_out, packs, bias, rows, load_mode, circular, in_amax=in_amax)
    _from_slices(out, s_out, res1, res2, B, Cout, D, H * W, out_stats, pad=P)
    if rows_buf is not None:
        ws.give(rows_buf)
    if am_buf is not None:
        ws.give(am_buf)
    if ws is not None:
        ws.give(s_in)
        ws.give(s_out)
    return out


def avgpool3d(x, out=None):
    """AvgPool3d(2) of a volume [B, C, 2D, 2H, 2W]."""
    require_device(x, "x")
    B, C, Di, Hi, Wi = x.shape
    if Di % 2 or Hi % 2 or Wi % 2:
        raise ValueError("avgpool3d needs even D, H, W")
    if out is None:
        out = torch.empty((B, C, Di // 2, Hi // 2, Wi // 2), dtype=torch.float32, device=x.device)
    N.check(N.lib().ds_avgpool3d(_p(out, "out"), _p(x, "x"), B * C, Di // 2, Hi // 2, Wi // 2, _stream()), "ds_avgpool3d")
    return out


def upsample3d(x, out=None):
    """Nearest x2 upsampling of a volume [B, C, D, H, W]."""
    require_device(x, "x")
    B, C, Di, Hi, Wi = x.shape
    if out is None:
        out = torch.empty((B, C, 2 * Di, 2 * Hi, 2 * Wi), dtype=torch.float32, device=x.device)
    N.check(N.lib().ds_upsample3d(_p(out, "out"), _p(x, "x"), B * C, Di, Hi, Wi, _stream()), "ds_upsample3d")
    return out


def gnorm1_stats(x, kind, eps=1e-5, stats=None, workspace=None):
    """Per-sample (mean, rstd) [kind 0] or (0, rms denominator) [kind 1] over (C, H, W)."""
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // max(B * C, 1)
    if stats is None:
        stats = torch.empty((B, 2), dtype=torch.float32, device=x.device)
    need = N.lib().ds_gnorm1_workspace_bytes(B)
    if workspace is None:
        workspace = torch.empty(need // 4, dtype=torch.float32, device=x.device)
    elif workspace.numel() * 4 < need:
        raise ValueError("gnorm1 workspace too small")
    N.check(N.lib().ds_gnorm1_stats(_p(stats), _p(workspace), _p(x), B, C, HW, float(eps), int(kind), _stream()),
            "ds_gnorm1_stats")
    return stats


def gnorm1_apply(x, stats, w, b, kind, pool=False, film=None, out=None):
    """kind 0: GroupNorm(1, C), kind 1: GroupRMSNorm(1, C), each followed by FiLM when `film` is given, then SiLU;
    kind 2: identity; then optional 2x2 average pooling.  film: [1 or B, 2C] rows of embed_linear(te)."""
    B, C, H, W = x.shape
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    if out is None:
        out = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=x.device)
    f1 = f2 = None
    stride = 0
    if film is not None:
        if film.dim() != 2 or film.shape[1] != 2 * C or film.shape[0] not in (1, B):
            raise ValueError("film must be [1 or B, 2C]")
        require_device(film, "film")
        stride = 0 if film.shape[0] == 1 else 2 * C
        f1, f2 = film.data_ptr(), film.data_ptr() + 4 * C
    N.check(N.lib().ds_gnorm1_apply(_p(out), _p(x), _p(stats), _p(w), _p(b), f1, f2, stride, B, C, H, W, int(kind),
                                    1 if pool else 0, _stream()), "ds_gnorm1_apply")
    return out


def concat2(a, b, out=None):
    """cat([a, b], dim=1) for [B, C, H, W] tensors."""
    B = a.shape[0]
    na, nb = a.numel() // max(B, 1), b.numel() // max(B, 1)
    if out is None:
        out = torch.empty((B, a.shape[1] + b.shape[1]) + tuple(a.shape[2:]), dtype=torch.float32, device=a.device)
    N.check(N.lib().ds_concat2(_p(out), _p(a), _p(b), B, na, nb, _stream()), "ds_concat2")
    return out


def add_act(a, add=None, act=0, out=None):
    M, Nn = a.shape
    out = torch.empty_like(a) if out is None else out
    rows = 0
    if add is not None:
        add = add.reshape(-1, Nn)
        rows = add.shape[0]
    N.check(N.lib().ds_add_act(_p(out), _p(a), _p(add), rows, M, Nn, int(act), _stream()), "ds_add_act")
    return out


def pack_conv_weight(w):
    """torch [Cout, Cin, k, k] (device, fp32) -> packed MFMA operand stream."""
    require_device(w, "conv weight")
    Cout, Cin, k, k2 = w.shape
    if k != k2 or k not in (1, 3):
        raise NotImplementedError(f"conv kernel {k}x{k2}: only 1x1 and 3x3 are implemented")
    n = N.lib().ds_conv2d_packed_floats(Cout, Cin, k)
    packed = torch.empty(n, dtype=torch.float32, device=w.device)
    N.check(N.lib().ds_conv2d_pack_weights(_p(packed), _p(w.contiguous()), Cout, Cin, k, _stream()),
            "ds_conv2d_pack_weights")
    return packed


class PackedConv:
    """A convolution weight repacked for one of the MFMA kernels.
    kind: "fp32" (exact-fp32 MFMA), "bf16x6" or "fp16x3" (fp32 emulated on the 16-bit matrix cores)."""
    __slots__ = ("data", "Cout", "Cin", "ks", "kind", "wshift", "up", "up_wshift", "subs")

    def __init__(self, data, Cout, Cin, ks, kind, wshift=0, up=None, up_wshift=0, subs=None):
        self.data, self.Cout, self.Cin, self.ks, self.kind, self.wshift = data, Cout, Cin, ks, kind, wshift
        # fp16x3 3x3 only: the four 2x2 parity kernels of "upsample x2, then this convolution" and their scale
        self.up, self.up_wshift = up, up_wshift
        # ks = 5, 7, ...: the kernel as ceil(ks/3)^2 zero-padded 3x3 blocks [(oy, ox, PackedConv 3x3)], see conv()
        self.subs = subs

    @property
    def x6(self):
        return self.kind == "bf16x6"


CONV_PRECISIONS = ("fp16x3", "bf16x6", "fp32")


def pack_conv(w, precision="bf16x6", upsampled=False):
    """Repack a torch conv weight [Cout, Cin, k, k] (device, fp32).  3x3 kernels honour
    `precision`; 1x1 kernels use the fp16x3 kernel for "fp16x3" and the exact-fp32 MFMA kernel otherwise.
    upsampled: the convolution follows a nearest x2 upsampling; the fp16x3 packing then also carries the
    collapsed parity kernels of ds_conv2d_h3_up."""
    require_device(w, "conv weight")
    if precision not in CONV_PRECISIONS:
        raise ValueError(f"unknown conv precision {precision!r}; choose from {CONV_PRECISIONS}")
    Cout, Cin, k, k2 = w.shape
    w = w.contiguous()
    if k != k2 or k % 2 == 0:
        raise NotImplementedError(f"conv kernel {k}x{k2}: square kernels of odd size are implemented")
    if k > 3:
        # k x k = sum of ceil(k/3)^2 shifted 3x3 convolutions over blocks of the taps (DS_TAP_OFFSET).  Block g starts at tap
        # min(3g, k - 3) -- the last block is pulled back inside the kernel, its taps already served by the previous block zeroed --
        # so no block reaches further out than the kernel itself does (k // 2 pixels: what periodic padding can wrap on a plane
        # that small); its centre tap start + 1 sits at offset start + 1 - k // 2
        if precision != "fp16x3":
            raise NotImplementedError(f"{k}x{k} kernels are implemented on the fp16x3 convolution only (conv_precision={precision!r})")
        ng = (k + 2) // 3
        starts = [min(3 * g, k - 3) for g in range(ng)]
        subs = []
        for gy, sy in enumerate(starts):
            for gx, sx in enumerate(starts):
                blk = w[:, :, sy:sy + 3, sx:sx + 3].clone()
                blk[:, :, :3 * gy - sy, :] = 0
                blk[:, :, :, :3 * gx - sx] = 0
                subs.append((sy + 1 - k // 2, sx + 1 - k // 2, pack_conv(blk.contiguous(), "fp16x3")))
        return PackedConv(None, Cout, Cin, k, "fp16x3", subs=subs)
    if precision == "bf16x6" and k == 3:
        nbytes = N.lib().ds_conv2d_x6_packed_bytes(Cout, Cin)
        data = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
        N.check(N.lib().ds_conv2d_x6_pack_weights(data.data_ptr(), _p(w), Cout, Cin, _stream()),
                "ds_conv2d_x6_pack_weights")
        return PackedConv(data, Cout, Cin, 3, "bf16x6")
    if precision == "fp16x3" and k in (1, 3):
        # per-layer power-of-two scale: largest weight lands in [2^13, 2^14), far inside fp16's
        # range, and typical weights get normal (not subnormal) low pieces
        wmax = float(w.abs().max())
        wshift = 0
        if wmax > 0 and wmax == wmax and wmax != float("inf"):
            import math
            wshift = max(-40, min(40, 13 - math.floor(math.log2(wmax))))
        size_fn, pack_fn = ((N.lib().ds_conv2d_h3_packed_bytes, N.lib().ds_conv2d_h3_pack_weights) if k == 3 else
                            (N.lib().ds_conv1x1_h3_packed_bytes, N.lib().ds_conv1x1_h3_pack_weights))
        data = torch.empty(size_fn(Cout, Cin) // 4, dtype=torch.float32, device=w.device)
        N.check(pack_fn(data.data_ptr(), _p(w), Cout, Cin, wshift, _stream()), "ds_conv*_h3_pack_weights")
        up, up_wshift = None, 0
        if upsampled and k == 3:
            up_wshift = max(-40, wshift - 2)          # a collapsed tap sums up to four weights: two bits of headroom
            up = torch.empty(N.lib().ds_conv2d_h3_up_packed_bytes(Cout, Cin) // 4, dtype=torch.float32, device=w.device)
            N.check(N.lib().ds_conv2d_h3_up_pack_weights(up.data_ptr(), _p(w), Cout, Cin, up_wshift, _stream()),
                    "ds_conv2d_h3_up_pack_weights")
        return PackedConv(data, Cout, Cin, k, "fp16x3", wshift, up, up_wshift)
    return PackedConv(pack_conv_weight(w), Cout, Cin, k, "fp32")


def conv(x, pw, **kw):
    """Dispatch on the packing: ds_conv2d_h3, ds_conv2d_x6 or ds_conv2d; kernels larger than 3x3 as a sum of shifted
    3x3 blocks accumulated in place (the first launch carries bias / shift / residuals, the last one the statistics)."""
    if pw.subs is None:
        return conv2d(x, pw.data, pw.Cout, pw.ks, kind=pw.kind, wshift=pw.wshift, w_up=pw.up, up_wshift=pw.up_wshift, **kw)
    if kw.get("res1_upsampled", False):
        raise NotImplementedError("res1_upsampled with kernels larger than 3x3")
    stats, out, out_amax = kw.pop("tile_stats", None), kw.pop("out", None), kw.pop("out_amax", None)
    first = dict(bias=kw.pop("bias", None), shift=kw.pop("shift", None), res1=kw.pop("res1", None), res2=kw.pop("res2", None))
    if kw.get("in_amax", None) is None and kw.get("prenorm", None) is None:
        kw["in_amax"] = absmax_rows(x)                                # one reduction for all the blocks
    n = len(pw.subs)
    for i, (oy, ox, sub) in enumerate(pw.subs):
        extra = first if i == 0 else dict(res1=out)
        out = conv2d(x, sub.data, sub.Cout, 3, kind="fp16x3", wshift=sub.wshift, tap_offset=(oy, ox), out=out,
                     tile_stats=stats if i == n - 1 else None, out_amax=out_amax if i == n - 1 else None, **extra, **kw)
    return out


def conv_direct(x, w, bias=None, circular=False, out=None):
    """3x3 'same' convolution with Cout <= 4, exact fp32, raw torch weight layout (output layers)."""
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    if tuple(w.shape) != (Cout, Cin, 3, 3):
        raise ValueError("conv_direct: weight must be [Cout, Cin, 3, 3]")
    if out is None:
        out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (B, Cout, H, W):
        raise ValueError(f"out has shape {tuple(out.shape)}, expected {(B, Cout, H, W)}")
    N.check(N.lib().ds_conv2d_direct(_p(out), _p(x), _p(w), _p(bias), B, Cin, Cout, H, W, 1 if circular else 0,
                                     _stream()), "ds_conv2d_direct")
    return out


def conv_tile_count(H, W):
    """Pixel tiles per channel plane in the fp16x3 kernels' tile_stats layout."""
    return N.lib().ds_conv_tile_count(int(H), int(W))


def table_channels(C):
    """Rows per sample of a prenorm table: channels padded to whole 16-channel chunks."""
    return (C + 15) // 16 * 16


def inorm_table(tile_stats, w, b, kind, count, eps=1e-5, out=None):
    """PUNetG norm table [B, ceil16(C), 4] from a convolution's tile statistics [B, C, ntiles, 4]: rows (M, A, C, 2^-k), k the
    sample's activation exponent for the consuming loader."""
    B, C, nt, _ = tile_stats.shape
    if out is None:
        out = torch.empty((B, table_channels(C), 4), dtype=torch.float32, device=tile_stats.device)
    elif tuple(out.shape) != (B, table_channels(C), 4):
        raise ValueError(f"table must be {(B, table_channels(C), 4)}")
    N.check(N.lib().ds_inorm_table(_p(out), _p(tile_stats), _p(w), _p(b), B, C, nt, int(count), float(eps), int(kind),
                                   _stream()), "ds_inorm_table")
    return out


def gnorm1_table(stats_a, w, b, kind, count, stats_b=None, film=None, eps=1e-5, out=None):
    """ADM norm table [B, ceil16(Ca+Cb), 4] from tile statistics of one tensor or of the two halves of a concat (rows as
    inorm_table)."""
    B, Ca, nta, _ = stats_a.shape
    Cb, ntb = (0, 0) if stats_b is None else (stats_b.shape[1], stats_b.shape[2])
    C = Ca + Cb
    if out is None:
        out = torch.empty((B, table_channels(C), 4), dtype=torch.float32, device=stats_a.device)
    elif tuple(out.shape) != (B, table_channels(C), 4):
        raise ValueError(f"table must be {(B, table_channels(C), 4)}")
    f1 = f2 = None
    stride = 0
    if film is not None:
        if film.dim() != 2 or film.shape[1] != 2 * C or film.shape[0] not in (1, B):
            raise ValueError("film must be [1 or B, 2C]")
        require_device(film, "film")
        stride = 0 if film.shape[0] == 1 else 2 * C
        f1, f2 = film.data_ptr(), film.data_ptr() + 4 * C
    N.check(N.lib().ds_gnorm1_table(_p(out), _p(stats_a), Ca, nta, _p(stats_b), Cb, ntb, _p(w), _p(b), f1, f2, stride,
                                    B, int(count), float(eps), int(kind), _stream()), "ds_gnorm1_table")
    return out


def gnorm1_stats_tiles(stats_a, kind, count, stats_b=None, eps=1e-5, stats=None):
    """gnorm1_stats [B, 2] from the tile statistics of the tensor's producer(s) instead of a pass over the tensor."""
    require_device(stats_a, "stats_a")
    B, Ca, nta, _ = stats_a.shape
    Cb, ntb = (0, 0) if stats_b is None else (stats_b.shape[1], stats_b.shape[2])
    if stats is None:
        stats = torch.empty((B, 2), dtype=torch.float32, device=stats_a.device)
    elif tuple(stats.shape) != (B, 2):
        raise ValueError(f"stats must be {(B, 2)}")
    N.check(N.lib().ds_gnorm1_stats_tiles(_p(stats), _p(stats_a), Ca, nta, _p(stats_b), Cb, ntb, B, int(count), float(eps),
                                          int(kind), _stream()), "ds_gnorm1_stats_tiles")
    return stats


def conv2d(x, w_packed, Cout, ks, bias=None, shift=None, res1=None, res2=None,
           load_mode=N.DS_LOAD_PLAIN, out=None, kind="fp32", wshift=0, prenorm=None, tile_stats=None, circular=False,
           res1_upsampled=False, w_up=None, up_wshift=0, tap_offset=None, in_amax=None, out_amax=None, amax_split=0):
    """'same' zero-padded conv; x [B, Cin, Hin, Win]; shift [1 or B, Cout] or None.
    fp16x3 kernels only: prenorm [B, ceil16(Cin), 4] (3x3) applies SiLU((x-M)*A+C) in the loader; tile_stats
    [B, Cout, conv_tile_count(H, W), 4] receives per-tile (K, sum(x-K), sum((x-K)^2), n) of the output;
    in_amax: int32 [B] per-sample max |x| (float bits) left by x's producer, NORMALISED for a norm + SiLU output, None = reduce
    x here (one extra read pass and an allocation: captured code passes slots); out_amax: zeroed int32 [B] slots that receive the
    per-sample max |out| for the next raw-input launch; amax_split (fp16x3 1x1 only): out_amax is [2, B] and channels >=
    amax_split report to its second row (the attention in-projection: q, k | v)."""
    B, Cin, Hin, Win = x.shape
    if load_mode in (N.DS_LOAD_MAXPOOL2, N.DS_LOAD_AVGPOOL2):
        if Hin % 2 or Win % 2:
            raise ValueError("pooling load needs even input H, W")
        if (load_mode == N.DS_LOAD_AVGPOOL2) != (kind == "fp16x3" and ks == 1):
            raise ValueError("load modes: AVGPOOL2 is for the fp16x3 1x1 kernel, MAXPOOL2 for the others")
        H, W = Hin // 2, Win // 2
    elif load_mode == N.DS_LOAD_UPSAMPLE2:
        H, W = Hin * 2, Win * 2
    else:
        H, W = Hin, Win
    if out is None:
        out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (B, Cout, H, W):
        raise ValueError(f"out has shape {tuple(out.shape)}, expected {(B, Cout, H, W)}")
    expect = {"bf16x6": lambda: N.lib().ds_conv2d_x6_packed_bytes(Cout, Cin) // 4,
              "fp16x3": lambda: (N.lib().ds_conv2d_h3_packed_bytes if ks == 3 else
                                 N.lib().ds_conv1x1_h3_packed_bytes)(Cout, Cin) // 4,
              "fp32": lambda: N.lib().ds_conv2d_packed_floats(Cout, Cin, ks)}[kind]()
    if w_packed.numel() != expect or (kind == "bf16x6" and ks != 3) or ks not in (1, 3):
        raise ValueError("packed weight size does not match (Cout, Cin, ks)")
    stride = 0
    if shift is not None:
        if shift.dim() != 2 or shift.shape[1] != Cout or shift.shape[0] not in (1, B):
            raise ValueError(f"shift must be [1 or B, Cout]; got {tuple(shift.shape)}")
        stride = 0 if shift.shape[0] == 1 else Cout
    if res1_upsampled:
        if kind != "fp16x3" or ks != 3 or res1 is None or H % 2 or W % 2 or tuple(res1.shape) != (B, Cout, H // 2, W // 2):
            raise ValueError("res1_upsampled: fp16x3 3x3 convolution with res1 of shape [B, Cout, H/2, W/2]")
    for r in ((res2,) if res1_upsampled else (res1, res2)):
        if r is not None and tuple(r.shape) != (B, Cout, H, W):
            raise ValueError("residual shape mismatch")
    if bias is not None and bias.numel() != Cout:
        raise ValueError("bias must have Cout entries")
    if (prenorm is not None or tile_stats is not None) and kind != "fp16x3":
        raise ValueError("prenorm / tile_stats are features of the fp16x3 kernels")
    pin = pout = None
    if kind == "fp16x3":
        pin = _in_amax(x, in_amax, B, raw=prenorm is None)
        if amax_split and (ks != 1 or amax_split % 64):
            raise ValueError("amax_split: the fp16x3 1x1 convolution, a multiple of 64")
        pout = _pi(out_amax, 2 * B if amax_split else B, "out_amax")
    elif out_amax is not None:
        raise ValueError("out_amax is a feature of the fp16x3 kernels (use absmax_rows on the result)")
    if circular and ks == 3 and kind != "fp16x3":
        raise NotImplementedError("periodic padding is implemented in the fp16x3 convolution only")
    if prenorm is not None and (ks != 3 or tuple(prenorm.shape) != (B, table_channels(Cin), 4)):
        raise ValueError(f"prenorm must be [B, ceil16(Cin), 4] on a 3x3 convolution; got {tuple(prenorm.shape)}")
    if tile_stats is not None and tuple(tile_stats.shape) != (B, Cout, conv_tile_count(H, W), 4):
        raise ValueError(f"tile_stats must be {(B, Cout, conv_tile_count(H, W), 4)}; got {tuple(tile_stats.shape)}")
    tap = 0
    if tap_offset is not None and tuple(tap_offset) != (0, 0):
        oy, ox = (int(v) for v in tap_offset)
        if kind != "fp16x3" or ks != 3 or not (-8 <= oy <= 7 and -8 <= ox <= 7):
            raise ValueError("tap_offset: fp16x3 3x3 convolution, offsets in [-8, 7]")
        tap = ((oy & 15) << 8) | ((ox & 15) << 12)                      # DS_TAP_OFFSET(oy, ox)
        w_up = None                                                      # the parity kernel has no offset form
    if kind == "fp16x3" and ks == 1:
        N.check(N.lib().ds_conv1x1_h3(_p(out), _p(x), _p(w_packed), int(wshift), _p(bias), _p(shift), stride,
                                      _p(res1), _p(res2), B, Cin, Cout, H, W, load_mode, _p(tile_stats), pin, pout,
                                      int(amax_split), _stream()), "ds_conv1x1_h3")
    elif kind == "fp16x3" and load_mode == N.DS_LOAD_UPSAMPLE2 and w_up is not None \
            and N.lib().ds_conv2d_h3_up_supported(Hin, Win):
        if w_up.numel() != N.lib().ds_conv2d_h3_up_packed_bytes(Cout, Cin) // 4:
            raise ValueError("w_up size does not match (Cout, Cin)")
        N.check(N.lib().ds_conv2d_h3_up(_p(out), _p(x), _p(w_up), int(up_wshift), _p(bias), _p(shift), stride,
                                        _p(res1), _p(res2), B, Cin, Cout, Hin, Win,
                                        (N.DS_PAD_CIRCULAR if circular else 0) | (N.DS_RES1_UPSAMPLED if res1_upsampled else 0),
                                        _p(prenorm), _p(tile_stats), pin, pout, _stream()), "ds_conv2d_h3_up")
    elif kind == "fp16x3":
        N.check(N.lib().ds_conv2d_h3(_p(out), _p(x), _p(w_packed), int(wshift), _p(bias), _p(shift), stride,
                                     _p(res1), _p(res2), B, Cin, Cout, H, W,
                                     load_mode | tap | (N.DS_PAD_CIRCULAR if circular else 0) | (N.DS_RES1_UPSAMPLED if res1_upsampled else 0),
                                     _p(prenorm), _p(tile_stats), pin, pout,
                                     _stream()), "ds_conv2d_h3")
    elif kind == "bf16x6":
        N.check(N.lib().ds_conv2d_x6(_p(out), _p(x), _p(w_packed), _p(bias), _p(shift), stride, _p(res1),
                                     _p(res2), B, Cin, Cout, H, W, load_mode, _stream()), "ds_conv2d_x6")
    else:
        N.check(N.lib().ds_conv2d(_p(out), _p(x), _p(w_packed), _p(bias), _p(shift), stride, _p(res1), _p(res2),
                                  B, Cin, Cout, H, W, ks, load_mode, _stream()), "ds_conv2d")
    return out


def conv_images_floats(B, C, H, W):
    """Floats of the pre-split image buffer of a [B, C, H, W] activation (ds_inorm_silu_images / conv_img)."""
    return N.lib().ds_conv_images_bytes(int(B), int(C), int(H), int(W)) // 4


def inorm_silu_images_supported(H, W):
    return bool(N.lib().ds_inorm_silu_images_supported(int(H), int(W)))


def inorm_silu_images(x, w, b, kind, eps=1e-5, out=None):
    """inorm_silu with the result written as the consuming convolution's pre-split fp16 hi / lo images (conv_img)."""
    require_device(x, "x")
    B, C, H, W = x.shape
    n = conv_images_floats(B, C, H, W)
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=x.device)
    elif out.numel() != n:
        raise ValueError("images buffer size does not match x")
    if w is not None and (w.numel() != C or b.numel() != C):
        raise ValueError("norm affine parameters must have C entries")
    N.check(N.lib().ds_inorm_silu_images(_p(out), _p(x), _p(w), _p(b), B, C, H, W, float(eps), int(kind), _stream()),
            "ds_inorm_silu_images")
    return out


def conv_img(images, pw, B, Cin, H, W, bias=None, shift=None, res1=None, res2=None, tile_stats=None, out=None,
             res1_upsampled=False, out_amax=None):
    """3x3 'same' zero-padded fp16x3 convolution whose input is given as pre-split fp16 hi / lo images (the layout
    ds_inorm_silu_images writes): patches are staged by LDS-DMA, no split in the kernel.  pw = pack_conv(weight, "fp16x3")."""
    require_device(images, "images")
    if pw.kind != "fp16x3" or pw.ks != 3 or pw.subs is not None:
        raise ValueError("conv_img: a 3x3 fp16x3 packing")
    Cout = pw.Cout
    if images.numel() != conv_images_floats(B, Cin, H, W):
        raise ValueError("images size does not match (B, Cin, H, W)")
    if out is None:
        out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=images.device)
    elif tuple(out.shape) != (B, Cout, H, W):
        raise ValueError(f"out has shape {tuple(out.shape)}, expected {(B, Cout, H, W)}")
    stride = 0
    if shift is not None:
        if shift.dim() != 2 or shift.shape[1] != Cout or shift.shape[0] not in (1, B):
            raise ValueError(f"shift must be [1 or B, Cout]; got {tuple(shift.shape)}")
        stride = 0 if shift.shape[0] == 1 else Cout
    if res1_upsampled and (res1 is None or H % 2 or W % 2 or tuple(res1.shape) != (B, Cout, H // 2, W // 2)):
        raise ValueError("res1_upsampled: res1 of shape [B, Cout, H/2, W/2]")
    for r in ((res2,) if res1_upsampled else (res1, res2)):
        if r is not None and tuple(r.shape) != (B, Cout, H, W):
            raise ValueError("residual shape mismatch")
    if tile_stats is not None and tuple(tile_stats.shape) != (B, Cout, conv_tile_count(H, W), 4):
        raise ValueError(f"tile_stats must be {(B, Cout, conv_tile_count(H, W), 4)}")
    N.check(N.lib().ds_conv2d_h3_img(_p(out), _p(images), _p(pw.data), int(pw.wshift), _p(bias), _p(shift), stride, _p(res1),
                                     _p(res2), B, Cin, Cout, H, W, N.DS_RES1_UPSAMPLED if res1_upsampled else 0,
                                     _p(tile_stats), _pi(out_amax, B, "out_amax"), _stream()), "ds_conv2d_h3_img")
    return out


def table_apply_images(x, table, out=None):
    """SiLU((x - M) * A + C) from a norm table [B, ceil16(C), 4] (inorm_table / gnorm1_table: the fused loader's arithmetic),
    written as the convolution's pre-split images (conv_img / conv_up_img).  Any plane size."""
    require_device(x, "x")
    require_device(table, "table")
    B, C, H, W = x.shape
    if tuple(table.shape) != (B, table_channels(C), 4):
        raise ValueError(f"table must be {(B, table_channels(C), 4)}; got {tuple(table.shape)}")
    n = conv_images_floats(B, C, H, W)
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=x.device)
    elif out.numel() != n:
        raise ValueError("images buffer size does not match x")
    N.check(N.lib().ds_table_apply_images(_p(out), _p(x), _p(table), B, C, H, W, _stream()), "ds_table_apply_images")
    return out


def conv_up_img_supported(pw, Hl, Wl):
    """conv_up_img takes this packing and low-resolution size."""
    return (pw.kind == "fp16x3" and pw.ks == 3 and pw.subs is None and pw.up is not None
            and bool(N.lib().ds_conv2d_h3_up_supported(int(Hl), int(Wl))))


def conv_up_img(images, pw, B, Cin, Hl, Wl, bias=None, shift=None, res1=None, res2=None, tile_stats=None, out=None,
                out_amax=None):
    """conv3x3(nearest_x2(a)) as the four collapsed parity kernels (ds_conv2d_h3_up) with the low-resolution activation a given as
    pre-split images; output [B, Cout, 2 Hl, 2 Wl].  pw = pack_conv(weight, "fp16x3", upsampled=True)."""
    require_device(images, "images")
    if not conv_up_img_supported(pw, Hl, Wl):
        raise ValueError("conv_up_img: a 3x3 fp16x3 packing with parity kernels and an input of whole 8x32 / 16x16 tiles")
    Cout, H, W = pw.Cout, 2 * Hl, 2 * Wl
    if images.numel() != conv_images_floats(B, Cin, Hl, Wl):
        raise ValueError("images size does not match (B, Cin, Hl, Wl)")
    if out is None:
        out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=images.device)
    elif tuple(out.shape) != (B, Cout, H, W):
        raise ValueError(f"out has shape {tuple(out.shape)}, expected {(B, Cout, H, W)}")
    stride = 0
    if shift is not None:
        if shift.dim() != 2 or shift.shape[1] != Cout or shift.shape[0] not in (1, B):
            raise ValueError(f"shift must be [1 or B, Cout]; got {tuple(shift.shape)}")
        stride = 0 if shift.shape[0] == 1 else Cout
    for r in (res1, res2):
        if r is not None and tuple(r.shape) != (B, Cout, H, W):
            raise ValueError("residual shape mismatch")
    if tile_stats is not None and tuple(tile_stats.shape) != (B, Cout, conv_tile_count(H, W), 4):
        raise ValueError(f"tile_stats must be {(B, Cout, conv_tile_count(H, W), 4)}")
    N.check(N.lib().ds_conv2d_h3_up_img(_p(out), _p(images), _p(pw.up), int(pw.up_wshift), _p(bias), _p(shift), stride, _p(res1),
                                        _p(res2), B, Cin, Cout, Hl, Wl, _p(tile_stats), _pi(out_amax, B, "out_amax"), _stream()),
            "ds_conv2d_h3_up_img")
    return out


def gnorm1_apply_images(x, stats, w, b, kind, pool=False, film=None, out=None):
    """gnorm1_apply (kinds 0 / 1) with the result written as the consuming convolution's pre-split images (conv_img)."""
    require_device(x, "x")
    B, C, H, W = x.shape
    if pool and (H % 2 or W % 2):
        raise ValueError("pooling needs even H, W")
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    n = conv_images_floats(B, C, Ho, Wo)
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=x.device)
    elif out.numel() != n:
        raise ValueError("images buffer size does not match x")
    f1 = f2 = None
    stride = 0
    if film is not None:
        if film.dim() != 2 or film.shape[1] != 2 * C or film.shape[0] not in (1, B):
            raise ValueError("film must be [1 or B, 2C]")
        require_device(film, "film")
        stride = 0 if film.shape[0] == 1 else 2 * C
        f1, f2 = film.data_ptr(), film.data_ptr() + 4 * C
    N.check(N.lib().ds_gnorm1_apply_images(_p(out), _p(x), _p(stats), _p(w), _p(b), f1, f2, stride, B, C, Ho, Wo, int(kind),
                                           1 if pool else 0, _stream()), "ds_gnorm1_apply_images")
    return out


def token_l2_normalize(x, c0, C, eps=1e-8, gain=1.0):
    """In place: channels [c0, c0+C) of x [B, Ctot, L] divided by (per-token L2 norm + eps), times gain."""
    require_device(x, "x")
    B, Ctot, L = x.shape
    N.check(N.lib().ds_token_l2_normalize(_p(x), B, Ctot, int(c0), int(C), L, float(eps), float(gain), _stream()),
            "ds_token_l2_normalize")
    return x


# Pre-split K / V images + LDS-DMA staging (ds_attention_h3_ws) against staging in every workgroup (ds_attention_h3),
# measured on MI355X (tools/attn_time.py): E = 256: 332 vs 353 us at L = 1024 (B = 64), 975 vs 1181 us at L = 4096
# (B = 16); E = 128 / 64 at L = 1024: 162 vs 160 / 91 vs 85 us.  A tile is re-split L/128 times without the images.
ATTN_IMAGES_MIN_L = 2048     # any head width
ATTN_IMAGES_MIN_L_WIDE = 1024    # E = 256


def _attention_uses_images(E, L, precision):
    if precision != "fp16x3" or E not in (32, 64, 128, 256) or L % 32:
        return False
    return L >= ATTN_IMAGES_MIN_L or (E == 256 and L >= ATTN_IMAGES_MIN_L_WIDE)


def attention_workspace_floats(B, E, L, precision="fp16x3"):
    """Floats of scratch attention() can use (0: none) -- for callers that keep buffers in a pool."""
    if _attention_uses_images(E, L, precision):
        return N.lib().ds_attention_h3_workspace_bytes(B, E, L) // 4
    return 0


def attention(qkv, E, out=None, precision="fp32", workspace=None, in_amax=None, out_amax=None):
    """qkv [B, 3E, L] channel-major -> out [B, E, L].  precision "fp16x3": split-fp16 MFMA
    (fp32-level accuracy, E <= 256); anything else, or wider heads: exact-fp32 MFMA.
    workspace: float tensor of attention_workspace_floats(...) elements; allocated here when needed and not given.
    in_amax: int32 [2, B] -- per-sample max |q, k| and max |v| (conv2d(..., amax_split=2E) leaves them), None = reduced here;
    out_amax [B]: as conv2d.  The fp16x3 kernels stage q, k and v times the sample's powers of two; the exact-fp32 kernels need
    none, and an out_amax request is served by a reduction over their result."""
    B, E3, L = qkv.shape
    if E3 != 3 * E:
        raise ValueError("qkv must be [B, 3E, L]")
    if out is None:
        out = torch.empty((B, E, L), dtype=torch.float32, device=qkv.device)
    h3 = precision == "fp16x3" and L % 32 == 0 and E in (32, 64, 128, 256)
    if h3:
        if in_amax is None:
            in_amax = amax_new(2 * B, qkv.device)
            absmax_rows(qkv[:, :2 * E], out=in_amax[:B])
            absmax_rows(qkv[:, 2 * E:], out=in_amax[B:])
        pin, pout = (None if in_amax is NORMALISED else _pi(in_amax, 2 * B, "in_amax")), _pi(out_amax, B, "out_amax")
    if h3 and _attention_uses_images(E, L, precision):
        need = attention_workspace_floats(B, E, L)
        if workspace is None:
            workspace = torch.empty(need, dtype=torch.float32, device=qkv.device)
        elif workspace.numel() < need:
            raise ValueError("attention workspace too small")
        N.check(N.lib().ds_attention_h3_ws(_p(out), _p(qkv), _p(workspace, "workspace"), B, E, L, pin, pout, _stream()),
                "ds_attention_h3_ws")
    elif h3:
        N.check(N.lib().ds_attention_h3(_p(out), _p(qkv), B, E, L, pin, pout, _stream()), "ds_attention_h3")
    else:
        if L % 32 != 0 or E not in (32, 64, 128, 256, 384, 512):
            N.check(N.lib().ds_attention_generic(_p(out), _p(qkv), B, E, L, _stream()), "ds_attention_generic")
        else:
            N.check(N.lib().ds_attention(_p(out), _p(qkv), B, E, L, _stream()), "ds_attention")
        if out_amax is not None:
            absmax_rows(out, B, out=out_amax)
    return out


def linear(x, w, b=None, act=0, out=None):
    M, K = x.shape
    Nn, K2 = w.shape
    if K != K2:
        raise ValueError("linear: inner dimensions differ")
    if out is None:
        out = torch.empty((M, Nn), dtype=torch.float32, device=x.device)
    N.check(N.lib().ds_linear(_p(out), _p(x), _p(w), _p(b), M, K, Nn, act, _stream()), "ds_linear")
    return out


def fourier_features(t, W, add=None, out=None):
    M, half = t.numel(), W.numel()
    if out is None:
        out = torch.empty((M, 2 * half), dtype=torch.float32, device=t.device)
    add_rows = 0
    if add is not None:
        add = add.reshape(-1, 2 * half)
        add_rows = add.shape[0]
    N.check(N.lib().ds_fourier_features(_p(out), _p(t), _p(W), _p(add), add_rows, M, half, _stream()),
            "ds_fourier_features")
    return out


def fourier_channels(x, W, out=None):
    """ConvolutionalFourierProjection: x [B, C, *spatial], W [C, D] -> [B, 2D, *spatial] = cat[sin, cos](x . 2*pi*W)."""
    require_device(x, "x")
    B, C = x.shape[0], x.shape[1]
    if W.dim() != 2 or W.shape[0] != C:
        raise ValueError(f"fourier_channels: W must be [{C}, D]; got {tuple(W.shape)}")
    D = W.shape[1]
    HW = x.numel() // max(B * C, 1)
    if out is None:
        out = torch.empty((B, 2 * D) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    N.check(N.lib().ds_fourier_channels(_p(out, "out"), _p(x, "x"), _p(W, "W"), B, C, D, HW, _stream()), "ds_fourier_channels")
    return out


class Graph:
    """A captured launch sequence (hipGraph) on torch's current stream."""

    def __init__(self):
        self._h = ctypes.c_void_p()
        self.nodes = 0

    @staticmethod
    def _allocations():
        return torch.cuda.memory_stats().get("allocation.all.allocated", 0)

    def __enter__(self):
        self._stream = _stream()
        self._alloc0 = self._allocations()
        N.check(N.lib().ds_graph_begin_capture(self._stream), "ds_graph_begin_capture")
        return self

    def __exit__(self, et, ev, tb):
        n = ctypes.c_int(0)
        rc = N.lib().ds_graph_end_capture(self._stream, ctypes.byref(self._h), ctypes.byref(n))
        if et is None:
            N.check(rc, "ds_graph_end_capture")
            # The graph bakes device addresses.  A tensor allocated inside the captured region belongs to torch's
            # caching allocator, which hands its block to someone else once the Python object dies -- while every
            # replay keeps writing there.  Captured code must take its buffers from a pre-filled workspace.
            made = self._allocations() - self._alloc0
            if made:
                raise RuntimeError(f"{made} device allocation(s) happened inside a captured region; the graph would "
                                   "write to memory it does not own on replay")
        self.nodes = n.value
        return False

    def launch(self):
        N.check(N.lib().ds_graph_launch(self._h, _stream()), "ds_graph_launch")

    def __del__(self):
        try:
            if self._h:
                N.lib().ds_graph_destroy(self._h)
        except Exception:
            pass
