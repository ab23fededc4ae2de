"""Domain guards of the default convolution arithmetic.

"fp16x3" computes fp32 products on the fp16 matrix cores from hi + lo fp16 pieces of each operand.  fp16 has five exponent
bits: a value keeps its 22 bits for |x| in [2^-3, 2^16), degrades below (absolute floor 2^-25) and overflows above, while the
reference's torch convolutions take the whole fp32 range (raw user fields concatenated by PUNetGCond, punetg.py:719-735; c_in = 1
parameterisations, preconditioners.py:56-161; an untrained network can drive a trajectory to 1e6..1e8).  Three mechanisms
make the drop-in take that range too:

1. Activation exponents (ops.py / include/diffsci_hip.h: in_amax, out_amax).  Every launch whose input is not normalised by
   construction scales each SAMPLE by the power of two that puts its max |x| at 2^13 and undoes it exactly in the epilogue; the
   maxima come from the producing epilogue or one reduction.  No guard is involved: nothing to detect.
2. Norm-fed launches need no exponent as long as the norm's affine parameters are of ordinary size; `norms_in_window` checks
   that on the host once per parameter version, and blocks that fail it run standalone norms with measured exponents.
3. Two residual cases are DETECTED and re-run (one host read per run, not per evaluation):
   * `needs_escalation` / `escalate`: a non-finite result from finite inputs (an overflow somewhere unexpected) switches the
     network to "bf16x6" (exact 3-way bf16 split, fp32's exponent range, half the rate) once, with a warning;
   * `input_layer_flag` / `escalate_input`: the channels of a network INPUT differ by more than 2^14 in magnitude inside one
     sample (x at unit scale next to a field of 1e-8 whose weights compensate): one exponent per sample cannot serve both, so
     the input layer alone moves to the exact-fp32 MFMA kernel (ds_conv2d; ~1 % of an evaluation), with a warning.
A network already on a range-free precision, or non-finite inputs, are left alone: those NaNs are the user's, as in the reference."""
import warnings

import torch

RANGE_FREE = "bf16x6"


def needs_escalation(model, out, *inputs, result_checked=False):
    """True when the run has to be repeated: the input layer's channel-disparity flag is up (`escalate` then moves that layer
    to the exact kernel), or `out` holds inf / NaN although every input is finite and `model` computes in fp16x3 with
    auto_precision on.  result_checked: the run's last step kernel already looked at the result (ds_eval_coef.nonfinite ->
    `result_word`), so the two guard words arrive in ONE host read and no reduction runs; otherwise isfinite(out) is reduced
    here.  Call it once per run, not per evaluation; not at all while a stream capture is in progress (a user capturing
    net(x, t) in a graph of their own takes the kernels' raw behaviour)."""
    if getattr(model, "conv_precision", None) != "fp16x3" or not getattr(model, "auto_precision", False):
        return False
    if torch.cuda.is_current_stream_capturing():
        return False
    input_raised, result_raised = _read_guard_words(model)
    if input_raised and not getattr(model, "exact_input_layer", True):
        model.__dict__["_input_due"] = True
        return True
    if result_checked:
        if not result_raised:
            return False
    elif not torch.is_tensor(out) or bool(torch.isfinite(out).all()):
        return False
    return all(bool(torch.isfinite(t).all()) for t in inputs if torch.is_tensor(t) and t.is_floating_point())


def guard_words(model, device):
    """int32 [2] device words of one (model, device), at a fixed address (captured graphs write to them): [0] raised by the
    input layer's channel reduction (ops.input_amax / absmax_channels), [1] by a run's last step kernel when its result holds
    inf / NaN (ds_eval_coef.nonfinite)."""
    words = model.__dict__.setdefault("_input_flags", {})
    f = words.get(str(device))
    if f is None:
        with torch.inference_mode(False):
            f = words[str(device)] = torch.zeros(2, dtype=torch.int32, device=device)
    return f


def input_layer_flag(model, device):
    return guard_words(model, device)[0:1]


def result_word(model, device):
    """The word a tabulated run's last step kernel raises, or None when `model` takes no part in the guard."""
    if getattr(model, "conv_precision", None) != "fp16x3" or not getattr(model, "auto_precision", False):
        return None
    return guard_words(model, device)[1:2]


def _read_guard_words(model):
    """(input flag, result word) over the model's devices: one host read per device; raised words are cleared."""
    inp = res = False
    for f in getattr(model, "_input_flags", {}).values():
        a, b = f.tolist()
        if a or b:
            f.zero_()
        inp, res = inp or bool(a), res or bool(b)
    return inp, res


def _big_kernels(model):
    cfg = getattr(model, "config", None)
    return max((getattr(cfg, a, 3) or 3) for a in ("kernel_size", "in_out_kernel_size", "transition_kernel_size"))


def escalate(model):
    if getattr(model, "_input_flags", None) is not None and not getattr(model, "exact_input_layer", True) and _input_escalation_due(model):
        return escalate_input(model)
    if _big_kernels(model) > 3:
        raise FloatingPointError(
            "activations left the fp16x3 kernels' range and the range-free kernels implement 3x3 convolutions only "
            "(kernel_size / in_out_kernel_size / transition_kernel_size > 3); rescale the data")
    if getattr(model, "circular", False) or getattr(getattr(model, "config", None), "convolution_type", "") == "circular":
        raise FloatingPointError(
            "activations left the fp16x3 kernels' range (|x| >= 65504) and the range-free kernels do not implement "
            "circular padding; rescale the data (the EDM preconditioner keeps network inputs at unit variance)")
    model.conv_precision = RANGE_FREE
    model.__dict__.pop("_input_due", None)
    warnings.warn("diffsci_amd: an activation exceeded the fp16x3 convolution range (|x| >= 65504); this network now "
                  f"runs conv_precision={RANGE_FREE!r} (no range limit, about half the convolution rate). Set "
                  "net.conv_precision yourself to choose, or net.auto_precision = False to get the inf/NaN instead.",
                  RuntimeWarning, stacklevel=3)


def norms_in_window(cache, key, norms):
    """True when every norm in `norms` is affine-free or carries affine parameters of ordinary size (the larger of max |w|,
    max |b| within [2^-6, 2^6]): SiLU(norm(x) * w + b) then lies inside the fp16x3 window -- x = hi + lo keeps 22 bits for
    |x| in [2^-3, 2^16) and degrades gracefully to an absolute 2^-25 below -- so the launches that read it need no activation
    exponent.  Domain of that statement: the norm really normalises, i.e. the variance of its input is well above eps = 1e-5
    (rms >~ 1e-2).  Below, (x - mean) / sqrt(var + eps) shrinks with the input -- rms 1e-7 gives 3e-5 = 2^-15, where hi + lo keeps
    about ten bits -- and only the FOLDED-loader route follows it (its exponent comes from the statistics, ds_normtab.hip); the
    image / standalone routes this check serves pass in_amax = NORMALISED and lose accuracy gracefully there (absolute error
    2^-25 of unit scale, a tensor that carries no signal at fp32's own resolution of the surrounding unit-scale terms).  Stated in
    INTEGRATION.md.  Host check (one sync), cached per parameter version in `cache`."""
    sig = tuple((t.data_ptr(), t._version) for n in norms for t in (getattr(n, "weight", None), getattr(n, "bias", None))
                if t is not None)
    hit = cache.get(key)
    if hit is not None and hit[0] == sig:
        return hit[1]
    ok = True
    for n in norms:
        w, b = getattr(n, "weight", None), getattr(n, "bias", None)
        if w is not None:
            wm = float(w.detach().abs().max())
            bm = float(b.detach().abs().max()) if b is not None else 0.0
            ok = ok and (2.0 ** -6 <= max(wm, bm) <= 2.0 ** 6)
    cache[key] = (sig, ok)
    return ok


def _input_escalation_due(model):
    return bool(model.__dict__.pop("_input_due", False))


def escalate_input(model):
    """Move the network's input layer to the exact-fp32 kernel (see the module docstring, 3)."""
    if getattr(model, "circular", False) and not hasattr(model, "input_layer"):
        raise FloatingPointError("the channels of the network input differ by more than 2^14 in magnitude and the exact-fp32 "
                                 "input layer does not implement circular padding; rescale the condition fields")
    if getattr(getattr(model, "config", None), "in_out_kernel_size", 3) != 3:
        raise FloatingPointError("the channels of the network input differ by more than 2^14 in magnitude and the exact-fp32 "
                                 "input layer implements 3x3 kernels only; rescale the condition fields")
    model.exact_input_layer = True
    warnings.warn("diffsci_amd: the channels of the network input differ by more than 2^14 in magnitude within a sample (e.g. a "
                  "raw condition field next to c_in * x); the input layer now runs on the exact-fp32 kernel (about 1 % of an "
                  "evaluation). Rescale the field to avoid it.", RuntimeWarning, stacklevel=3)
