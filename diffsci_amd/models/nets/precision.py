"""Range guard of the default convolution arithmetic.

"fp16x3" computes fp32 products on the fp16 matrix cores from hi + lo fp16 pieces of each operand; its domain is
|activation| < 65504 (fp16's range).  Beyond it the kernels return inf / NaN -- never a wrong finite value -- so a
non-finite result from finite inputs is the signature of a range overflow.  The reference's torch convolutions take
the whole fp32 range (e.g. the VE parameterisation feeds c_in = 1 inputs of magnitude ~sigma_max, preconditioners.py:
56-136, and an untrained network can drive a trajectory to 1e6..1e8), so a drop-in must too: the guard switches the
network to "bf16x6" (exact 3-way bf16 split, fp32's exponent range, half the rate) once, warns, and the caller
recomputes.  A network already on a range-free precision, or non-finite inputs, are left alone: those NaNs are the
user's, as in the reference."""
import warnings

import torch

RANGE_FREE = "bf16x6"


def needs_escalation(model, out, *inputs):
    """True when `out` holds inf / NaN although every input is finite and `model` computes in fp16x3 with
    auto_precision on.  One device reduction and a host read: call it once per run, not per evaluation."""
    if getattr(model, "conv_precision", None) != "fp16x3" or not getattr(model, "auto_precision", False):
        return False
    if not torch.is_tensor(out) or bool(torch.isfinite(out).all()):
        return False
    return all(bool(torch.isfinite(t).all()) for t in inputs if torch.is_tensor(t) and t.is_floating_point())


def escalate(model):
    if getattr(model, "circular", False) or getattr(getattr(model, "config", None), "convolution_type", "") == "circular":
        raise FloatingPointError(
            "activations left the fp16x3 kernels' range (|x| >= 65504) and the range-free kernels do not implement "
            "circular padding; rescale the data (the EDM preconditioner keeps network inputs at unit variance)")
    model.conv_precision = RANGE_FREE
    warnings.warn("diffsci_amd: an activation exceeded the fp16x3 convolution range (|x| >= 65504); this network now "
                  f"runs conv_precision={RANGE_FREE!r} (no range limit, about half the convolution rate). Set "
                  "net.conv_precision yourself to choose, or net.auto_precision = False to get the inf/NaN instead.",
                  RuntimeWarning, stacklevel=3)
