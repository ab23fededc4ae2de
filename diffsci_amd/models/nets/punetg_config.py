"""PUNetGConfig -- same constructor arguments, defaults and (de)serialisation as the reference
(diffsci/models/nets/punetg_config.py:8-122).  Options outside the HIP path's coverage are
accepted here and rejected with a clear error when the network is built."""
from typing import Any
import pathlib

import yaml

_FIELDS = dict(
    input_channels=1, output_channels=1, dimension=2, model_channels=64,
    channel_expansion=(2, 4),
    number_resnet_downward_block=2, number_resnet_upward_block=2,
    number_resnet_attn_block=2, number_resnet_before_attn_block=2,
    number_resnet_after_attn_block=2,
    kernel_size=3, in_out_kernel_size=3, in_embedding=False,
    time_projection_scale=30.0, input_projection_scale=1.0,
    transition_scale_factor=2, transition_kernel_size=3,
    dropout=0.0, cond_dropout=0.0, cond_drop=0.0, cond_drop_learnable=True,
    first_resblock_norm="GroupLN", second_resblock_norm="GroupRMS", affine_norm=True,
    convolution_type="default", num_groups=1, attn_residual=False, attn_type="default",
    bias=True)


class PUNetGConfig(object):
    # positional order and defaults of the reference's constructor (punetg_config.py:8-38)
    def __init__(self, input_channels=1, output_channels=1, dimension=2, model_channels=64, channel_expansion=(2, 4),
                 number_resnet_downward_block=2, number_resnet_upward_block=2, number_resnet_attn_block=2,
                 number_resnet_before_attn_block=2, number_resnet_after_attn_block=2, kernel_size=3, in_out_kernel_size=3,
                 in_embedding=False, time_projection_scale=30.0, input_projection_scale=1.0, transition_scale_factor=2,
                 transition_kernel_size=3, dropout=0.0, cond_dropout=0.0, cond_drop=0.0, cond_drop_learnable=True,
                 first_resblock_norm="GroupLN", second_resblock_norm="GroupRMS", affine_norm=True, convolution_type="default",
                 num_groups=1, attn_residual=False, attn_type="default", bias=True):
        given = locals()
        for k in _FIELDS:
            v = given[k]
            if k == "channel_expansion":
                v = list(v)
            setattr(self, k, v)

    @property
    def extended_channel_expansion(self):
        return [1] + list(self.channel_expansion)

    @property
    def magnitude_preserving(self):
        return self.convolution_type == "mp"

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
        """None if the HIP path implements this configuration, else why not."""
        checks = [
            (self.dimension in (2, 3), "2-D fields or 3-D volumes (dimension 2 or 3)"),
            (self.convolution_type in ("default", "circular", "mp"), "convolution_type 'default', 'circular' or 'mp'"),
            (all(k in (1, 3, 5, 7) for k in (self.kernel_size, self.in_out_kernel_size)) and self.transition_kernel_size in (3, 5, 7),
             "kernel_size / in_out_kernel_size 1, 3, 5 or 7, transition_kernel_size 3, 5 or 7"),
            (self.transition_scale_factor == 2, "transition_scale_factor=2"),
            (not self.in_embedding or not self.bias,
             "in_embedding only with bias=False (the reference's ConvolutionalFourierProjection raises with bias=True, "
             "commonlayers.py:251-253)"),
            (self.attn_type in ("default", "cosine"), "attn_type 'default' or 'cosine'"),
        ]
        bad = [msg for ok, msg in checks if not ok]
        return None if not bad else "diffsci_amd PUNetG supports: " + "; ".join(bad)
