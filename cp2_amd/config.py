"""Minimal stand-in for `mmengine.Config.fromfile` (reference main.py:338): a Python file whose
top-level names become config entries, with attribute access and `.get`."""
from __future__ import annotations

import os
import runpy


class ConfigDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


class Config(ConfigDict):
    @classmethod
    def fromfile(cls, path: str) -> "Config":
        if not os.path.isfile(path):
            raise FileNotFoundError(path)
        ns = runpy.run_path(path)
        return cls({k: _wrap(v) for k, v in ns.items() if not k.startswith("_") and not callable(v)
                    and not isinstance(v, type(os))})


def segmentor_config(depth: int = 50, head: str = "aspp", output_stride: int = 16, head_channels: int = 512,
                     contrast: bool = True, num_convs: int = 2, concat_input: bool = True, head_dilation: int = 1,
                     checkpoint=None, num_classes: int = 2) -> dict:
    """Model dict in the schema `build_segmentor` reads (the mmseg `EncoderDecoder` schema the reference's
    configs/*.py use), assembled from a few knobs.  output_stride 32 = plain ResNet; 16 / 8 dilate the last one /
    two stages instead of striding them (with mmseg's `contract_dilation`)."""
    if output_stride not in (8, 16, 32):
        raise ValueError(output_stride)
    plan = {32: ((1, 2, 2, 2), (1, 1, 1, 1)), 16: ((1, 2, 2, 1), (1, 1, 1, 2)), 8: ((1, 2, 1, 1), (1, 1, 2, 4))}[output_stride]
    bn = {"type": "BN", "requires_grad": True}
    feat = 512 if depth in (18, 34) else 2048
    backbone = {"type": "ResNet", "depth": depth, "num_stages": 4, "out_indices": (0, 1, 2, 3), "strides": plan[0],
                "dilations": plan[1], "norm_cfg": bn, "norm_eval": False, "style": "pytorch",
                "contract_dilation": output_stride != 32}
    if checkpoint:
        backbone["init_cfg"] = {"type": "Pretrained", "checkpoint": checkpoint}
    common = {"in_channels": feat, "in_index": 3, "channels": head_channels, "dropout_ratio": 0.1,
              "num_classes": num_classes, "norm_cfg": bn, "align_corners": False}
    if head == "aspp":
        decode = dict(common, type="ASPPHead", contrast=contrast, dilations=(1, 6, 12, 18))
    elif head == "fcn":
        decode = dict(common, type="FCNHead", contrast=contrast, num_convs=num_convs, concat_input=concat_input,
                      dilation=head_dilation)
    else:
        raise ValueError(head)
    return {"type": "EncoderDecoder", "backbone": backbone, "decode_head": decode, "auxiliary_head": None,
            "train_cfg": {}, "test_cfg": {"mode": "whole"}}
