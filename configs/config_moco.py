"""DenseCL / MoCo backbone of the reference (its configs/config_moco.py:5-33): plain ResNet-50 (output stride 32);
forward_densecl only uses `.backbone` plus the DenseCL neck, the conv-less FCN head is a placeholder.  BASELINE config 5."""
from cp2_amd.config import segmentor_config

model = segmentor_config(depth=50, head="fcn", output_stride=32, head_channels=2048, contrast=False, num_convs=0,
                         concat_input=True)
