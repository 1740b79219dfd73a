"""CP2 pre-training encoder of the reference (its configs/config_pretrain.py:5-35): ResNet-50 at output stride 16
with the ASPP head and its 128-d contrast projector.  ImageNet initialisation needs a LOCAL checkpoint path here
(the reference's 'torchvision://resnet50' downloads); pass --pretrain_from_scratch otherwise."""
from cp2_amd.config import segmentor_config

model = segmentor_config(depth=50, head="aspp", output_stride=16, checkpoint="torchvision://resnet50")
