# DenseCL / MoCo backbone config (reference configs/config_moco.py:5-33): ResNet-50 at output stride 32,
# FCN head with no convs (only `.backbone` and the DenseCL neck are used by forward_densecl).  BASELINE config 5.
norm_cfg = dict(type="BN", requires_grad=True)
model = dict(
    type="EncoderDecoder",
    backbone=dict(type="ResNet", depth=50, num_stages=4, out_indices=(0, 1, 2, 3), dilations=(1, 1, 1, 1),
                  strides=(1, 2, 2, 2), norm_cfg=norm_cfg, norm_eval=False, style="pytorch", contract_dilation=False),
    decode_head=dict(type="FCNHead", num_convs=0, in_channels=2048, in_index=3, channels=2048, num_classes=2,
                     norm_cfg=norm_cfg),
    train_cfg=dict(),
    test_cfg=dict(mode="whole"),
)
