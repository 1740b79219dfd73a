# BASELINE config 2 wording: "ResNet-50 + FCN decoder" (FCNHead with contrast=True, fcn_head.py:27,75-79), OS16.
norm_cfg = dict(type="BN", requires_grad=True)
model = dict(
    type="EncoderDecoder",
    backbone=dict(type="ResNet", depth=50, num_stages=4, out_indices=(0, 1, 2, 3), dilations=(1, 1, 1, 2),
                  strides=(1, 2, 2, 1), norm_cfg=norm_cfg, norm_eval=False, style="pytorch", contract_dilation=True),
    decode_head=dict(type="FCNHead", in_channels=2048, in_index=3, channels=512, num_convs=2, concat_input=True,
                     contrast=True, dropout_ratio=0.1, num_classes=2, norm_cfg=norm_cfg, align_corners=False),
    train_cfg=dict(),
    test_cfg=dict(mode="whole"),
)
