# BASELINE config 1 (plumbing): ResNet-18, 64x64 crops, queue 1024.  decode_head.in_channels = 512.
norm_cfg = dict(type="BN", requires_grad=True)
model = dict(
    type="EncoderDecoder",
    backbone=dict(type="ResNet", depth=18, num_stages=4, out_indices=(0, 1, 2, 3), dilations=(1, 1, 1, 2),
                  strides=(1, 2, 2, 1), norm_cfg=norm_cfg, norm_eval=False, style="pytorch", contract_dilation=True),
    decode_head=dict(type="FCNHead", in_channels=512, in_index=3, channels=128, num_convs=1, concat_input=False,
                     contrast=True, dropout_ratio=0.1, num_classes=2, norm_cfg=norm_cfg, align_corners=False),
    train_cfg=dict(),
    test_cfg=dict(mode="whole"),
)
