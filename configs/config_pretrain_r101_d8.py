# BASELINE config 4: ResNet-101 + dilated FCN head, 512x512 crops, output stride 8 (P = 4096).
norm_cfg = dict(type="BN", requires_grad=True)
model = dict(
    type="EncoderDecoder",
    backbone=dict(type="ResNet", depth=101, num_stages=4, out_indices=(0, 1, 2, 3), dilations=(1, 1, 2, 4),
                  strides=(1, 2, 1, 1), norm_cfg=norm_cfg, norm_eval=False, style="pytorch", contract_dilation=True),
    decode_head=dict(type="FCNHead", in_channels=2048, in_index=3, channels=512, num_convs=2, concat_input=True,
                     dilation=6, contrast=True, dropout_ratio=0.1, num_classes=2, norm_cfg=norm_cfg,
                     align_corners=False),
    train_cfg=dict(),
    test_cfg=dict(mode="whole"),
)
