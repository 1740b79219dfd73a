# ResNet-50 (output stride 16) + ASPP head with the 128-d contrast projector.
# Same schema and values as the reference's configs/config_pretrain.py:5-35 (BASELINE configs 2/3).
norm_cfg = dict(type="BN", requires_grad=True)
pretrain_path = "torchvision://resnet50"  # needs a LOCAL file path here; see cp2_amd/encoder.py ResNet.init_weights
model = dict(
    type="EncoderDecoder",
    backbone=dict(type="ResNet", depth=50, num_stages=4, out_indices=(0, 1, 2, 3), dilations=(1, 1, 1, 2),
                  strides=(1, 2, 2, 1), norm_cfg=norm_cfg, norm_eval=False, style="pytorch",
                  init_cfg=dict(type="Pretrained", checkpoint=pretrain_path), contract_dilation=True),
    decode_head=dict(type="ASPPHead", in_channels=2048, in_index=3, channels=512, contrast=True,
                     dilations=(1, 6, 12, 18), dropout_ratio=0.1, num_classes=2, norm_cfg=norm_cfg,
                     align_corners=False),
    auxiliary_head=None,
    train_cfg=dict(),
    test_cfg=dict(mode="whole"),
)
