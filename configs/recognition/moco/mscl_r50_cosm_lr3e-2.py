# MSCL SlowOnly-50 pre-training config for the MI355X build (BASELINE.json configs[4]).  The `model`, `optimizer`,
# `optimizer_config` and `lr_config` dicts are equal-valued to the reference's
# configs/recognition/moco/mscl_r50_cosm_lr3e-2.py:14-61,119-130 (pinned by tests/golden/ref_config_r50.json);
# the dataset section of the reference needs Megvii-internal storage and is replaced by synthetic clips.
_base_ = ['../../_base_/default_runtime.py']

ft_dim = 128
image_shape = (224, 224)
num_frames = 8
stride = 8
total_epochs = 200
dataset_size = 219136

rgb_recognizer = dict(
    type='MoCoV2',
    backbone=dict(
        type='ResNet3dSlowOnly', depth=50, pretrained=None, pretrained2d=False, lateral=False, num_stages=4,
        conv1_kernel=(5, 7, 7), conv1_stride_t=2, pool1_stride_t=1, spatial_strides=(1, 2, 2, 2), out_indices=(0, 1, 2, 3)),
    neck=dict(
        type='TPNMoCo', in_channels=[512, 1024, 2048], out_channels=128,
        sepc_cfg=dict(in_channels=[128, 128, 128], out_channels=128, stride=(1, 2, 2), iBN=False, Pconv_num=1)),
    moco_head=dict(type='MoCoHead', basename='', loss_cls=dict(type='CrossEntropyLoss_torch', ignore_index=-1)),
    im_key='imgs', dim_in=2048, dim=ft_dim,
    K=65536, m_base=0.994, max_iters=dataset_size * total_epochs, T=0.07, mlp=True, aux_info=[],
    aug=dict(type='IdentityAug'))
flow_recognizer = dict(
    type='MoCoV2',
    backbone=dict(type='resnet_flow.r2d_50'),
    neck=dict(type='BaseMoCo'),
    moco_head=dict(type='MoCoHead', basename='flow', loss_cls=dict(type='CrossEntropyLoss_torch', ignore_index=-1),
                   in_channels=256),
    im_key='imgs', dim_in=256, dim=ft_dim,
    K=65536, m_base=0.994, max_iters=dataset_size * total_epochs, T=0.07, mlp=True, aux_info=[],
    aug=dict(type='IdentityAug'))
model = dict(
    type='MSCLWithAug',
    recognizer=rgb_recognizer, recognizer_flow=flow_recognizer,
    moco_mx_head=dict(type='MSCLWithAugMxHead', basename='mx',
                      loss_cls=dict(type='CrossEntropyLoss_torch', ignore_index=-1), same_kn=True, T=0.07),
    sup_head=dict(type='MSCLWithAugPosHeadV2', basename='',
                  loss_pos=dict(type='CrossEntropyLoss_torch', ignore_index=-1),
                  bkb_channels=(None, 256), t=num_frames // 2, T=0.07,
                  aux_keys=dict(im_features=dict(q_mlvl='q_mlvl'),
                                base_flow_features=dict(q_mlvl='q_flow_mlvl'),
                                aug_flow_features=dict(q_mlvl='q_aug_flow_mlvl'))),
    im_key='imgs', flow_key='flow_imgs', aux_info=[], update_aug_flow=False, weight_aug_flow=(1.0, 1.0),
    aug=dict(type='SyncMoCoAugmentV5', crop_size=image_shape[0], sync_level=('batch', 'batch'),
             t=(num_frames, num_frames), flow_suffix='flow_imgs', weak_aug=(False, False), visualize=True),
    same_kn=True)

data = dict(videos_per_gpu=8, workers_per_gpu=0, train=dict(type='SyntheticClipPairs'),
            train_dataloader=dict(drop_last=True), val_dataloader=dict(drop_last=True))
evaluation = dict(interval=5, simple=True)

optimizer = dict(type='SGD', lr=0.0075, momentum=0.9, weight_decay=1e-4)
optimizer_config = dict(grad_clip=dict(max_norm=40, norm_type=2))
lr_config = dict(policy='CosineAnnealing', min_lr=0, warmup_iters=5, warmup_by_epoch=True)
checkpoint_config = dict(interval=10)
find_unused_parameters = True
