# MSCL R3D-18 pre-training config for the MI355X build.  The `model`, `optimizer`, `optimizer_config`
# and `lr_config` dicts are equal-valued to the reference's
# configs/recognition/moco/mscl_r18_cosm_lr2e-2.py:15-55,114-123 (pinned by tests/golden/ref_config.json);
# the dataset section of the reference needs Megvii-internal storage and is replaced by synthetic clips.
_base_ = ['../../_base_/default_runtime.py']

ft_dim = 128
image_shape = (112, 112)
num_frames = 8
stride = 8
crop_shape = 128
total_epochs = 400
dataset_size = 219136

rgb_recognizer = dict(
    type='MoCoV2',
    backbone=dict(type='torchvision.r3d_18'),
    neck=dict(
        type='TPNMoCo', in_channels=[128, 256, 512], out_channels=128,
        sepc_cfg=dict(in_channels=[128, 128, 128], out_channels=128, stride=(2, 2, 2), iBN=False, Pconv_num=2)),
    moco_head=dict(type='MoCoHead', basename='', loss_cls=dict(type='CrossEntropyLoss_torch', ignore_index=-1)),
    im_key='imgs', dim_in=512, dim=ft_dim,
    K=65536, m_base=0.994, max_iters=dataset_size * total_epochs, T=0.07, mlp=True, aux_info=[],
    aug=dict(type='IdentityAug'))
flow_recognizer = dict(
    type='MoCoV2',
    backbone=dict(type='resnet_flow.r2d_18'),
    neck=dict(type='BaseMoCo'),
    moco_head=dict(type='MoCoHead', basename='flow', loss_cls=dict(type='CrossEntropyLoss_torch', ignore_index=-1)),
    im_key='imgs', dim_in=128, dim=ft_dim,
    K=65536, m_base=0.994, max_iters=dataset_size * total_epochs, T=0.07, mlp=True, aux_info=[],
    aug=dict(type='IdentityAug'))
model = dict(
    type='MSCLWithAug',
    recognizer=rgb_recognizer, recognizer_flow=flow_recognizer,
    moco_mx_head=dict(type='MSCLWithAugMxHead', basename='mx',
                      loss_cls=dict(type='CrossEntropyLoss_torch', ignore_index=-1), same_kn=True, T=0.07),
    sup_head=dict(type='MSCLWithAugPosHeadV2', basename='',
                  loss_pos=dict(type='CrossEntropyLoss_torch', ignore_index=-1),
                  bkb_channels=(None, None), t=num_frames // 2, T=0.07,
                  aux_keys=dict(im_features=dict(q_mlvl='q_mlvl'),
                                base_flow_features=dict(q_mlvl='q_flow_mlvl'),
                                aug_flow_features=dict(q_mlvl='q_aug_flow_mlvl'))),
    im_key='imgs', flow_key='flow_imgs', aux_info=[], update_aug_flow=False, weight_aug_flow=(1.0, 1.0),
    aug=dict(type='SyncMoCoAugmentV5', crop_size=image_shape[0], sync_level=('batch', 'batch'),
             t=(num_frames, num_frames), flow_suffix='flow_imgs', weak_aug=(False, False), visualize=True),
    same_kn=True)

# Data: the pipeline sections are the reference's (mscl_r18_cosm_lr2e-2.py:66-87), read by mscl_amd.data.MSCLPipeline.from_cfg;
# `NoriDecode` (Megvii-internal nori / redis storage) is replaced by the plain-file ClipStore, and the per-pixel steps (flow
# rotation augmentation, crop, resize, normalise) run on the GPU.  Without a store the benchmark feeds synthetic clips.
def _pipeline(sampler):
    return [
        dict(type='MatchFlow', gap=2, adjacent=8, flow_key='nids_flow'),
        sampler,
        dict(type='NoriDecode'),
        dict(type='NormFlowWithStidedAug', ratios=(0.2, 1.8), num_chunks=8, merge_aug=True),
        dict(type='MoCoRandomResizedCrop', area_range=(0.2, 1.0), flow_key='flow_imgs'),
        dict(type='MoCoResize', scale=image_shape, keep_ratio=False, flow_key='flow_imgs', suffix='_q'),
        dict(type='MoCoResize', scale=image_shape, keep_ratio=False, flow_key='flow_imgs', suffix='_k'),
        dict(type='MoCoNormalize', ori_flow=True),
        dict(type='Collect', keys=['imgs', 'flow_imgs'], meta_keys=[]),
        dict(type='ToTensor', keys=['imgs', 'flow_imgs'], batched=True),
    ]


train_pipeline = _pipeline(dict(type='TemporalShiftChosenSampleFrames', clip_len=num_frames, frame_interval=stride, num_clips=1, shift_range=1))
val_pipeline = _pipeline(dict(type='ChosenSampleFrames', clip_len=num_frames, frame_interval=stride, num_clips=1))
data = dict(videos_per_gpu=32, workers_per_gpu=0, train=dict(type='SyntheticClipPairs'),
            train_dataloader=dict(drop_last=True), val_dataloader=dict(drop_last=True))
evaluation = dict(interval=5, simple=True)

optimizer = dict(type='SGD', lr=0.02, momentum=0.9, weight_decay=1e-4)
optimizer_config = dict(grad_clip=dict(max_norm=40, norm_type=2))
lr_config = dict(policy='CosineAnnealing', min_lr=0, warmup_iters=5, warmup_by_epoch=True)
checkpoint_config = dict(interval=10)
find_unused_parameters = True
