"""Small model configurations used by the golden vectors (pure data; shared by make_golden.py and the tests)."""

UNET_A = dict(  # full 4-level topology at 1/5 width: every block type of the MDM config appears
    in_channels=12, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1], num_res_blocks=2,
    channel_mult=[1, 2, 4, 4], dropout=0.1, num_head_channels=64, transformer_depth=1, context_dim=128,
    use_linear=True, use_checkpoint=False, temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
    use_relative_position=False, use_causal_attention=False, temporal_length=16, addition_attention=True,
    image_cross_attention=True, default_fs=24, fs_condition=True, class_label_condition=True)
UNET_A_SHAPE = dict(B=1, T=16, H=16, W=24)

UNET_B = dict(  # 2-level, batch of 3 modalities, short clip
    in_channels=12, out_channels=4, model_channels=64, attention_resolutions=[2, 1], num_res_blocks=1,
    channel_mult=[1, 2], dropout=0.1, num_head_channels=64, transformer_depth=1, context_dim=64,
    use_linear=True, use_checkpoint=False, temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
    use_relative_position=False, use_causal_attention=False, temporal_length=4, addition_attention=True,
    image_cross_attention=True, default_fs=24, fs_condition=True, class_label_condition=True)
UNET_B_SHAPE = dict(B=3, T=4, H=8, W=8)

VAE_DD = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)

DIFFUSION = dict(rescale_betas_zero_snr=True, parameterization="v", linear_start=0.00085, linear_end=0.012,
                 num_timesteps_cond=1, timesteps=1000, first_stage_key="video", cond_stage_key="caption",
                 cond_stage_trainable=False, conditioning_key="hybrid", image_size=[8, 8], channels=4,
                 scale_by_std=False, scale_factor=0.18215, use_ema=False, uncond_type="empty_seq",
                 use_dynamic_rescale=True, base_scale=0.3, fps_condition_type="fps", perframe_ae=True)

SAMPLER = dict(steps=2, eta=1.0, cfg_scale=7.5, guidance_rescale=0.7, spacing="uniform_trailing", fs=10,
               class_labels=[0, 500, 1])
SEED = 123

RESAMPLER = dict(dim=128, depth=2, dim_head=64, heads=2, num_queries=16, embedding_dim=96, output_dim=64,
                 ff_mult=4, video_length=4)

# 50-step runs of the same pipeline (the reference's real step count, render.sh:25-31): eta 1 with recorded noise, and eta 0
SAMPLER50 = dict(SAMPLER, steps=50)
# three-way classifier-free guidance (ddim_multiplecond.py:213-236)
THREEWAY = dict(SAMPLER, cfg_img=2.0)
# image_guided_synthesis-shaped driver run (virtual_pose_render.py:62-147) with fake CLIP towers (towers.py)
DRIVER = dict(pixels=64, clip_tokens=257, clip_dim=96, tower_seed_img=11, tower_seed_txt=12, cpu_seed=3, cfg_img=2.0,
              resampler=dict(dim=128, depth=1, dim_head=64, heads=2, num_queries=16, embedding_dim=96, output_dim=64,
                             ff_mult=2, video_length=4))
