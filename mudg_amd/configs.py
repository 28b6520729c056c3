"""The MuDG model configurations the benchmarks and the driver use, as plain dicts (same constructor kwargs as the
reference's YAML: configs/stage2-1024_mdm_waymo_infer.yaml:1-100 and the stage-1 512 config)."""
import copy

UNET_MDM = dict(
    in_channels=12, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
    channel_mult=[1, 2, 4, 4], dropout=0.1, num_head_channels=64, transformer_depth=1, context_dim=1024,
    use_linear=True, use_checkpoint=False, temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
    use_relative_position=False, use_causal_attention=False, temporal_length=16, addition_attention=True,
    image_cross_attention=True, default_fs=24, fs_condition=True, class_label_condition=True)

VAE_DDCONFIG = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                    ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)

_PLACEHOLDER = {"target": "torch.nn.Identity"}      # CLIP text/image encoders and the Resampler are out of scope


def latent_visual_diffusion(resolution="1024", with_conditioners=None):
    """Constructor kwargs of lvdm.models.ddpm3d.LatentVisualDiffusion for MDM1024 (576x1024) or MDM512 (320x512)."""
    big = str(resolution) == "1024"
    cond = with_conditioners or {}
    return dict(
        rescale_betas_zero_snr=True, parameterization="v", linear_start=0.00085, linear_end=0.012,
        num_timesteps_cond=1, timesteps=1000, first_stage_key="video", cond_stage_key="caption",
        cond_stage_trainable=False, conditioning_key="hybrid", image_size=[72, 128] if big else [40, 64], channels=4,
        scale_by_std=False, scale_factor=0.18215, use_ema=False, uncond_type="empty_seq", use_dynamic_rescale=True,
        base_scale=0.3 if big else 0.7, fps_condition_type="fps", perframe_ae=True,
        unet_config={"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": copy.deepcopy(UNET_MDM)},
        first_stage_config={"target": "lvdm.models.autoencoder.AutoencoderKL",
                            "params": {"embed_dim": 4, "monitor": "val/rec_loss",
                                       "ddconfig": copy.deepcopy(VAE_DDCONFIG),
                                       "lossconfig": {"target": "torch.nn.Identity"}}},
        cond_stage_config=cond.get("cond_stage_config", _PLACEHOLDER),
        img_cond_stage_config=cond.get("img_cond_stage_config", _PLACEHOLDER),
        image_proj_stage_config=cond.get("image_proj_stage_config", _PLACEHOLDER))


LATENT_SHAPE = {"1024": (4, 16, 72, 128), "512": (4, 16, 40, 64)}
# Algorithmic work, FLOP = 2 MAC, measured on the reference graph (SURVEY.md §8(d) / BASELINE.md §2)
UNET_TFLOP = {"1024": 52.340, "512": 12.604}
VAE_DECODE_TFLOP_PER_FRAME = {"1024": 5.754, "512": 1.564}
