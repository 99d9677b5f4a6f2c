"""SD-2.1-base / SDMatte architecture constants for the native engine.

The reference reads these from `<diffusers>/stable-diffusion-2-1-base/{unet,vae}/config.json`
(downloaded at run time, /root/reference/sdmatte_nodes.py:20-31, consumed at
/root/reference/src/modeling/SDMatte/meta_arch.py:95-118) and injects three SDMatte-specific
defaults in code (meta_arch.py:107-112).  There is no network on the GPU box, so the constants are
embedded here.  `tiny()` is a structurally identical, narrow model used
only by the parity tests so that the CPU oracle finishes in seconds.
"""
from dataclasses import dataclass, asdict, field
from typing import Tuple


@dataclass(frozen=True)
class SDMatteConfig:
    # --- VAE (AutoencoderKL) ---
    vae_channels: Tuple[int, ...] = (128, 256, 512, 512)
    vae_layers_per_block: int = 2
    vae_latent_channels: int = 4
    vae_groups: int = 32
    vae_eps: float = 1e-6
    vae_scaling_factor: float = 0.18215
    # --- U-Net (CustomUNet, /root/reference/src/utils/replace.py:125-362) ---
    unet_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    unet_heads: Tuple[int, ...] = (5, 10, 20, 20)      # config key `attention_head_dim` = head COUNT
    unet_layers_per_block: int = 2
    unet_groups: int = 32
    unet_res_eps: float = 1e-5                          # ResnetBlock2D GroupNorm eps
    unet_tf_gn_eps: float = 1e-6                        # Transformer2DModel GroupNorm eps
    unet_ln_eps: float = 1e-5
    cross_attention_dim: int = 1024                     # aux_conv_in out-channels (utils.py:34)
    unet_in_channels: int = 8                           # after replace_unet_conv_in (utils.py:13-30)
    unet_out_channels: int = 4
    point_embeddings_input_dim: int = 1680              # meta_arch.py:107-108 (unused on trimap path)
    bbox_embeddings_input_dim: int = 1280               # meta_arch.py:109-110
    attn_mask_value: float = -10000.0                   # replace.py:402
    name: str = "sd21-sdmatte"

    @property
    def time_embed_dim(self) -> int:
        return self.unet_channels[0] * 4

    @property
    def head_dim(self) -> int:
        return self.unet_channels[0] // self.unet_heads[0]

    @property
    def bbox_coord_embed_dim(self) -> int:
        # meta_arch.py:181-186 embeds each of 4 coords with dim 320 (= bbox_embeddings_input_dim/4)
        return self.bbox_embeddings_input_dim // 4

    def as_dict(self):
        return asdict(self)

    @staticmethod
    def full() -> "SDMatteConfig":
        return SDMatteConfig()

    @staticmethod
    def tiny() -> "SDMatteConfig":
        """Same graph, narrow channels (head_dim stays 64; GroupNorm stays 32 groups)."""
        return SDMatteConfig(
            vae_channels=(32, 64, 64, 64),
            unet_channels=(64, 128, 128, 128),
            unet_heads=(1, 2, 2, 2),
            cross_attention_dim=64,
            point_embeddings_input_dim=64,
            bbox_embeddings_input_dim=256,
            name="tiny",
        )

    @staticmethod
    def tiny_d512() -> "SDMatteConfig":
        """Tiny U-Net but a VAE whose mid-block attention has the real d=512 single head."""
        return SDMatteConfig(
            vae_channels=(32, 64, 512, 512),
            unet_channels=(64, 128, 128, 128),
            unet_heads=(1, 2, 2, 2),
            cross_attention_dim=64,
            point_embeddings_input_dim=64,
            bbox_embeddings_input_dim=256,
            name="tiny_d512",
        )


# (tests) tiny U-Net behind a VAE of 128 | 128 | 512 | 512 channels: the encoder's 3x3 convs then run on the fp8-residual producer / consumer
# kernel, the one that leaves the constant tiles of the trimap images to a fill kernel
def _tiny_wide_vae() -> "SDMatteConfig":
    return SDMatteConfig(
        vae_channels=(128, 128, 512, 512),
        unet_channels=(64, 128, 128, 128),
        unet_heads=(1, 2, 2, 2),
        cross_attention_dim=64,
        point_embeddings_input_dim=64,
        bbox_embeddings_input_dim=256,
        name="tiny_wide_vae",
    )


SDMatteConfig.tiny_wide_vae = staticmethod(_tiny_wide_vae)

INFERENCE_SIZES = [512, 640, 768, 896, 1024]   # sdmatte_nodes.py:226
