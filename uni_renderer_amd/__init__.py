"""uni_renderer_amd -- MI355X (gfx950) native dual-stream diffusion denoiser for Uni-Renderer's hot path.

Public surface = the reference's (models/controlnet.py, models/unet_2d_blocks.py, models/pipeline.py):
``UNet2DConditionModel``, ``AttributeEncoderModel``, ``AttributeDecoderModel``, the block factories and
``UniRendererPipeline``.  Compute happens in hand-written HIP kernels behind a C ABI
(``include/ur_kernels.h`` -> ``liburhip.so``); importing this package does not need a GPU, running it does.
"""
from .controlnet import AttributeDecoderModel, AttributeEncoderModel, UNet2DConditionModel, UNet2DConditionOutput
from .optim import FusedAdamW
from .pipeline import UniRendererPipeline
from .schedulers import DDIMScheduler, UniPCMultistepScheduler
from .vae import AutoencoderKL
from .unet_2d_blocks import (CrossAttnDownBlock2D, CrossAttnUpBlock2D, CrossAttnUpResBlock2D, DownBlock2D,
                             UNetMidBlock2DCrossAttn, UpBlock2D, UpResBlock2D, get_down_block, get_up_block)

__all__ = [
    "UNet2DConditionModel", "AttributeEncoderModel", "AttributeDecoderModel", "UNet2DConditionOutput",
    "CrossAttnDownBlock2D", "DownBlock2D", "UNetMidBlock2DCrossAttn", "UpBlock2D", "CrossAttnUpBlock2D",
    "UpResBlock2D", "CrossAttnUpResBlock2D", "get_down_block", "get_up_block",
    "UniRendererPipeline", "AutoencoderKL", "DDIMScheduler", "UniPCMultistepScheduler", "FusedAdamW",
]
__version__ = "0.1.0"
