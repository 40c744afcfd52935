"""alg_amd -- MI355X-native Adaptive Low-pass Guidance (ALG) image-to-video sampler.

Only what the hot path needs: the HIP kernels + C ABI (``csrc/``, ``libalg_hip.so``), and the host-side
mirror of the reference interface (``lp_utils``, the CogVideoX pipeline / transformer / scheduler, and the Wan /
HunyuanVideo samplers with their DiT forwards and UniPC / flow-match Euler schedulers, the CogVideoX VAE decoder).
"""
from . import lp_utils  # noqa: F401
from .image_encoder_clip import CLIPImageProcessor, CLIPVisionEncoderConfig, CLIPVisionModel  # noqa: F401
from .text_encoder_clip import CLIPTextEncoderConfig, CLIPTextModel  # noqa: F401
from .text_encoder_llava import LlavaConfig, LlavaForConditionalGeneration  # noqa: F401
from .text_encoder_t5 import T5EncoderConfig, T5EncoderModel, UMT5EncoderModel  # noqa: F401
from .autoencoder_kl_cogvideox import AutoencoderKLCogVideoX, AutoencoderKLCogVideoXConfig  # noqa: F401
from .autoencoder_kl_wan import AutoencoderKLWan, AutoencoderKLWanConfig  # noqa: F401
from .autoencoder_kl_hunyuan_video import AutoencoderKLHunyuanVideo, AutoencoderKLHunyuanVideoConfig  # noqa: F401
from ._lib import AlgHipError, build_library, load_library  # noqa: F401
from .pipeline_cogvideox_image2video_lowpass import CogVideoXImageToVideoPipeline, CogVideoXPipelineOutput  # noqa: F401
from .pipeline_hunyuan_video_image2video_lowpass import HunyuanVideoImageToVideoPipeline  # noqa: F401
from .pipeline_wan_image2video_lowpass import WanImageToVideoPipeline  # noqa: F401
from .schedulers import (CogVideoXDDIMScheduler, CogVideoXDPMScheduler, FlowMatchEulerDiscreteScheduler,  # noqa: F401
                         UniPCMultistepScheduler)
from .transformer_cogvideox import CogVideoXTransformer3DModel, CogVideoXTransformerConfig  # noqa: F401
from .transformer_hunyuan_video import HunyuanVideoTransformer3DModel, HunyuanVideoTransformerConfig  # noqa: F401
from .transformer_wan import WanTransformer3DModel, WanTransformerConfig  # noqa: F401

__version__ = "0.1.0"
