"""Drop-in counterparts of the reference's ``src/models`` classes for the denoising path."""
from .attention import CrossAttention, SelfAttention
from .blocks import AttentionBlock, ResidualBlock
from .clip import CLIP, CLIPEmbedding, CLIPLayer
from .diffusion import Diffusion, FinalLayer
from .embeddings import TimeEmbedding
from .unet import SwitchSequential, UNet, Upsample
from .vae import Decoder, Encoder, VAEAttentionBlock, VAEResidualBlock

__all__ = ["SelfAttention", "CrossAttention", "ResidualBlock", "AttentionBlock", "TimeEmbedding",
           "Upsample", "SwitchSequential", "UNet", "FinalLayer", "Diffusion", "CLIPEmbedding", "CLIPLayer", "CLIP", "Decoder", "Encoder", "VAEResidualBlock", "VAEAttentionBlock"]
