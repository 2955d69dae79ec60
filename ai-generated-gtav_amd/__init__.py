"""gtav_amd — MI355X-native (gfx950) implementation of the AI-Generated-GTAV hot path:
spatio-temporal DiT denoising + ViT-VAE encode/decode behind the reference's Python class API.

Sub-modules are imported lazily; `gtav_amd.lib` loads the HIP shared library and raises if it is
missing (there is no CPU fallback in the product path)."""
__version__ = "0.1.0"
