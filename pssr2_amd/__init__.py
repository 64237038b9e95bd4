"""pssr2_amd — MI355X-native hot path of PSSR2 (ResUNet train/infer, MS-SSIM+L1 loss, crappifiers)."""
__version__ = "0.1.0"
