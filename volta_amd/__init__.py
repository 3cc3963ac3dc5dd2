"""volta_amd: MI355X-native (gfx950) pre-training step behind volta's BertConfig / BertForVLPreTraining API."""
from .config import BertConfig  # noqa: F401

__all__ = ["BertConfig"]
