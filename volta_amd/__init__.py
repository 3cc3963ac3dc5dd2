"""volta_amd: MI355X-native (gfx950) pre-training step behind volta's BertConfig / BertForVLPreTraining API."""
import os

# The engine runs four to five streams beside each other (compute, weight gradients, pipelined optimizer, gradient reduction, loader); HIP
# maps streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and streams that share one serialise (volta_amd/streams.py).  Ask for 8
# unless the caller chose a number; it takes effect when the HIP runtime has not been initialised yet (set it in the environment otherwise).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .config import BertConfig  # noqa: F401,E402

__all__ = ["BertConfig"]
