"""MI355X-native engine for the Video-Depth-Anything infer_video_depth hot path."""
from .config import get_config, ModelConfig  # noqa: F401
