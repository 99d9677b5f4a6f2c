"""ComfyUI custom-node package: MI355X-native drop-in for the `Apply SDMatte` node of
flybirdxx/ComfyUI-SDMatte (same exports as /root/reference/__init__.py:1-6)."""
from .sdmatte_nodes import NODE_CLASS_MAPPINGS, NODE_DISPLAY_NAME_MAPPINGS

__all__ = [
    "NODE_CLASS_MAPPINGS",
    "NODE_DISPLAY_NAME_MAPPINGS",
]
