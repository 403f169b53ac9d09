"""Re-export kept for parity with the reference's python/vshade package."""
from vulkan_forge import Renderer  # noqa: F401
