"""Drop-in alias: `import vulkan_forge` resolves to the MI355X-native implementation.

`import vulkan_forge._vulkan_forge as vf` (as the reference's tests do, e.g. tests/test_t41_scene.py:2)
returns the HIP-backed extension module.
"""
import sys as _sys

import vulkan_forge_amd as _impl
from vulkan_forge_amd import *  # noqa: F401,F403
from vulkan_forge_amd import __all__, __version__  # noqa: F401
from vulkan_forge_amd import _validate  # noqa: F401

_vulkan_forge = _impl._ext
_sys.modules[__name__ + "._vulkan_forge"] = _vulkan_forge
_sys.modules[__name__ + "._validate"] = _validate
