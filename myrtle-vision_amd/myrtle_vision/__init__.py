"""myrtle_vision -- MI355X-native drop-in for the ViT hot path of MyrtleSoftware/myrtle-vision.

Same package name, model classes, config schema and checkpoint format as the reference
(``src/myrtle_vision``); the arithmetic runs in hand-written HIP kernels (``csrc/``) behind the C ABI
declared in ``include/myrtle_vision_hip.h``.  There is no CPU fallback: computing on a machine
without the built library / a GPU raises.
"""
__version__ = "0.1.0"
