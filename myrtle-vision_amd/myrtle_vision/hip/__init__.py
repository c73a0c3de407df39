"""Host-side binding of libmyrtle_vision_hip.so (ctypes) and the autograd functions built on it."""
