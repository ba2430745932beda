"""mamdr_amd -- MI355X-native hot path of MAMDR (RManLuo/MAMDR) behind a C ABI.

Only what the hot path needs lives here: csrc/ (HIP kernels + C ABI), the ctypes
binding, the device-resident tower engine and the host-side mirror of the
reference's run.py / model_zoo / utils interface.
"""
__version__ = "0.1.0"
