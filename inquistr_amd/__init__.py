"""inquistr_amd — MI355X-native implementation of inquiSTR's `call` hot path.

Only what the path needs: csrc/ (HIP kernels + the C ABI of include/inquistr_hip.h),
the ctypes binding, the batch layout and the host-side mirror of the reference's
`call::genotype_repeats` interface.
"""
__version__ = "0.1.0"
