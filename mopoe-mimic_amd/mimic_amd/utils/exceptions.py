"""Exceptions of the epoch driver's caller contract (reference: mimic/utils/exceptions.py)."""


class NaNInLatent(Exception):
    """raised when an encoder's latents contain NaN (reference: utils.check_latents, utils.py:201-208)"""


class CudaOutOfMemory(Exception):
    """raised when the device allocator runs out of memory and batch_size > 10 (run_epochs.py:37-49)"""
