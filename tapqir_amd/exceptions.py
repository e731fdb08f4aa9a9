"""
Exceptions of the drop-in surface (mirrors tapqir/exceptions.py:9-39: same names,
same constructor arguments, same messages).
"""

from pathlib import Path
from typing import Union


class TapqirException(Exception):
    """Base class of all tapqir exceptions."""

    def __init__(self, msg, *args):
        assert msg
        self.msg = msg
        super().__init__(msg, *args)


class TapqirFileNotFoundError(TapqirException):
    """A data / parameter / summary / model file is missing."""

    def __init__(self, name: str, path: Union[str, Path]):
        self.name = name
        self.path = path
        super().__init__(f"Unable to find {name} file '{path}'")


class CudaOutOfMemoryError(TapqirException):
    """Device memory exhausted (the ROCm runtime words it "HIP out of memory")."""

    def __init__(self):
        super().__init__("CUDA out of memory. Try to use smaller AOI/frame batch size")


class HipExtensionError(TapqirException):
    """The hand-written HIP library is missing or failed: there is NO fallback path."""
