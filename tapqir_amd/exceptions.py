"""
Exception types of the drop-in surface.  Names, constructor arguments and message texts are the interface of
tapqir/exceptions.py:9-39 (callers match on them: ``main.fit`` logs ``err.name``; ``Model.run`` converts a device
out-of-memory RuntimeError into ``CudaOutOfMemoryError``); ``HipExtensionError`` is this build's own.
"""

import os
from typing import Union

PathLike = Union[str, "os.PathLike[str]"]


class TapqirException(Exception):
    """Root of the package's exceptions; ``msg`` keeps the text for callers that log it."""

    def __init__(self, msg, *extra):
        if not msg:
            raise AssertionError("a TapqirException needs a message")
        super().__init__(msg, *extra)
        self.msg = msg


class TapqirFileNotFoundError(TapqirException):
    """A file of the workspace (``name`` = "data", "params", "summary", "model" ...) is not where it should be."""

    MESSAGE = "Unable to find {name} file '{path}'"

    def __init__(self, name: str, path: PathLike):
        super().__init__(self.MESSAGE.format(name=name, path=path))
        self.name, self.path = name, path


class CudaOutOfMemoryError(TapqirException):
    """Device memory is exhausted (the ROCm runtime words it "HIP out of memory")."""

    MESSAGE = "CUDA out of memory. Try to use smaller AOI/frame batch size"

    def __init__(self):
        super().__init__(self.MESSAGE)


class HipExtensionError(TapqirException):
    """libtapqir_hip.so is missing, failed to load or returned an error: the HIP path has NO fallback."""
