"""The C-ABI library loads and exports every symbol include/tapqir_hip.h declares (no GPU needed)."""

import ctypes
import os
import re

import pytest

from tapqir_amd import _lib
from tapqir_amd.exceptions import HipExtensionError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "tapqir_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tq_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    from tapqir_amd.build import build

    build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert _lib.load().tq_version() >= 100


def test_struct_sizes_match_the_c_header():
    """ctypes mirrors of the argument structs have the C compiler's layout."""
    import subprocess
    import tempfile

    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "tapqir_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(tq_ksmogn_args), offsetof(tq_ksmogn_args, m_kstride),
         offsetof(tq_ksmogn_args, scale), sizeof(tq_cosmos_args), offsetof(tq_cosmos_args, Nt),
         offsetof(tq_cosmos_args, seed), sizeof(tq_xtalk_args), offsetof(tq_xtalk_args, m_kstride),
         offsetof(tq_xtalk_args, scale), sizeof(tq_probs_args), sizeof(tq_glimpse_args),
         offsetof(tq_glimpse_args, offset_P));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        got = [int(v) for v in subprocess.check_output([exe]).split()]
    K, Cs, X = _lib.KsmognArgs, _lib.CosmosArgs, _lib.XtalkArgs
    want = [ctypes.sizeof(K), K.m_kstride.offset, K.scale.offset, ctypes.sizeof(Cs), Cs.Nt.offset, Cs.seed.offset,
            ctypes.sizeof(X), X.m_kstride.offset, X.scale.offset, ctypes.sizeof(_lib.ProbsArgs),
            ctypes.sizeof(_lib.GlimpseArgs), _lib.GlimpseArgs.offset_P.offset]
    assert got == want


def test_argument_validation_returns_error_codes_without_a_gpu():
    lib = _lib.load()
    a = _lib.KsmognArgs()
    assert lib.tq_ksmogn_log_prob(ctypes.byref(a), None) == 1  # TQ_ERR_ARG
    assert b"NULL" in lib.tq_last_error()
    c = _lib.CosmosArgs()
    assert lib.tq_cosmos_step(ctypes.byref(c), None) == 1


def test_no_cpu_fallback():
    """The product refuses to run the SVI step anywhere but on the HIP device."""
    from tapqir_amd.models.cosmos import cosmos
    from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

    m = cosmos(device="cpu")
    m.data = simulate(m, 2, 3, 1, 14, 0, TEST_PARAMS)
    with pytest.raises(HipExtensionError):
        m.init()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(HipExtensionError):
        _lib.load()
