"""The C-ABI library loads and exports every symbol include/qcpinn_hip.h declares (no GPU needed:
nothing is computed here)."""
import ctypes
import os
import re

from conftest import ROOT, pkg


def test_library_exports_every_declared_symbol():
    L = pkg("hip.lib")
    lib = L.load()
    header = open(os.path.join(ROOT, "include", "qcpinn_hip.h")).read()
    declared = set(re.findall(r"\b(qc_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(L.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.qc_version() == 4
    assert lib.qc_error_string(-1).decode() == "invalid argument"


def test_struct_sizes_match_header_layout():
    L = pkg("hip.lib")
    assert ctypes.sizeof(L.QcPde) == 72
    assert ctypes.sizeof(L.QcOptHyper) == 56


def test_argument_errors_do_not_need_a_gpu():
    L = pkg("hip.lib")
    lib = L.load()
    h = ctypes.c_void_p()
    assert lib.qc_program_create(None, 0, 4, 12, ctypes.byref(h)) == -1
    assert lib.qc_forward_expval(None, None, None, None, None, 0, None, 0, None) == -1
    assert lib.qc_reduce_rows(None, 0, 0, 0, None, None) == -1
