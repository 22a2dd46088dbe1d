"""CPU: no wide store in the built gfx950 code has its data registers rewritten by a vector-ALU instruction fewer than two wait
states behind it (tools/check_store_hazard.py; the hazard and its measurement: DESIGN 5.26, tools/micro_store_hazard.hip)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_store_data_hazard_in_the_built_library():
    import check_store_hazard as lint
    if not (os.path.exists(lint.OBJDUMP) and os.path.exists(lint.BUNDLER) and shutil.which("objcopy")):
        pytest.skip("llvm-objdump / clang-offload-bundler / objcopy not available")
    from echoglad_amd import _lib
    lib = _lib.library_path() if hasattr(_lib, "library_path") else os.path.join(ROOT, "echoglad_amd", "lib", "libechoglad_hip.so")
    assert os.path.exists(lib), "build the library first (python -m echoglad_amd.build)"
    found, n_stores = lint.scan(lib)
    assert n_stores > 100                                     # the scan saw the kernels' stores at all
    assert not found, "\n".join(f"{k}: {st}  ->  {wr} ({ws} wait states)" for k, st, wr, ws in found)
