"""CPU: no wide store in the built gfx950 code has its data registers rewritten by a vector-ALU instruction fewer than two wait
states behind it (tools/check_store_hazard.py; the hazard and its measurement: DESIGN 5.26, tools/micro_store_hazard.hip).

The lint follows the control-flow graph (fall-through and target of every direct branch, loop back edges included); only an
indirect jump (s_setpc / s_swappc) ends a path unexamined."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_store_data_hazard_in_the_built_library(built_lib):
    import check_store_hazard as lint
    if not (os.path.exists(lint.OBJDUMP) and os.path.exists(lint.BUNDLER) and shutil.which("objcopy")):
        pytest.skip("llvm-objdump / clang-offload-bundler / objcopy not available")
    lib = os.environ.get("ECHOGLAD_LIB") or str(built_lib)       # the library the other tests load (conftest builds it on demand)
    if not os.path.exists(lib):
        pytest.skip(f"{lib} does not exist")
    found, n_stores = lint.scan(lib)
    assert n_stores > 100                                     # the scan saw the kernels' stores at all
    assert not found, "\n".join(f"{k}: {st}  ->  {wr} ({ws} wait states)" for k, st, wr, ws in found)
