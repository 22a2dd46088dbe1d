"""The C-ABI library: builds with hipcc, loads, and exports every symbol the header declares.
No compute is launched here (no GPU in this container)."""
import ctypes
import os
import subprocess

from echoglad_amd import _lib


def test_library_builds_and_loads(built_lib):
    assert os.path.exists(built_lib)
    lib = _lib.load()
    assert lib.eg_version() == _lib.ABI_VERSION           # _lib.load() refuses any other library
    hdr = open(_lib.HEADER_PATH).read()
    assert f"#define EG_ABI_VERSION {_lib.ABI_VERSION}" in hdr
    assert _lib.last_error() == "" or isinstance(_lib.last_error(), str)


def test_every_header_symbol_is_exported(built_lib):
    names = _lib.header_symbols()
    assert "eg_gcn_layer_fwd" in names and "eg_classifier_fwd" in names and "eg_topo_create" in names
    assert _lib.check_exports() == []
    for n in names:
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"


def test_header_signatures_and_exports_agree_in_both_directions(built_lib):
    """Every ctypes signature names a declared symbol, and the library exports no eg_* entry point the header hides."""
    declared = set(_lib.header_symbols())
    assert set(_lib.SIGNATURES) == declared, (sorted(set(_lib.SIGNATURES) - declared), sorted(declared - set(_lib.SIGNATURES)))
    out = subprocess.run(["nm", "-D", "--defined-only", str(built_lib)], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("eg_")
                and ln.split()[-2] in ("T", "t", "W")}
    # C++-mangled internals do not start with eg_; what remains must be exactly the public ABI
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))


def test_documents_name_only_entry_points_that_exist():
    """Every eg_* name in INTEGRATION.md, README.md and DESIGN.md's coverage / boundary sections is a declared entry point
    (a maintainer who follows the documents must not meet an undefined symbol)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    declared = set(_lib.header_symbols())
    internal = {"eg_graph", "eg_stream_t", "eg_cls_train_params", "eg_last_error", "eg_launch_layer_ps", "eg_launch_layer_sym",
                "eg_launch_bn_bwd", "eg_launch_dweight", "eg_allreduce_"}        # types, and internal launchers DESIGN.md names as such
    for doc in ("INTEGRATION.md", "README.md", "DESIGN.md"):
        text = open(os.path.join(root, doc)).read()
        names = set(re.findall(r"\b(eg_[a-z0-9_]+)", text))
        # wildcard mentions like eg_bce_logits_* / eg_heatmap_expect_*: a declared name must start with the stem
        bad = []
        for n in sorted(names):
            if n in declared or n in internal:
                continue
            if n.endswith("_") and any(d.startswith(n) for d in declared):
                continue
            if doc == "DESIGN.md" and n == "eg_graph_set_precision":      # named there as REMOVED (the history of an experiment)
                assert re.search(r"REMOVED[^.]*eg_graph_set_precision|eg_graph_set_precision[^.]*(removed|gone)", text)
                continue
            bad.append(n)
        assert not bad, f"{doc} names entry points that do not exist: {bad}"


def test_code_object_targets_gfx950(built_lib):
    out = subprocess.run(["strings", "-a", str(built_lib)], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_bad_arguments_return_error_codes(built_lib):
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.eg_topo_create(1, 7, 0, 0, 0, 0, 0, ctypes.byref(h)) == _lib.EG_ERR_ARG          # frame < 2
    assert "frame" in _lib.last_error()
    assert lib.eg_topo_create(224, 40, 0, 0, 0, 0, 0, ctypes.byref(h)) == _lib.EG_ERR_ARG        # naux too large
    # every flag of the reference's builder has stencil tables (round 4: connection nodes, 'grid-diagonal' levels): the
    # host tables are built, then -- without a GPU -- the device allocation fails
    for flags in ((1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1)):
        rc = lib.eg_topo_create(64, 5, 0, 0, *flags, ctypes.byref(h))
        assert rc in (_lib.EG_OK, _lib.EG_ERR_HIP), (flags, rc, _lib.last_error())
        if rc == _lib.EG_OK:
            assert lib.eg_graph_destroy(h) == _lib.EG_OK
    assert lib.eg_gcn_layer_fwd(None, 1, None, None, None, None, None, 0, 0, None, None) == _lib.EG_ERR_ARG
    assert lib.eg_graph_destroy(None) == _lib.EG_OK
