"""Instruction census of a kernel's loops from the built library's gfx950 code: how many instructions of each class a wave
executes per iteration of a loop body (straight-line count between two addresses of the disassembly).
    python tools/isa_census.py <kernel substring>                 -> barriers and back edges (to find the loops)
    python tools/isa_census.py <kernel substring> <lo hex> <hi hex> [...]   -> class counts of [lo, hi) ranges
Used for DESIGN 9.4 (what a consumer / producer wave of the fused layer kernel executes per tile)."""
import collections
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_store_hazard as lint

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(mn):
    if mn.startswith("v_mfma"):
        return "mfma"
    if mn.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "v_readlane"
    if mn.startswith(("v_permlane", "v_mov_b32_dpp", "ds_bpermute", "ds_permute", "ds_swizzle")):
        return "cross-lane"
    if mn.startswith("v_pk_"):
        return "valu packed"
    if mn.startswith("v_"):
        return "valu"
    if mn.startswith("ds_read") or mn.startswith("ds_load"):
        return "lds read"
    if mn.startswith("ds_write") or mn.startswith("ds_store"):
        return "lds write"
    if mn.startswith("ds_"):
        return "lds other"
    if mn.startswith(("buffer_load", "global_load", "flat_load", "scratch_load")):
        return "vmem load"
    if mn.startswith(("buffer_store", "global_store", "flat_store", "scratch_store")):
        return "vmem store"
    if mn.startswith(("buffer_atomic", "global_atomic", "flat_atomic")):
        return "vmem atomic"
    if mn.startswith("s_waitcnt"):
        return "s_waitcnt"
    if mn.startswith("s_nop"):
        return "s_nop"
    if mn.startswith("s_barrier"):
        return "s_barrier"
    if mn.startswith(("s_branch", "s_cbranch")):
        return "branch"
    if mn.startswith("s_load") or mn.startswith("s_buffer_load"):
        return "smem load"
    if mn.startswith("s_"):
        return "salu"
    return "other"


def kernels(lib):
    text = "\n".join(subprocess.run([lint.OBJDUMP, "-d", co], check=True, capture_output=True, text=True).stdout
                     for co in lint.device_code(lib))
    return lint.parse(text)


def census(ins, lo, hi):
    c = collections.Counter()
    for a, mn, ops, t, code in ins:
        if a is not None and lo <= a < hi:
            c[classify(mn)] += 1
    return c


if __name__ == "__main__":
    lib = os.environ.get("ECHOGLAD_LIB", os.path.join(ROOT, "echoglad_amd", "lib", "libechoglad_hip.so"))
    ks = kernels(lib)
    names = [k for k in ks if sys.argv[1] in k]
    assert len(names) == 1, names
    ins = ks[names[0]]
    if len(sys.argv) == 2:
        print(names[0], len(ins), "instructions")
        for a, mn, ops, t, code in ins:
            if mn == "s_barrier" or (mn.startswith(("s_cbranch", "s_branch")) and t is not None and t < a):
                print(hex(a), code, "->", hex(t) if t else "")
    else:
        for lo, hi in zip(sys.argv[2::2], sys.argv[3::2]):
            c = census(ins, int(lo, 16), int(hi, 16))
            tot = sum(c.values())
            print(f"[{lo}, {hi}): {tot} instructions, {tot - c['mfma']} non-MFMA")
            for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
                print(f"    {k:14s} {v}")
