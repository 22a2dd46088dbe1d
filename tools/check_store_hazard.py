"""Lint over the built library's gfx950 code: a store of more than 8 bytes whose data registers are written again by a vector
ALU instruction with fewer than two wait states in between.  `tools/micro_store_hazard.hip` measured on gfx950 that ONE wait
state (what LLVM's hazard recogniser inserts for flat / global stores and for buffer stores with a literal scalar offset; it
inserts none for buffer stores with a register scalar offset) still lets the new value reach memory now and then; two are safe
(DESIGN 5.26).  Usage: python tools/check_store_hazard.py [library.so]; exit code 1 if anything is found.
Linear scan per kernel (fall-through order); a label or branch between the store and the write ends the window."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
WIDE_STORE = re.compile(r"^(global|flat|buffer|scratch)_store_(dwordx3|dwordx4|b96|b128)\b")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
NEED = 2                     # wait states between the store and a write of its data


def regs_of(operand):
    m = REG.search(operand)
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def store_data(mn, ops):
    if mn.startswith("buffer_store"):
        return regs_of(ops[0])
    return regs_of(ops[1]) if len(ops) > 1 else set()          # global / flat / scratch: (address, data, ...)


def written(mn, ops):
    """VGPRs a vector-ALU instruction writes (first operand); compares, readlanes and no-destination forms write none."""
    if not mn.startswith("v_") or not ops:
        return set()
    if mn.startswith(("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane", "v_nop")):
        return set()
    w = regs_of(ops[0])
    if mn.startswith(("v_permlane16_swap", "v_permlane32_swap", "v_swap")) and len(ops) > 1:
        w |= regs_of(ops[1])
    return w


def device_code(lib):
    """The gfx950 code objects inside the host library: section .hip_fatbin holds one clang offload bundle per translation unit."""
    tmp = tempfile.mkdtemp(prefix="eg_hazard_")
    fat = os.path.join(tmp, "fatbin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    outs = []
    for i, a in enumerate(starts):
        b = starts[i + 1] if i + 1 < len(starts) else len(blob)
        piece = os.path.join(tmp, f"bundle{i}")
        open(piece, "wb").write(blob[a:b])
        out = os.path.join(tmp, f"gfx950_{i}.co")
        r = subprocess.run([BUNDLER, "--unbundle", "--type=o", "--input=" + piece, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--output=" + out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if r.returncode == 0 and os.path.getsize(out) > 0:
            outs.append(out)
    return outs


def scan(lib):
    text = "\n".join(subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
                     for co in device_code(lib))
    findings, kernel, window = [], "?", []          # window: [data regs, wait states so far, store text]
    n_stores = 0
    for line in text.splitlines():
        s = line.strip()
        if not s:
            continue
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", s)
        if m:
            name = m.group(1)
            if not name.startswith("L"):        # a function symbol (local labels look like <L12>)
                kernel = name
            window = []
            continue
        s = s.split("//")[0].strip()
        if not s:
            continue
        parts = s.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if mn.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc", "s_swappc")):
            window = []
            continue
        w = written(mn, ops)
        for data, ws, what in window:
            if ws < NEED and (w & data):
                findings.append((kernel, what, s, ws))
        passed = (int(ops[0], 0) + 1) if mn == "s_nop" and ops else 1
        window = [[d, ws + passed, what] for d, ws, what in window if ws + passed < NEED]
        if WIDE_STORE.match(mn):
            n_stores += 1
            window.append([store_data(mn, ops), 0, s])
    return findings, n_stores


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "echoglad_amd", "lib", "libechoglad_hip.so")
    found, n = scan(lib)
    for kernel, store, write, ws in found:
        print(f"{kernel[:70]}\n    {store}\n    {write}    <- {ws} wait state(s) behind the store")
    print(f"{len(found)} finding(s) over {n} wide stores in {lib}")
    sys.exit(1 if found else 0)
