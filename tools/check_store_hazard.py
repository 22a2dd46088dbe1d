"""Lint over the built library's gfx950 code: a store of more than 8 bytes whose data registers are written again by a vector
ALU instruction with fewer than two wait states in between.  `tools/micro_store_hazard.hip` measured on gfx950 that ONE wait
state (what LLVM's hazard recogniser inserts for flat / global stores and for buffer stores with a literal scalar offset; it
inserts none for buffer stores with a register scalar offset) still lets the new value reach memory now and then; two are safe
(DESIGN 5.26).  Usage: python tools/check_store_hazard.py [library.so]; exit code 1 if anything is found.
The scan follows the control-flow graph: both successors of a conditional branch, the target of an unconditional one (loop
back edges included); only indirect jumps end a path."""
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


def parse(text):
    """-> {kernel: [(address, mnemonic, operands, branch target address | None, text)]} from llvm-objdump -d output."""
    kernels, cur, base = {}, None, {}
    for line in text.splitlines():
        s = line.strip()
        if not s:
            continue
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", s)
        if m:
            name = m.group(2)
            if not re.fullmatch(r"L\d+", name):        # a function symbol (local labels look like <L12>)
                cur = kernels.setdefault(name, [])
                base[name] = int(m.group(1), 16)
            continue
        if cur is None:
            continue
        code, _, comment = s.partition("//")
        code = code.strip()
        if not code:
            continue
        am = re.match(r"\s*([0-9A-Fa-f]+):", comment)
        addr = int(am.group(1), 16) if am else None
        parts = code.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        target = None
        if mn.startswith(("s_branch", "s_cbranch")):
            tm = re.search(r"<(.+?)\+0x([0-9a-fA-F]+)>", comment)
            if tm and tm.group(1) in base:
                target = base[tm.group(1)] + int(tm.group(2), 16)
            elif re.search(r"<(.+?)>", comment) and re.search(r"<(.+?)>", comment).group(1) in base:
                target = base[re.search(r"<(.+?)>", comment).group(1)]
        cur.append((addr, mn, ops, target, code))
    return kernels


def scan(lib):
    """Every path of the control-flow graph behind a wide store is followed until NEED wait states have passed: the fall-through
    AND the target of a conditional branch, the target of an unconditional one (so a store at the end of a loop body is checked
    against the first instructions of the next iteration).  A branch itself counts as one wait state.  Indirect jumps
    (s_setpc / s_swappc) and s_endpgm end a path."""
    text = "\n".join(subprocess.run([OBJDUMP, "-d", co], check=True, capture_output=True, text=True).stdout
                     for co in device_code(lib))
    findings, n_stores = [], 0
    for kernel, ins in parse(text).items():
        index = {a: i for i, (a, *_r) in enumerate(ins) if a is not None}
        for i, (_a, mn, ops, _t, code) in enumerate(ins):
            if not WIDE_STORE.match(mn):
                continue
            n_stores += 1
            data = store_data(mn, ops)
            seen = set()
            stack = [(i + 1, 0)]
            while stack:
                j, ws = stack.pop()
                if ws >= NEED or j >= len(ins) or (j, ws) in seen:
                    continue
                seen.add((j, ws))
                _aj, mj, oj, tj, cj = ins[j]
                if written(mj, oj) & data:
                    findings.append((kernel, code, cj, ws))
                if mj.startswith(("s_endpgm", "s_setpc", "s_swappc")):
                    continue
                passed = (int(oj[0], 0) + 1) if mj == "s_nop" and oj else 1
                if mj.startswith("s_branch"):
                    if tj in index:
                        stack.append((index[tj], ws + passed))
                    continue
                if mj.startswith("s_cbranch") and tj in index:
                    stack.append((index[tj], ws + passed))
                stack.append((j + 1, ws + passed))
    return findings, n_stores


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "echoglad_amd", "lib", "libechoglad_hip.so")
    found, n = scan(lib)
    for kernel, store, write, ws in found:
        print(f"{kernel[:70]}\n    {store}\n    {write}    <- {ws} wait state(s) behind the store")
    print(f"{len(found)} finding(s) over {n} wide stores in {lib}")
    sys.exit(1 if found else 0)
