#!/usr/bin/env python3
"""Where do a kernel's scratch (spill) instructions sit -- inside its hot loop or around it?

    python tools/isa_loops.py [substring of a kernel name ...]

Reads the gfx950 code objects out of csrc/libgs_amd.so (as tests/test_kernel_resources.py does), disassembles them with
llvm-objdump and, per kernel, finds the loops (a backward branch and its target) and reports for every loop its size, how
many v_mfma / v_exp / scratch_ / buffer / global / ds instructions it holds.  `hot_loop()` picks the loop the tests assert
on: the innermost loop that contains a given marker instruction (v_mfma_* for the SH backward, v_exp_f32 for the
compositing kernels).  CPU only: no GPU, no recompilation."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")

_INSN = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_FUNC = re.compile(r"^[0-9a-f]+ <(\S+)>:$")


def disassemble_library(lib=None):
    """{kernel symbol: [(address, mnemonic, operands)]} for every kernel of every gfx950 code object in the library."""
    from test_kernel_resources import LIB, code_objects

    blob = open(lib or LIB, "rb").read()
    out = {}
    for n, elf in enumerate(code_objects(blob)):
        path = f"/tmp/gs_isa_{os.getpid()}_{n}.elf"
        with open(path, "wb") as fh:
            fh.write(elf)
        try:
            txt = subprocess.run([OBJDUMP, "-d", path], check=True, capture_output=True, text=True).stdout
        finally:
            os.unlink(path)
        cur = None
        for line in txt.splitlines():
            m = _FUNC.match(line)
            if m:
                cur = out.setdefault(m.group(1), [])
                continue
            m = _INSN.match(line)
            if m and cur is not None:
                cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def loops(insns):
    """[(first index, last index)] of every backward branch's range [target, branch], innermost first."""
    addr_to_idx = {a: i for i, (a, _, _) in enumerate(insns)}
    out = []
    for i, (a, mn, ops) in enumerate(insns):
        if not (mn.startswith("s_cbranch") or mn == "s_branch"):
            continue
        try:
            simm = int(ops.split()[0], 0)
        except (ValueError, IndexError):
            continue
        if simm >= 0x8000:
            simm -= 0x10000
        tgt = a + 4 + 4 * simm
        if tgt <= a and tgt in addr_to_idx:
            out.append((addr_to_idx[tgt], i))
    out.sort(key=lambda r: r[1] - r[0])
    return out


def count(insns, lo, hi, prefix):
    return sum(1 for _, mn, _ in insns[lo:hi + 1] if mn.startswith(prefix))


def hot_loop(insns, marker, min_markers=1):
    """The innermost loop holding at least `min_markers` instructions whose mnemonic starts with `marker`."""
    for lo, hi in loops(insns):
        if count(insns, lo, hi, marker) >= min_markers:
            return lo, hi
    return None


def report(name, insns):
    print(f"{name[:110]}: {len(insns)} instructions, {count(insns, 0, len(insns) - 1, 'scratch_')} scratch")
    for lo, hi in loops(insns):
        n = hi - lo + 1
        if n < 24:
            continue
        print(f"   loop [{lo:6d}, {hi:6d}] {n:5d} insns: mfma {count(insns, lo, hi, 'v_mfma'):3d}  exp {count(insns, lo, hi, 'v_exp'):3d}"
              f"  dpp {sum(1 for _, _, o in insns[lo:hi + 1] if 'row_shr' in o or 'dpp' in o):3d}"
              f"  scratch {count(insns, lo, hi, 'scratch_'):3d}  global {count(insns, lo, hi, 'global_'):3d}"
              f"  ds {count(insns, lo, hi, 'ds_'):3d}")


if __name__ == "__main__":
    want = sys.argv[1:] or ["raster_backward_mfma_sh_kernel", "raster_forward_kernelILi27ELb1", "raster_forward_kernelILi48ELb1"]
    for name, insns in sorted(disassemble_library().items()):
        if any(w in name for w in want):
            report(name, insns)
