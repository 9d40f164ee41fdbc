"""CPU tier: register / scratch budget of the hot kernels, read from the gfx950 code objects inside the built
libgs_amd.so (no GPU, no recompilation).

The kernels' speed rests on a few occupancy facts that a harmless-looking edit can break without any test noticing --
the rgb compositing kernels must fit 128 VGPRs (four waves per SIMD; the checkpointing variant once had 130), none of
the frame path's hot kernels may spill to scratch, the per-tile sort must fit four workgroups per CU.  The numbers come
from the AMDGPU metadata note of every code object in the library's `.hip_fatbin` (clang offload bundles)."""
import os
import struct

import msgpack
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "3d-gaussian-splatting_amd", "csrc", "libgs_amd.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    """Every gfx950 ELF inside the clang offload bundles of the library."""
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        cur = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, cur)
            triple = blob[cur + 24:cur + 24 + tlen].decode()
            cur += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def kernel_metadata(elf):
    """{kernel symbol: metadata dict} from the NT_AMDGPU_METADATA note (type 32, name "AMDGPU") of an ELF64 code object."""
    assert elf[:4] == b"\x7fELF" and elf[4] == 2
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    out = {}
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        if sh_type != 7:  # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            name = elf[p + 12:p + 12 + namesz].rstrip(b"\0")
            dpos = p + 12 + (namesz + 3) // 4 * 4
            if name == b"AMDGPU" and ntype == 32:
                md = msgpack.unpackb(elf[dpos:dpos + descsz], raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    out[k[".name"]] = k
            p = dpos + (descsz + 3) // 4 * 4
    return out


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(LIB):
        pytest.skip("libgs_amd.so is not built")
    blob = open(LIB, "rb").read()
    out = {}
    for elf in code_objects(blob):
        out.update(kernel_metadata(elf))
    assert len(out) > 60, f"only {len(out)} kernels found in the library's code objects"
    return out


def pick(kernels, *parts):
    hits = [v for k, v in kernels.items() if all(p in k for p in parts)]
    assert len(hits) == 1, (parts, [h[".name"] for h in hits])
    return hits[0]


# (substrings of the mangled name, VGPR limit, scratch allowed?)  <C, FRAME, CKPT, SIG, WN, EXACT> / <DIST> / ...
BUDGETS = [
    (("raster_forward_kernelILi3ELb1ELb0ELb0ELb0ELb0E",), 128, False),   # rgb inference frame: four waves per SIMD
    (("raster_forward_kernelILi3ELb1ELb1ELb0ELb0ELb0E",), 128, False),   # rgb training frame (checkpoints)
    (("raster_forward_kernelILi27ELb1ELb0ELb0ELb0ELb0E",), 168, True),   # SH degree 2: three waves (a few spilled tile constants)
    (("raster_forward_kernelILi48ELb1ELb0ELb0ELb0ELb0E",), 256, True),   # SH degree 3: two waves
    (("raster_backward_pixel_sh_kernelILi3ELb1ELb0E",), 96, False),      # rgb backward: five waves (LDS-limited at 7.6 KiB)
    (("raster_backward_rows_kernel",), 128, False),                      # rgb backward, row layout (round 5): four waves
    (("raster_backward_pixel_sh_kernelILi27ELb1ELb0E",), 128, True),
    (("raster_backward_pixel_sh_kernelILi48ELb1ELb0E",), 168, True),
    # SH backward on the matrix pipe (round 4): three waves per SIMD; the spills the allocator leaves at that budget sit in
    # the per-tile / per-group set-up, not in the pixel-row loop (checked in the ISA when the kernel was written)
    (("raster_backward_mfma_sh_kernelILi27ELi2E",), 168, True),          # two waves per workgroup: 1,024 tiles and more
    (("raster_backward_mfma_sh_kernelILi48ELi2E",), 168, True),
    (("raster_backward_mfma_sh_kernelILi27ELi4E",), 168, True),          # four: small tile grids
    (("raster_backward_mfma_sh_kernelILi48ELi4E",), 168, True),
    (("frame_project_backward_kernelILi3ELi0ELi256ELi0E",), 80, False),  # rgb projection backward: six waves per SIMD
    (("frame_project_backward_kernelILi3ELi0ELi256ELi1E",), 104, False), # ... with the Adam step fused in (round 5): four
    (("frame_project_count_kernelILb0E",), 128, False),                   # 1024 threads per workgroup: 128 is the hard limit
    (("frame_project_cull_count_kernel",), 128, False),                   # ... of an occlusion-culled frame (round 6): compacting
    (("frame_project_bin_count_kernelILb0E",), 128, False),
    (("strip_sort_kernelILi2048ELb0E",), 128, False),                    # four workgroups of 256 per CU
    (("loss_fused_kernel",), 256, False),                                # one workgroup of 512 per CU: two waves per SIMD
    (("adam_kernelILb0E",), 64, False),
    (("adam_kernelILb1E",), 64, False),
]


@pytest.mark.parametrize("parts,vgprs,scratch_ok", BUDGETS, ids=[b[0][0] for b in BUDGETS])
def test_hot_kernel_fits_its_register_budget(kernels, parts, vgprs, scratch_ok):
    k = pick(kernels, *parts)
    assert k[".vgpr_count"] <= vgprs, (k[".name"], k[".vgpr_count"])
    if not scratch_ok:
        assert k[".vgpr_spill_count"] == 0 and k[".private_segment_fixed_size"] == 0, \
            (k[".name"], k[".vgpr_spill_count"], k[".private_segment_fixed_size"])


def test_lds_budgets(kernels):
    """Static LDS (dynamic LDS is added at launch): the compositing kernels must leave room for their resident waves."""
    fwd = pick(kernels, "raster_forward_kernelILi3ELb1ELb0ELb0ELb0ELb0E")
    assert fwd[".group_segment_fixed_size"] * 16 <= 160 * 1024      # 16 waves (= one-wave workgroups) per CU
    bwd = pick(kernels, "raster_backward_pixel_sh_kernelILi3ELb1ELb0E")
    assert bwd[".group_segment_fixed_size"] * 20 <= 160 * 1024      # five waves per SIMD
    # the rgb backward stages its 64 rows in LDS for the one-line stores: it must stay at 20 one-wave workgroups per CU
    assert bwd[".group_segment_fixed_size"] <= 8192
    pb = pick(kernels, "frame_project_backward_kernelILi3ELi0ELi256ELi0E")
    assert pb[".group_segment_fixed_size"] * 6 <= 160 * 1024        # six workgroups of 256 per CU
    for c in ("27", "48"):  # SH backward on the matrix pipe: twelve waves per CU (the register limit) in workgroups of 2 / 4
        assert pick(kernels, f"raster_backward_mfma_sh_kernelILi{c}ELi2E")[".group_segment_fixed_size"] * 6 <= 160 * 1024
        assert pick(kernels, f"raster_backward_mfma_sh_kernelILi{c}ELi4E")[".group_segment_fixed_size"] * 3 <= 160 * 1024
    # SH projection backward: its row walk is bound by load latency -- ten workgroups of two waves per CU at degree 3
    # (the walk runs in two passes over 32 owners each: half the column sums in LDS; one pass left five workgroups)
    assert pick(kernels, "frame_project_backward_kernelILi48ELi0ELi128ELi0E")[".group_segment_fixed_size"] * 10 <= 160 * 1024
    assert pick(kernels, "frame_project_backward_kernelILi27ELi0ELi128ELi0E")[".group_segment_fixed_size"] * 15 <= 160 * 1024
    sort = pick(kernels, "strip_sort_kernelILi2048ELb0E")
    assert sort[".group_segment_fixed_size"] * 4 <= 160 * 1024      # four workgroups per CU
    assert pick(kernels, "frame_project_count_kernelILb0E")[".group_segment_fixed_size"] + 8192 * 8 <= 160 * 1024


# ---------------------------------------------------------------------------------------------------------------------
# "The spills sit in the set-up, not in the hot loop" as an assertion.  The SH compositing
# kernels and the SH backward on the matrix pipe are allocated 168 / 255 VGPRs and the allocator leaves a few spilled
# registers (scratch_load / scratch_store); what matters is WHERE: a spill in the per-Gaussian / per-pixel-row loop is paid
# hundreds of times per tile, one in the per-tile or per-group set-up once.  tools/isa_loops.py disassembles the code
# objects of the built library, finds the loops (backward branches) and the innermost one that holds the kernel's
# characteristic instructions; that loop must be free of scratch instructions.
HOT_LOOPS = [
    # (kernel name substrings, marker mnemonic prefix, markers the hot loop holds at least)
    (("raster_forward_kernelILi27ELb1ELb0ELb0ELb0ELb0E",), "v_exp_f32", 32),   # SH degree 2 compositing (inference frame)
    (("raster_forward_kernelILi27ELb1ELb1ELb0ELb0ELb0E",), "v_exp_f32", 32),   # ... training frame (checkpoints)
    (("raster_forward_kernelILi48ELb1ELb0ELb0ELb0ELb0E",), "v_exp_f32", 32),   # SH degree 3
    (("raster_forward_kernelILi48ELb1ELb1ELb0ELb0ELb0E",), "v_exp_f32", 32),
    (("raster_backward_mfma_sh_kernelILi27ELi2E",), "v_mfma_f32_16x16x4", 12),  # the pixel-row loop: 16 Gaussians x 16 pixels
    (("raster_backward_mfma_sh_kernelILi27ELi4E",), "v_mfma_f32_16x16x4", 12),
    (("raster_backward_mfma_sh_kernelILi48ELi2E",), "v_mfma_f32_16x16x4", 12),
    (("raster_backward_mfma_sh_kernelILi48ELi4E",), "v_mfma_f32_16x16x4", 12),
    (("raster_forward_kernelILi3ELb1ELb0ELb0ELb0ELb0E",), "v_exp_f32", 8),     # rgb compositing: no scratch anywhere anyway
    (("raster_backward_pixel_sh_kernelILi3ELb1ELb0E",), "v_exp_f32", 4),
    (("raster_backward_rows_kernel",), "v_exp_f32", 4),                        # rgb backward, row layout: the pixel-row step
]


@pytest.fixture(scope="module")
def disassembly():
    import shutil
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_loops

    if not os.path.exists(LIB):
        pytest.skip("libgs_amd.so is not built")
    if not (os.path.exists(isa_loops.OBJDUMP) or shutil.which(isa_loops.OBJDUMP)):
        pytest.skip("llvm-objdump not found")
    return isa_loops, isa_loops.disassemble_library()


@pytest.mark.parametrize("parts,marker,at_least", HOT_LOOPS, ids=[h[0][0] for h in HOT_LOOPS])
def test_hot_loop_is_free_of_scratch_instructions(disassembly, parts, marker, at_least):
    isa, dis = disassembly
    hits = [k for k in dis if all(p in k for p in parts)]
    assert len(hits) == 1, (parts, hits)
    insns = dis[hits[0]]
    # EVERY innermost loop with the marker count (the compositing kernels hold two copies of their loop: with and without
    # the NaN-exact multiply) -- innermost = no other qualifying loop nested inside it
    cands = [(lo, hi) for lo, hi in isa.loops(insns) if isa.count(insns, lo, hi, marker) >= at_least]
    assert cands, (hits[0], "no loop with", at_least, marker)
    inner = [(lo, hi) for lo, hi in cands if not any((l2, h2) != (lo, hi) and lo <= l2 and h2 <= hi for l2, h2 in cands)]
    for lo, hi in inner:
        n_scratch = isa.count(insns, lo, hi, "scratch_")
        assert n_scratch == 0, (hits[0], f"hot loop [{lo}, {hi}] of {hi - lo + 1} instructions holds {n_scratch} scratch "
                                         f"instructions: a spill inside the per-row / per-Gaussian loop")
        assert hi - lo + 1 >= 100  # (a real loop body, not a two-instruction wait loop)
