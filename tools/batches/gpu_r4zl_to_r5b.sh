#!/bin/bash
# round 4, calls zl ... r5b (issued inline, collected here): what each of them ran.  Variant libraries under build/variants/
# come from `python tools/ab_variants.py build name="-DFLAG=..."` (the -D switches named per call).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; OUT=$R/gpurun_out/r4zl_r5b; mkdir -p "$OUT"
V=build/variants
# zl / zm: cfg2 + a 100,000-Gaussian pile with SH, in-tree against old = -DGS_BWD_SH_MFMA=0 (before / after the long-list hand-over)
for lib in "" $V/old/libgs_amd.so; do for d in 2 3; do GS_AMD_LIB=$lib timeout 300 python tools/long_list.py 100000 $d | tail -1; done; done > "$OUT/pile.txt" 2>&1
# zn / zo / zp / zq / zr / zs / zw: kernel traces of the densifying soaks (rgb: degree 0; SH: 2, 3); zs with big128 / big64 = -DGS_PB_BIG=128 / 64
(cd /tmp && export TMPDIR=/tmp && for d in 0 2 3; do timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/soak_prof_$d" -o s -- python "$R/tools/soak.py" 0 1500 $d > "$OUT/soak_$d.json" 2> /dev/null; done)
# zf: dead-pixel statistics (diag = -DGS_DIAG_CKPT)
[ -f $V/diag/libgs_amd.so ] && GS_AMD_LIB=$V/diag/libgs_amd.so timeout 300 python tools/dead_pixel_stats.py cfg5 > "$OUT/dead_pixels.txt" 2>&1
# zi / zj: SH projection backward passes (p1 / p4 = -DGS_PB_SH_PASSES=1 / 4) and loads in flight (u4 / u12 = -DGS_PB_SH_U=4 / 12)
# zy / zz / r5a: waves per workgroup of the MFMA SH backward (t2 = -DGS_BWD_MFMA_TILES=2, w2 = -DGS_BWD_MFMA_WAVES=2,
#   w4s1 = -DGS_BWD_MFMA_WAVES=4 -DGS_BWD_MFMA_SPLIT=1, w2s1 = -DGS_BWD_MFMA_SPLIT=1 on a tree whose default was SPLIT=2)
names=$(ls $V 2>/dev/null | tr '\n' ',' | sed 's/,$//')
[ -n "$names" ] && timeout 900 python tools/mfma_bwd_check.py compare "$names" c2 c2_deg3 cfg4 cfg4_deg3 d2 d3 > "$OUT/compare.txt" 2> "$OUT/compare.err"
# zt: static scenes with GS_PB_BIG = 64 / 256 / 32: python tools/ab_variants.py run cfg5 cfg2 cfg3
# zx: tiny scenes, fused column scan (nofuse = -DGS_BIN_FUSED_N=0 on the experimental tree): python tools/sweep_n.py 10000 16000 100000
# zu / r5b: python -m pytest tests -m gpu -x -q; python -c "import __graft_entry__ as g; g.smoke()"; python bench.py
cat "$OUT/pile.txt" | cut -c1-300
