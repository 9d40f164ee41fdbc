#!/bin/bash
# round 4, call y: SH backward on the matrix pipe (raster_backward_mfma_sh_kernel) against the pixel-parallel kernel:
# gradients and stage times on small scenes (three builds of the new kernel), then the 2.4 M scene, then the SH tests of
# the GPU suite on the variant library
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r4y; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 600 python tools/mfma_bwd_check.py compare mfma,mfmab,mfma2 s2 s3 d2 d3 > "$OUT/small.txt" 2> "$OUT/small.err"; echo "small rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/small.txt" | cut -c1-230
timeout 900 python tools/mfma_bwd_check.py compare mfma,mfma2 cfg4 cfg4_deg3 > "$OUT/full.txt" 2> "$OUT/full.err"; echo "full rc=$?" | tee -a "$OUT/steps.txt"
cat "$OUT/full.txt" | cut -c1-230
GS_AMD_LIB=$R/build/variants/mfma/libgs_amd.so timeout 900 python -m pytest tests -m gpu -x -q -k "sh or deg3 or backward_parity or long_lists or block_boundaries or degenerate" > "$OUT/pytest.txt" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/steps.txt"
tail -15 "$OUT/pytest.txt"
