"""CPU tier: the contract between the SH big-row pre-pass (cull_project.hip: sh_big_rows_kernel) and the projection backward's row
walk, as a NumPy model.

A Gaussian's per-pair gradient rows lie in one contiguous region (emission order); which rows the raster backward wrote
is a predicate of the row (`flags` below: until round 4 a flag byte per row, since round 5 "the Gaussian's key <= the stop
key of the row's tile", evaluated by both kernels with the same expression).  The projection backward sums a Gaussian's existing rows.  For a Gaussian beyond GS_PB_SH_BIG rows the
pre-pass -- sixteen waves, each adding a sixteenth of the region's existing rows in ascending order, a fixed pairwise tree over
the sixteen partial sums -- stores the TOTAL in the region's first row; the walk then presents such a Gaussian with exactly that
row, whether or not that row's own pair exists, in every part and slice.  The model checks that (1) the walk's result equals the plain sum of
the existing rows for small and big Gaussians alike, (2) it does so when the first row did not exist, when no row existed,
and when the region is cut short by the workspace capacity, (3) running the walk twice (geometry part, colour part) reads the
same totals."""
import numpy as np

BIG = 64      # GS_PB_SH_BIG
WAVES = 16
RW = 64       # floats per row at degree 3 (whole 64-byte lines since round 5)


def prepass(rows, flags, offsets, counts, max_pairs):
    rows = rows.copy()
    for off, cnt in zip(offsets, counts):
        if cnt <= BIG or off >= max_pairs:
            continue
        cnt = min(cnt, max_pairs - off)
        chunk = -(-cnt // WAVES)
        part = np.zeros((WAVES, RW), np.float32)
        for w in range(WAVES):
            for k in range(w * chunk, min(cnt, (w + 1) * chunk)):  # ascending, existing rows only
                if flags[off + k]:
                    part[w] += rows[off + k]
        st = 1
        while st < WAVES:  # the fixed pairwise tree
            for w in range(0, WAVES, 2 * st):
                part[w] += part[w + st]
            st *= 2
        rows[off] = part[0]
    return rows


def walk(rows, flags, offsets, counts, max_pairs):
    out = np.zeros((len(offsets), RW), np.float32)
    for g, (off, cnt) in enumerate(zip(offsets, counts)):
        nrow = 0 if off >= max_pairs else min(cnt, max_pairs - off)
        if cnt > BIG:
            if nrow:
                out[g] = rows[off]  # its first row holds the total
            continue
        for k in range(nrow):
            if flags[off + k]:
                out[g] += rows[off + k]
    return out


def test_walk_after_prepass_equals_the_sum_of_the_existing_rows():
    rng = np.random.default_rng(3)
    counts = np.array([3, 0, 64, 65, 700, 1, 300, 5000, 12, 90])
    offsets = np.concatenate([[0], np.cumsum(counts)[:-1]])
    total = int(counts.sum())
    max_pairs = total - 40  # the last Gaussian's region is cut short by the capacity
    rows = rng.normal(size=(total, RW)).astype(np.float32)
    flags = (rng.random(total) < 0.3).astype(np.uint8)
    flags[offsets[3]] = 0                                      # a big Gaussian whose first row does not exist
    flags[offsets[6]:offsets[6] + counts[6]] = 0               # ... and one with no existing row at all
    want = np.zeros((len(counts), RW), np.float64)
    for g, (off, cnt) in enumerate(zip(offsets, counts)):
        for k in range(min(cnt, max(0, max_pairs - off))):
            if flags[off + k]:
                want[g] += rows[off + k]
    after = prepass(rows, flags, offsets, counts, max_pairs)
    got = walk(after, flags, offsets, counts, max_pairs)
    scale = np.abs(rows).max() * np.maximum(counts, 1)[:, None]
    assert np.all(np.abs(got - want) <= 1e-6 * scale)
    assert np.all(got[6] == 0) and np.all(got[1] == 0)
    # rows of small Gaussians are untouched; of a big one only the first
    small = counts <= BIG
    for g in np.flatnonzero(small):
        assert np.array_equal(after[offsets[g]:offsets[g] + counts[g]], rows[offsets[g]:offsets[g] + counts[g]])
    for g in np.flatnonzero(~small):
        assert np.array_equal(after[offsets[g] + 1:offsets[g] + counts[g]], rows[offsets[g] + 1:offsets[g] + counts[g]])
    # the geometry part and the colour part walk the same rows: same totals
    assert np.array_equal(walk(after, flags, offsets, counts, max_pairs), got)
