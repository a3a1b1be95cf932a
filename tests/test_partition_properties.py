"""CPU: properties of the rank partitioning the multi-GPU path stands on (gtsfm_amd/parallel.py), over random scene sizes,
world sizes and pair subsets (hypothesis): every image and every pair has exactly one owner, a rank's 2-D share touches only its block rows and
columns, the feature-table index is a bijection onto the rows ``all_gather_feature_table`` fills, and the shares of BASELINE config 4 are balanced."""

import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from gtsfm_amd import parallel

worlds = st.sampled_from([1, 2, 3, 4, 6, 8, 12, 16])


@settings(max_examples=60, deadline=None)
@given(n=st.integers(0, 120), world=worlds)
def test_every_image_has_one_owner_and_one_table_row(n, world):
    owned = [parallel.partition_images(n, r, world) for r in range(world)]
    assert sorted(i for part in owned for i in part) == list(range(n))
    slots = -(-n // world) if n else 0
    assert all(len(part) <= slots for part in owned)
    rows = [parallel.table_index(i, n, world) for i in range(n)]
    assert len(set(rows)) == n and all(0 <= r < world * slots for r in rows)
    for r, part in enumerate(owned):  # rank-major, then the rank's own order: where all_gather_into_tensor puts rank r's s-th image
        assert [parallel.table_index(i, n, world) for i in part] == [r * slots + s for s in range(len(part))]


@settings(max_examples=60, deadline=None)
@given(n=st.integers(2, 60), world=worlds, block=st.integers(1, 5), keep=st.floats(0.05, 1.0), seed=st.integers(0, 2**31 - 1))
def test_every_pair_has_one_owner_in_both_partitionings(n, world, block, keep, seed):
    pairs = parallel.exhaustive_pairs(n)
    rng = np.random.default_rng(seed)
    pairs = [p for p in pairs if rng.random() < keep]  # a visibility graph is a subset of the exhaustive pairs
    flat = [parallel.partition_pairs(pairs, r, world) for r in range(world)]
    assert sorted(p for part in flat for p in part) == sorted(pairs)
    sizes = [len(part) for part in flat]
    assert max(sizes) - min(sizes) <= 1  # contiguous blocks: as even as integers allow
    grid = [parallel.partition_pairs_2d(pairs, r, world, block) for r in range(world)]
    assert sorted(p for part in grid for p in part) == sorted(pairs)
    rows, cols = parallel.process_grid(world)
    assert rows * cols == world and rows <= cols
    for r, part in enumerate(grid):
        gr, gc = divmod(r, cols)
        assert part == sorted(part)
        assert all((i // block) % rows == gr and (j // block) % cols == gc for i, j in part)
        touched = parallel.images_touched(part)
        assert set(touched) == {i for p in part for i in p} and touched == sorted(touched)


def test_config4_shares_are_balanced_and_touch_a_fraction_of_the_images():
    """BASELINE config 4: 101 views, the first 5000 exhaustive pairs, 8 ranks = 2 x 4 grid."""
    pairs = parallel.exhaustive_pairs(101)[:5000]
    shares = [parallel.partition_pairs_2d(pairs, r, 8) for r in range(8)]
    sizes = [len(s) for s in shares]
    assert sum(sizes) == 5000 and max(sizes) == 643 and max(sizes) <= 1.03 * (5000 / 8)
    touched = [len(parallel.images_touched(s)) for s in shares]
    assert max(touched) <= 101 // 2 + 101 // 4 + 2  # n / rows + n / cols images instead of all 101
