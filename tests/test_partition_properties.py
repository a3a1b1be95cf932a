"""Property tests (hypothesis) of the multi-GPU partitioning in ``gtsfm_amd/parallel.py``: for ANY pair list, world size and
block size every pair is owned by exactly one rank, 2-D ownership is balanced, and a rank of an R x C process grid touches the
images of one grid row and one grid column only (~2n / sqrt(R) images per rank instead of n)."""

import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from gtsfm_amd import parallel


@settings(max_examples=60, deadline=None)
@given(n=st.integers(2, 40), world=st.integers(1, 16), block=st.integers(1, 4), drop=st.integers(0, 5), seed=st.integers(0, 10_000))
def test_every_pair_has_exactly_one_owner(n, world, block, drop, seed):
    pairs = parallel.exhaustive_pairs(n)
    rng = np.random.default_rng(seed)
    if drop and len(pairs) > drop:  # retrieval-style lists: not every edge is present
        keep = sorted(rng.choice(len(pairs), len(pairs) - drop, replace=False).tolist())
        pairs = [pairs[k] for k in keep]
    owned = [parallel.partition_pairs_2d(pairs, r, world, block) for r in range(world)]
    assert sorted(sum(owned, [])) == sorted(pairs)
    assert sum(len(o) for o in owned) == len(pairs)
    blocks = [parallel.partition_pairs(pairs, r, world) for r in range(world)]
    assert sorted(sum(blocks, [])) == sorted(pairs)
    images = [parallel.partition_images(n, r, world) for r in range(world)]
    assert sorted(sum(images, [])) == list(range(n))
    for r in range(world):  # the feature table a rank gathers is indexed consistently
        for i in images[r]:
            assert 0 <= parallel.table_index(i, n, world) < -(-n // world) * world


@settings(max_examples=40, deadline=None)
@given(n=st.integers(8, 64), world=st.sampled_from([1, 2, 4, 8, 16]))
def test_two_d_ownership_is_balanced_and_local(n, world):
    pairs = parallel.exhaustive_pairs(n)
    rows, cols = parallel.process_grid(world)
    assert rows * cols == world
    sizes, touched = [], []
    for r in range(world):
        mine = parallel.partition_pairs_2d(pairs, r, world)
        sizes.append(len(mine))
        touched.append(len(parallel.images_touched(mine)))
    # cyclic ownership: no rank holds more than its share plus the boundary of the triangle
    assert max(sizes) - min(sizes) <= n
    if world >= 4 and n >= 4 * world:
        assert max(touched) <= -(-n // rows) + -(-n // cols)  # one grid row + one grid column of images
        if 1.0 / rows + 1.0 / cols < 0.99:  # from 8 ranks (2 x 4: three quarters of the images) on: fewer than all of them
            assert max(touched) < n
