"""Host logic of the multi-GPU sharding: homogeneous blocks and the mixed-N bucket split."""
import itertools

import numpy as np
import pytest

from qadapt_hip import shard


def test_homogeneous_blocks_cover_every_env_once():
    world, B = 8, 4096
    owned = np.zeros(world * B, int)
    for r in range(world):
        first, n = shard.shard_env_ids(r, world, B)
        owned[first:first + n] += 1
    assert np.all(owned == 1)
    with pytest.raises(ValueError):
        shard.shard_env_ids(8, 8, 4)


@pytest.mark.parametrize("counts", [{2: 8192, 4: 8192, 6: 8192, 8: 8192}, {2: 13, 4: 7, 6: 29, 8: 3}, {4: 5, 8: 1001}])
def test_mixed_buckets_cover_every_global_id_once_and_balance(counts):
    world = 8
    total = sum(counts.values())
    owned = np.zeros(total, int); bucket_of = np.zeros(total, int)
    off = 0
    for n in sorted(counts):
        bucket_of[off:off + counts[n]] = n
        off += counts[n]
    loads = []
    for r in range(world):
        a = shard.shard_mixed(counts, r, world)
        assert sorted(a) == sorted(counts)
        load = 0.0
        for n, (first, cnt) in a.items():
            owned[first:first + cnt] += 1
            assert np.all(bucket_of[first:first + cnt] == n)          # a slice never crosses its bucket
            assert abs(cnt - counts[n] / world) < 1                      # every bucket is split over ALL ranks
            load += cnt * shard.bucket_cost(n)
        loads.append(load)
    assert np.all(owned == 1)
    # cost-weighted remainders: no rank is more than one (most expensive) env above another
    assert max(loads) - min(loads) <= shard.bucket_cost(max(counts)) + 1e-9


def test_mixed_assignment_is_deterministic_and_rank_local():
    counts = {2: 10, 4: 11, 6: 12, 8: 13}
    a = [shard.shard_mixed(counts, r, 4) for r in range(4)]
    b = [shard.shard_mixed(dict(reversed(list(counts.items()))), r, 4) for r in range(4)]
    assert a == b
    for r, s in itertools.product(range(4), range(4)):
        if r < s:
            for n in counts:
                assert a[r][n][0] + a[r][n][1] <= a[s][n][0]
