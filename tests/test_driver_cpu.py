"""-m "not gpu": host-side scheduling rules of the batched drivers (no kernel is called)."""
from vectorquantizedcpc_amd import driver


def test_bucketing_rules():
    lengths = [40, 41, 300, 44, 256, 258, 36]
    modes = [driver.batch1_conv_mode(80, t) for t in lengths]
    assert modes == [1, 1, 2, 1, 1, 2, 1]
    buckets = driver.make_buckets(lengths, modes, max_batch=3, max_pad_frac=0.25)
    assert sorted(i for b in buckets for i in b) == list(range(7))
    for b in buckets:
        assert len(b) <= 3 and len({modes[i] for i in b}) == 1
        hi = max(lengths[i] for i in b)
        assert hi * len(b) - sum(lengths[i] for i in b) <= 0.25 * hi * len(b)
