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


def test_fit_slots_covers_total_over_longest_in_whole_tiles():
    # 512 utterances of 1-10 s keep only total / longest slots busy for the schedule's whole length
    lengths = [16000 * (1 + i % 10) for i in range(512)]          # total 45 056 000, longest 160 000 -> 282 -> 288
    assert driver.fit_slots(lengths, 512) == 288
    assert driver.fit_slots(lengths, 256) == 256                    # never more than asked for
    assert driver.fit_slots([32000] * 256, 256) == 256              # equal lengths: one utterance per slot
    assert driver.fit_slots([32000] * 40, 256) == 48                # 40 utterances -> 3 tiles
    assert driver.fit_slots([160000] + [16000] * 9, 256) == 16      # one long utterance dominates: a single tile
    assert driver.fit_slots([5, 7], 8) == 8                         # below one tile: as asked
    assert driver.fit_slots([], 64) == 64
