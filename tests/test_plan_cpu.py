"""The host rule that picks a call's decode loop and lays out the resident decoders' slot schedule (vqcpc_vocoder_plan: pure
host arithmetic of libvqcpc_hip.so, no GPU) -- ADVICE r3: a slot whose back-to-back schedule reaches 2^24 - 1 samples must fall
through to the launch-per-step kernels in auto mode instead of failing the call."""
import ctypes as C

import pytest

from vectorquantizedcpc_amd import _lib


def plan(samples, xcd=-1, xcm=-1, slots=0, xcd_slots=0, xcm_slots=0):
    lib = _lib.load()
    arr = (C.c_int * len(samples))(*samples)
    path, used, longest = C.c_int(), C.c_int(), C.c_int64()
    rc = lib.vqcpc_vocoder_plan(xcd, xcm, -1, -1, xcd_slots, xcm_slots, slots, arr, len(samples), C.byref(path), C.byref(used), C.byref(longest))
    if rc != 0:
        raise RuntimeError(lib.vqcpc_last_error().decode())
    return path.value, used.value, longest.value


def test_paths_by_utterances_in_flight():
    assert plan([32000] * 1) == (2, 1, 32000)
    assert plan([32000] * 32) == (2, 32, 32000)                    # BASELINE configs[3]'s per-GPU shard: the bench workload
    assert plan([32000] * 64) == (2, 32, 64000)                    # two utterances back to back in each of the 32 slots
    assert plan([32000] * 69) == (3, 69, 32000)
    assert plan([32000] * 256) == (3, 128, 64000)                  # configs[3] whole on one GPU
    assert plan([32000] * 600)[0] == 0                             # 512 and more in flight: launch-per-step kernels
    assert plan([32000] * 600, slots=256) == (3, 128, 160000)      # continuous batching: 256 < 512 in flight -> 128 resident slots
    assert plan([32000] * 8, xcd=0)[0] == 0
    assert plan([32000] * 8, xcm=1) == (3, 8, 32000)


def test_longest_first_onto_the_slot_that_frees_up_first():
    path, used, longest = plan([100, 900, 500, 400, 300], slots=2)
    assert (path, used) == (2, 2)
    assert longest == 1200                                         # {900, 300} and {500, 400, 100}: LPT
    assert plan([0, 0, 5], slots=0) == (2, 1, 5)                    # utterances that produce no samples take no slot


def test_a_schedule_of_2_pow_24_samples_falls_through_in_auto_mode_and_is_an_error_by_name():
    lim = (1 << 24) - 1
    assert plan([lim - 1]) == (2, 1, lim - 1)                      # the last step's tag is (lim - 1 + 1) << 8: still 32 bits
    assert plan([lim])[0] == 0                                     # one sample more: the launch path takes the call
    three = [6_400_000] * 3                                        # 3 x 400 s through ONE slot: 19.2 M samples back to back
    assert plan(three, slots=1)[0] == 0
    assert plan(three, slots=3) == (2, 3, 6_400_000)
    with pytest.raises(RuntimeError, match="does not fit the resident decoders"):
        plan(three, slots=1, xcd=1)
    with pytest.raises(RuntimeError, match="does not fit the resident decoders"):
        plan(three, slots=1, xcm=1)


def test_decode_chunks_respect_the_memory_budget_and_keep_every_utterance():
    """driver.decode_chunks: the reference's 9 474-utterance set (README.md:125) as ONE decode call would want ~35 GB of conditioning
    rows; the driver cuts the list into calls under a work-space budget."""
    import random
    from vectorquantizedcpc_amd import driver
    rnd = random.Random(13)
    n_codes = [rnd.randint(50, 500) for _ in range(9474)]                     # 1 .. 10 s
    for budget in (1 << 30, 8 << 30):
        chunks = driver.decode_chunks(n_codes, budget)
        assert [i for c in chunks for i in c] == list(range(9474))             # in order, nothing lost
        for c in chunks:
            own = sum(2 * n_codes[i] for i in c)
            t2 = max(2 * n_codes[i] for i in c)
            need = own * driver.BYTES_PER_OWN_FRAME + len(c) * t2 * (driver.BYTES_PER_PADDED_FRAME + 4 * 160)
            assert need <= budget or len(c) == 1
        assert len(chunks) > 1
    assert driver.decode_chunks([100, 100], 1) == [[0], [1]]                   # a budget below one utterance: one per call
    assert driver.decode_chunks([100] * 32, 8 << 30) == [list(range(32))]      # the bench workload: one call
