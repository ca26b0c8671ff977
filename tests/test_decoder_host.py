"""Host logic of crab_amd/decoder.py that needs no GPU: the one-retry wrapper around generate() / generate_many() (lazy eviction)."""
import gc
import weakref

import pytest
import torch

from crab_amd import decoder


class _Engine:
    """The four attributes _retry_after_eviction touches."""

    def __init__(self):
        self._kv, self._dec, self._ws, self.invalidated = {"slot": 1}, {}, {}, 0

    def invalidate(self):
        self.invalidated += 1
        self._kv = {}


class _Big:                       # stands for a tensor a failed attempt still holds in a local
    pass


def test_retry_runs_once_after_dropping_everything_and_outside_the_except_clause():
    eng, calls, held = _Engine(), [], []

    def attempt():
        calls.append(len(calls))
        if len(calls) == 1:
            big = _Big()
            held.append(weakref.ref(big))
            raise torch.cuda.OutOfMemoryError("HIP out of memory (simulated)")     # `big` lives in this frame, the frame in the traceback
        # second attempt: the first attempt's locals must be gone (a retry from INSIDE the except clause would still see them through the traceback)
        gc.collect()
        assert held[0]() is None
        return "ok"

    assert decoder.GenerationEngine._retry_after_eviction(eng, attempt) == "ok"
    assert calls == [0, 1] and eng.invalidated == 1


def test_retry_gives_up_when_nothing_is_cached_and_after_a_second_failure():
    eng = _Engine()
    eng._kv = {}

    def always():
        raise torch.cuda.OutOfMemoryError("HIP out of memory (simulated)")

    with pytest.raises(torch.cuda.OutOfMemoryError):
        decoder.GenerationEngine._retry_after_eviction(eng, always)                # nothing to drop: the caller's error
    assert eng.invalidated == 0
    eng._kv = {"slot": 1}
    with pytest.raises(torch.cuda.OutOfMemoryError):
        decoder.GenerationEngine._retry_after_eviction(eng, always)                # one retry, then the caller's
    assert eng.invalidated == 1
