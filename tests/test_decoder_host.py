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


def test_workspace_queries_are_monotone_so_that_one_buffer_serves_every_smaller_call():
    """r06: callers size ONE workspace for their largest chunk (decoder._Workspace.t for the prefill chunk) and reuse it for the smaller last chunk.
    crab_hyperlora_route_workspace used to return slices(M) * M, which is not monotone (M = 57 344: 5 slices, M = 50 000: 6 -> 5 % more), and
    generate_avs_many with 512 samples failed with "hyperlora_route: workspace too small".  Every query is now non-decreasing in every argument."""
    from crab_amd import _lib
    lib = _lib.load()
    for K in (768, 4096, 11008, 18944):
        prev = 0
        for M in list(range(1, 5000, 7)) + list(range(5000, 400000, 997)):
            w = lib.crab_hyperlora_route_workspace(M, K, 48)
            assert w >= prev, (M, K, w, prev)
            prev = w
        assert lib.crab_hyperlora_route_workspace(50000, K, 48) <= lib.crab_hyperlora_route_workspace(57344, K, 48)
    assert lib.crab_hyperlora_route_workspace(4096, 4096, 48) <= lib.crab_hyperlora_route_workspace(4096, 11008, 48)
    for N in (1, 64, 1000, 16384):
        prev = 0
        for M in range(1, 40000, 13):
            w = lib.crab_vq_nearest_f32_workspace(M, N)
            assert w >= prev, (M, N, w, prev)
            prev = w
    for q, args in ((lib.crab_rowfin_workspace, [(m, 4096) for m in range(1, 17)]), (lib.crab_attn_decode_rope_workspace, [(b, 32, 128) for b in range(1, 64)]),
                    (lib.crab_groupnorm_workspace, [(2, hw, 32) for hw in range(1, 70000, 101)])):
        vals = [q(*a) for a in args]
        assert vals == sorted(vals)
