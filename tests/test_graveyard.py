"""
Host logic of the device-block cache (``lsqfitgp_amd/_device.py``), run against a stand-in for the library: the order in which
the garbage collector finalises a context and its buffers is arbitrary when they die together in a reference cycle, and a block
must never be given back through a context that no longer exists (a segmentation fault in ``lgp_free`` in the round-3 GPU suite).
"""

from lsqfitgp_amd._device import _Graveyard, _destroy_context
from lsqfitgp_amd import _lib


class FakeLib:
    def __init__(self):
        self.calls = []
        self.alive = True

    def lgp_free(self, handle, ptr):
        assert self.alive, 'lgp_free through a destroyed context'
        self.calls.append(('free', ptr.value))
        return 0

    def lgp_ctx_destroy(self, handle):
        assert self.alive
        self.alive = False
        self.calls.append(('destroy',))
        return 0


def adopt(g, n=1):
    with g._lock:
        g.live += n


def test_context_finalised_before_its_buffers():
    lib = FakeLib()
    g = _Graveyard(lib, 'handle')
    _lib._on_enomem.append(g.flush)
    adopt(g, 3)
    g.bury(101, 64)                                  # cached for reuse
    _destroy_context(lib, 'handle', g)               # the context object dies while two blocks are still out
    assert lib.alive and g.closing
    g.bury(102, 3 << 30)                             # a block above every cache limit: queued, not freed through a dead context
    assert lib.alive
    g.bury(103, 64)                                  # the last one: everything is given back, then the context goes
    assert not lib.alive
    assert lib.calls[-1] == ('destroy',) and sorted(c[1] for c in lib.calls[:-1]) == [101, 102, 103]
    assert g.flush not in _lib._on_enomem


def test_context_finalised_after_its_buffers():
    lib = FakeLib()
    g = _Graveyard(lib, 'handle')
    _lib._on_enomem.append(g.flush)
    adopt(g, 2)
    g.bury(7, 128)
    g.bury(8, 128)
    again = g.take(128)                              # reuse by size
    assert again in (7, 8)
    adopt(g)
    g.bury(again, 128)
    _destroy_context(lib, 'handle', g)
    assert not lib.alive and lib.calls[-1] == ('destroy',) and sorted(c[1] for c in lib.calls[:-1]) == [7, 8]
