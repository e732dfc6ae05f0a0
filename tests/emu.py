"""Test-only: the product's HIP sources built against the fiber emulator (tests/hipemu)."""
import os, sys
import functools

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'hipemu'))


@functools.lru_cache(None)
def emu_lib():
    import build_emu
    from segtran_amd.segx import SegxLib
    return SegxLib(build_emu.build())
