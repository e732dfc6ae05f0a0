"""Test-only: the product's HIP sources built against the fiber emulator (tests/hipemu)."""
import os, sys
import functools

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'hipemu'))


@functools.lru_cache(None)
def emu_lib():
    from segtran_amd.segx import SegxLib
    override = os.environ.get('SEGX_EMU_LIB')              # e.g. an AddressSanitizer build of the same sources (run with the ASan runtime preloaded)
    if override:
        return SegxLib(override)
    import build_emu
    return SegxLib(build_emu.build())
