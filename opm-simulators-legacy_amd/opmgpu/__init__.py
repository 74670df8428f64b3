"""opmgpu -- Python harness over libopmgpu.so (the MI355X black-oil Newton step).

The compute lives in ../csrc (hand-written HIP for gfx950) behind the C ABI of include/opmgpu.h;
this package only binds it (capi), mirrors the reference's model/solver interfaces (model),
synthesises decks (decks) and partitions grids for multi-GPU runs (partition).
"""
from . import capi, decks, model  # noqa: F401
