"""psac_amd -- MI355X-native suffix array / inverse SA / LCP construction.

Host-side mirror of psac's `suffix_array<char_t, index_t, LCP>` class
(/root/reference/include/suffix_array.hpp:170-228, :469-486) over the C-ABI of
libpsacx.so (include/psacx.h).  The compute path is HIP only.
"""
from .suffix_array import Context, SuffixArray, parse_stringset, ansv, ansv_device, check_device, suffix_tree, NEAREST_SM, NEAREST_EQ, FURTHEST_EQ  # noqa: F401
from ._lib import PsacxError, LIB_PATH  # noqa: F401
from .multi import MultiContext, unique_id  # noqa: F401

__all__ = ["Context", "SuffixArray", "parse_stringset", "suffix_tree", "check_device", "ansv", "PsacxError", "MultiContext", "unique_id", "NEAREST_SM", "NEAREST_EQ", "FURTHEST_EQ"]
