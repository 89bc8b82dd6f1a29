"""dynaalign_amd -- MI355X-native all-pairs similarity hot path of DynaAlign.

Drop-in for the reference's ``similarityMH()`` / ``similarityNW()`` (reference
R/RcppExports.R:15-17, :34-36).  The compute lives in hand-written HIP kernels
for gfx950 behind a C ABI (include/dynaalign.h, libdynaalign_hip.so); this
package is the thin host side: argument marshalling (`similarity`), torch-tensor
plumbing for device-resident use (`device`), row-sharding over GPUs
(`sharding`) and synthetic inputs (`synth`).  There is no CPU implementation of
the hot path in the package: without the HIP library and a GPU every compute
call raises.
"""
from ._capi import DynaAlignError, load as load_library  # noqa: F401
from .clusterbreak import clusterbreak, louvain, louvain_csr, netcluster  # noqa: F401
from .similarity import (  # noqa: F401
    SimilarityMatrix, get_option, hash_family_seeds, mh_counts, minhash_signatures, nw_pairs,
    pack_sequences, quantile_type7, set_option, similarityMH, similarityMH_edges, similarityNW, similarityNW_edges,
)

__all__ = [
    "clusterbreak", "netcluster", "louvain", "louvain_csr",
    "similarityMH", "similarityNW", "similarityMH_edges", "similarityNW_edges", "quantile_type7", "minhash_signatures", "mh_counts", "nw_pairs", "hash_family_seeds",
    "pack_sequences", "set_option", "get_option", "SimilarityMatrix", "DynaAlignError", "load_library",
]
