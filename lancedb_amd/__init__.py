"""lancedb_amd — MI355X-native ANN scan engine for LanceDB's vector-search path.

Product code only: the C-ABI library (csrc/ -> libmi355_ann.so) and the
host-side mirror of the reference's query interface.  Nothing here imports
oracle/ (test infrastructure) and nothing falls back to the CPU.
"""
from . import _abi  # noqa: F401
from ._lib import (EngineError, InvalidInput, NotSupported, QueryTimeout, build, device_count,  # noqa: F401
                   lib)
from ._hip import DeviceArray, HostAllocArray, HostMappedArray, synchronize  # noqa: F401
from .index import (FlatIndex, IvfPqIndex, SearchResult, ivf_residuals, ivfpq_encode, kmeans_train,  # noqa: F401
                    merge_topk, pq_train, shard_plan)
from .build import IvfPqBuilder, suggested_num_partitions, suggested_num_sub_vectors  # noqa: F401
from .query import DEFAULT_TOP_K, VectorQuery, VectorQueryRequest, VectorTable  # noqa: F401

__version__ = "0.1.0"
