"""Tensor-level wrappers and autograd Functions over libvnpcc (include/vnpcc.h).

Everything here works on the *rows layout*: a VN feature map with P points and C channels is a
2-D fp32 tensor ``[P*3, C]`` (row = point*3 + t, channels contiguous, ``stride(1) == 1``).
torch supplies device memory, the current HIP stream and the autograd graph; all arithmetic
is done by the HIP kernels.  No function in this package has a CPU / PyTorch fallback.

Modules: ``core`` (GEMM wrapper, reductions, BatchNorm / leaky passes, layout Functions), ``linear``, ``dirlayer``, ``pool``, ``fold``,
``chamfer``, ``graph``, ``attention``; ``_state`` holds the switches (``SW``) and the test / bench hooks (``hooks(...)``).
"""
from ._state import HK, POOL_PATH_COUNTS, SW, hooks, reserve_timers, switches, trace
from .attention import (AddLayerNormFn, VNAttentionFn, VNAttentionPackedFn, VNLayerNormFn, _attn_dims, _attn_token_major, _from_heads, _heads_fm,
                        _heads_tm, _rows_from_fm, _to_heads, gemm_batched)
from .chamfer import ChamferFn, ChamferL1LossFn
from .core import (BN_EPS, BN_MOMENTUM, FromRowsFn, SplitColsFn, ToRowsFn, WeightProductFn, _bn_leaky_backward, _bn_leaky_forward, _dgrad,
                   _ld, _rows, amax_word, cat_rows, drop_amax, gemm, pack_hint, pack_pd_cols, PackPdColsFn, PackRowsFn,
                   gemm_pd_stats, reduce_pairs, reduce_rows, set_amax, split_cols)
from .dirlayer import DirContractFn, DirExpandFn, DirLinearMaxPoolFn, VNDirLinearLeakyFn, unit_dirs
from .fold import FoldBnLeakyFn, FoldFn, GroupFoldBnLeakyFn, LinOutFn, group_fold_ok
from .graph import (EdgeConvFn, EdgeConvPackedFn, EdgeConvRank1Fn, EdgeGatherAddFn, EdgeWeightsFn, FpsPlan, GatherPointsFn, edge_reverse, edge_reverse_ahead,
                    fps_indices,
                    knn_indices)
from .linear import VNBnLeakyFn, VNLeakyDirFn, VNLeakyFn, VNLinearFn, VNLinearLeakyFn, VNLinearLeakyOutFn
from .pool import _screen_state, VNLinearMaxPoolFn, VNLinearPoolTapFn, VNMaxPoolFn, VNMaxPoolTapFn, VNMeanPoolFn
