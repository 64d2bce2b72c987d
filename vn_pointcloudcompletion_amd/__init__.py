"""vn_pointcloudcompletion_amd — MI355X-native hot path of ChenBarryHu/VN_PointCloudCompletion.

Host side: Python on PyTorch-ROCm mirroring the reference's plugin surface
(``PCNNet(config, enc_type, dec_type)``, the VN layer classes, ``ChamferDistance``, ``main.py train|test``).
Device side: hand-written gfx950 HIP kernels behind the C-ABI of ``include/vnpcc.h`` (``csrc/``).
"""
__version__ = "0.1.0"
