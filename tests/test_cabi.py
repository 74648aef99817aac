"""The C-ABI library loads on a machine without a GPU and exports exactly what include/*.h
declares (no compute calls here)."""
import glob
import os
import re

import pytest

from dspnet_amd import _lib
from dspnet_amd import functional  # noqa: F401  (registers the include/dspn_nn.h signatures)
from dspnet_amd.detect import nms as _nms_front  # noqa: F401  (registers include/dspn_nms.h)
from dspnet_amd.dataset import iterator as _iter_front  # noqa: F401  (registers include/dspn_augment.h)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(dspn_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_exists_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "build with __graft_entry__.build()"
    lib = _lib.lib()
    assert lib.dspn_abi_version() >= 1
    assert lib.dspn_last_error() is not None


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.lib()
    decl = declared_symbols()
    assert len(decl) >= 8
    for name in sorted(decl):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in dspnet_amd/_lib.py"
    for name in _lib.SIGNATURES:
        assert name in decl, f"{name} bound in _lib.py but not declared in include/*.h"


def test_workspace_queries_are_pure():
    lib = _lib.lib()
    assert lib.dspn_multibox_target_workspace_bytes(32, 6132, 200) > 32 * 6132 * 13
    assert lib.dspn_multibox_detection_workspace_bytes(2, 6132) > 2 * 6132 * 7 * 4
    assert lib.dspn_multibox_target_workspace_bytes(0, 10, 10) == 0


def test_argument_validation_without_gpu():
    """shape/attribute checks happen before any HIP call and carry the reference's texts"""
    import ctypes
    lib = _lib.lib()
    rc = lib.dspn_multibox_prior_f32(_lib.floats([0.1]), 1, _lib.floats([1.0]), 1, 0, 4,
                                     -1.0, -1.0, 0.5, 0.5, 0, ctypes.c_void_p(16), None)
    assert rc == -1 and b"Input height should > 0" in lib.dspn_last_error()
    rc = lib.dspn_multibox_target_f32(16, 16, 16, 1, 4, 3, 5, 3, 0.5, -1.0, 3.0, 0.5, 0,
                                      _lib.floats([.1, .1, .2, .2]), 16, 16, 16, 16, 1 << 20, None)
    assert rc == -1 and b"Label width should be 6" in lib.dspn_last_error()


def test_operator_front_end_rejects_cpu_tensors():
    import torch
    from dspnet_amd import operator as op
    with pytest.raises(_lib.DspnError):
        op.MultiBoxTarget(torch.zeros(1, 4, 4), torch.zeros(1, 2, 6), torch.zeros(1, 3, 4))
    with pytest.raises(_lib.DspnError, match="Label width should be 6"):
        op.MultiBoxTarget(torch.zeros(1, 4, 4), torch.zeros(1, 2, 5), torch.zeros(1, 3, 4))
    with pytest.raises(_lib.DspnError, match="anchors mismatch"):
        op.MultiBoxDetection(torch.zeros(1, 3, 4), torch.zeros(1, 19), torch.zeros(1, 4, 4))


def test_convolution_argument_validation_without_gpu():
    """the graph kernels' entry points reject bad geometry / math modes / unpaired affine vectors before any HIP call"""
    import ctypes
    lib = _lib.lib()
    p = ctypes.c_void_p(256)        # never dereferenced: every call below fails its checks first

    def fwd(N=1, H=8, W=8, Cin=4, math=0, in_scale=None, in_shift=None):
        return lib.dspn_conv2d_forward_bn_f32(p, in_scale, in_shift, 0, p, None, None, None, p, N, H, W, Cin, 8, 3, 3, 1, 1, 1, 1,
                                              8, 8, 8 * 8 * 8, 8, 0, 0, None, 0, None, math, None, None, None, 0, None)

    assert fwd(N=0) == -1 and b"bad geometry" in lib.dspn_last_error()
    assert fwd(math=4) == -1 and b"DSPN_MATH" in lib.dspn_last_error()
    assert fwd(math=-1) == -1
    assert fwd(in_scale=p) == -1 and b"go together" in lib.dspn_last_error()
    for math in (0, 1, 2, 3):       # DSPN_MATH_FP32, DSPN_MATH_BF16, DSPN_MATH_F32_BF16X3, DSPN_MATH_F32_F16X2 are the accepted values
        assert fwd(N=0, math=math) == -1 and b"bad geometry" in lib.dspn_last_error()
    # the split math reads whole 32-channel blocks from piece planes: without them the call is rejected (before any launch)
    assert fwd(Cin=64, math=2) == -1 and b"piece planes" in lib.dspn_last_error()
    assert fwd(Cin=64, math=3) == -1
    # round 4: the two-piece math without the operands' magnitude blocks is an error, not a silent scale of 1 ...
    assert fwd(math=3) == -1 and b"DSPN_MATH_UNSCALED_OK" in lib.dspn_last_error()
    # ... unless the caller vouches for |operand| < 65504 (the call then gets as far as the next check)
    assert fwd(Cin=64, math=3 | 0x100) == -1 and b"piece planes" in lib.dspn_last_error()
    assert fwd(math=4 | 0x100) == -1 and b"DSPN_MATH" in lib.dspn_last_error()
    assert lib.dspn_conv2d_weight_planes_f32(p, p, None, 8, 9, 64, 0, 2, None, None) == -1 and b"magnitude" in lib.dspn_last_error()
    assert lib.dspn_conv2d_weight_planes_f32(p, p, None, 8, 9, 64, 0, 4, None, None) == -1     # pieces: 2 or 3
    assert lib.dspn_conv2d_weight_planes_f32(p, p, None, 8, 9, 48, 0, 3, None, None) == -1     # forward planes: Cin % 32
    assert lib.dspn_conv2d_weight_planes_f32(p, None, p, 8, 9, 64, 48, 3, None, None) == -1    # transposed planes: cols_t % 32
    assert lib.dspn_conv2d_weight_planes_f32(p, None, p, 40, 9, 64, 32, 3, None, None) == -1   # ... and cols_t >= Cout
    assert lib.dspn_conv2d_weight_planes_f32(p, None, None, 8, 9, 64, 0, 3, None, None) == -1  # nothing to write
    assert lib.dspn_absmax_f32(p, 8, 6, None, None, 0, p, None) == -1                # C % 4
    assert lib.dspn_absmax_f32(p, 8, 8, p, None, 0, p, None) == -1                   # scale without shift
    # BatchNorm finalize from tile statistics: the per-tile extremes and the magnitude block they feed go together
    import ctypes
    f = ctypes.c_float(1e-5)
    assert lib.dspn_bn_stats_from_tiles_f32(p, 4, 128, 512, 8, f, None, p, p, p, p, p, p, 0, None, None, None, None, 0, None) == -1
    assert b"go together" in lib.dspn_last_error()
    assert lib.dspn_bn_stats_from_tiles_f32(p, 4, 128, 9999, 8, f, None, p, p, p, p, p, None, 0, None, None, None, None, 0, None) == -1   # rows vs tiles
    # round 4: gradients as fp16 piece planes -- the flag needs the two-piece math, whole 32-channel blocks and the block the
    # planes were cut by; the BatchNorm backward that writes them needs what it bounds dx from
    def dgrad(math, ldy=64, dy_absmax=p):
        return lib.dspn_conv2d_dgrad_bn_f32(p, p, None, p, 1, 8, 8, 32, ldy, 3, 3, 1, 1, 1, 1, 8, 8, 32, 0, None, None, None, None,
                                            None, 0, None, 0, None, math, dy_absmax, p, None, 0, None)
    assert dgrad(0 | 0x200) == -1 and b"DSPN_MATH_DY_PLANES" in lib.dspn_last_error()       # not the two-piece math
    assert dgrad(3 | 0x200, ldy=48) == -1 and b"DSPN_MATH_DY_PLANES" in lib.dspn_last_error()
    assert dgrad(3 | 0x200, dy_absmax=None) == -1
    assert lib.dspn_bn_backward_from_sums_f32(p, p, p, p, p, p, None, p, 4, p, None, p, 512, 48, 1, 0, p, None, p, p, 1, p, 1 << 20, None) == -1
    assert b"piece planes" in lib.dspn_last_error()                                         # C % 32
    assert lib.dspn_bn_backward_from_sums_f32(p, p, p, p, p, p, None, p, 4, p, None, p, 512, 64, 1, 1, p, None, p, p, 1, p, 1 << 20, None) == -1   # accumulate
    assert lib.dspn_bn_backward_from_sums_f32(p, p, p, p, p, p, None, p, 4, p, None, p, 512, 64, 1, 0, p, None, None, p, 1, p, 1 << 20, None) == -1  # no dy_absmax
    # ... and the INPUT as piece planes (written by dspn_bn_apply_planes_f32): two-piece math, Cin % 32, no affine, its block
    assert fwd(Cin=64, math=0 | 0x400) == -1 and b"DSPN_MATH_X_PLANES" in lib.dspn_last_error()
    assert fwd(Cin=48, math=3 | 0x400) == -1 and b"DSPN_MATH_X_PLANES" in lib.dspn_last_error()
    assert fwd(Cin=64, math=3 | 0x400) == -1 and b"DSPN_MATH_X_PLANES" in lib.dspn_last_error()      # no x_absmax
    def wgrad(math, Cin=64, x_absmax=p, in_scale=None):
        return lib.dspn_conv2d_wgrad_bn_f32(p, in_scale, in_scale, 0, p, p, 1, 8, 8, Cin, 64, 64, 3, 3, 1, 1, 1, 1, 8, 8, 0, math,
                                            x_absmax, p, None, 0, None)
    assert wgrad(2 | 0x400) == -1 and b"DSPN_MATH_X_PLANES" in lib.dspn_last_error()
    assert wgrad(3 | 0x400, Cin=48) == -1 and b"DSPN_MATH_X_PLANES" in lib.dspn_last_error()
    assert wgrad(3 | 0x400, x_absmax=None) == -1 and b"DSPN_MATH_X_PLANES" in lib.dspn_last_error()
    assert wgrad(3 | 0x400, in_scale=p) == -1 and b"DSPN_MATH_X_PLANES" in lib.dspn_last_error()
    assert lib.dspn_bn_apply_planes_f32(p, p, p, p, 64, 48, 1, p, None) == -1 and b"multiple of 32" in lib.dspn_last_error()
    assert lib.dspn_bn_apply_planes_f32(p, p, p, p, 64, 64, 1, None, None) == -1      # the block the planes are cut by
    assert lib.dspn_bn_apply_planes_f32(p, p, p, ctypes.c_void_p(256), 64, 64, 1, p, None) == -1 and b"in place" in lib.dspn_last_error()
    assert lib.dspn_copy_block_batch_f32(None, 6, 100, None) == -1 and lib.dspn_copy_block_batch_f32(p, 0, 100, None) == -1
    assert lib.dspn_bn_backward_maxpool_f32(p, p, p, p, p, 1, 8, 8, 6, 3, 2, 1, 4, 4, p, p, None, p, None, p, 1, None, p, 1 << 20, None) == -1   # C % 4
    assert lib.dspn_bn_backward_maxpool_f32(p, p, p, p, None, 1, 8, 8, 8, 3, 2, 1, 4, 4, p, p, None, p, None, p, 1, None, p, 1 << 20, None) == -1   # no argmax record
    assert lib.dspn_bn_backward_maxpool_f32(p, p, p, p, p, 1, 8, 8, 8, 3, 2, 1, 4, 4, p, p, None, p, None, p, 1, None, p, 16, None) != 0 and b"workspace" in lib.dspn_last_error()
    # batched weight transposes count 32 x 32 tiles of a tap
    assert lib.dspn_conv2d_weight_transpose_tiles(64, 9, 64, 64) == 2 * 9 * 2 and lib.dspn_conv2d_weight_transpose_tiles(19, 1, 128, 24) == 4
    assert lib.dspn_conv2d_weight_transpose_tiles(64, 9, 64, 32) == 0                # Cout_pad < Cout
    assert lib.dspn_conv2d_weight_transpose_batch_f32(None, 1, 4, None) == -1
    assert lib.dspn_conv2d_weight_planes_tiles(64, 9, 64, 64, 1) == 2 * 9 * 2 and lib.dspn_conv2d_weight_planes_tiles(40, 1, 64, 64, 1) == 4


def test_oracle_is_clean_under_address_and_ub_sanitizers():
    """SURVEY.md section 5 (sanitizers on the host build): the C oracle's three operators over exactly-sized heap buffers and the
    degenerate inputs of the GPU tests, built with -fsanitize=address,undefined (oracle/sanitize_driver.c)"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "sanitize"], capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and "LeakSanitizer has encountered a fatal error" in r.stderr:
        # leak checking needs ptrace, which some sandboxes deny: the address / UB checks still run without it
        r = subprocess.run([os.path.join(ROOT, "oracle", "_build", "sanitize_driver")], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and "sanitize_driver: ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
