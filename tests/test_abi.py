"""The C-ABI library builds, loads and exports every symbol include/refinenet_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

from conftest import ROOT
from hipvsr import lib as L


def _declared():
    text = open(os.path.join(ROOT, 'include', 'refinenet_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rnh_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(L.LIB_PATH), 'run csrc/build.sh or __graft_entry__.build()'
    lib = ctypes.CDLL(L.LIB_PATH)
    names = _declared()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(L.EXPORTS) == names


def test_binding_struct_layout_and_version():
    lib = L.load()
    assert lib.rnh_abi_version() == L.ABI_VERSION == 7
    sizes = (ctypes.c_int32 * 4)()
    lib.rnh_struct_sizes(ctypes.byref(sizes))
    assert list(sizes) == [ctypes.sizeof(L.Src), ctypes.sizeof(L.Dst), ctypes.sizeof(L.ConvArgs), ctypes.sizeof(L.WgradArgs)]
    assert ctypes.sizeof(L.Src) == 48 and ctypes.sizeof(L.Dst) == 32


def test_argument_errors_are_reported_without_a_gpu():
    lib = L.load()
    a = L.ConvArgs()
    assert lib.rnh_conv_igemm(ctypes.byref(a), None) != 0          # nsrc == 0 -> RNH_E_RANGE before any launch
    assert b'nsrc' in lib.rnh_last_error()
    assert lib.rnh_ew_add(None, None, None, None, 4, 0, None) == -1


def test_product_has_no_cpu_path():
    import pytest
    import torch
    from hipvsr.hip_ops import HipOps
    with pytest.raises(L.HipKernelError):
        HipOps('cpu')
    from src.model.nets import RefineNet
    net = RefineNet(1, 1, [8, 8], num_stages=2, update_memory=True, num_updated_frames=2, positional_encoding=True)
    with pytest.raises(RuntimeError, match='HIP device'):
        net([torch.zeros(1, 1, 4, 4)] * 6, torch.zeros(1, 6, 1))


import pytest  # noqa: E402


@pytest.mark.gpu
def test_library_loaded_before_torch_still_sees_the_gpu():
    """build() loads the library before anything imported torch: the process must still end up with ONE HIP runtime
    (a second one reports 'no ROCm-capable device' at the first launch).  Run in a child so that the import order is ours."""
    import subprocess
    import sys
    from conftest import PKG
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "from hipvsr import lib as L\nL.load()\n"
            "import torch\nfrom hipvsr.step_tail import psnr_ssim\n"
            "x = torch.rand(2, 16, 16, device='cuda:0')\n"
            "r = psnr_ssim(x, x, 2, 1, 16, 16)\ntorch.cuda.synchronize()\nprint('ok', float(r[1]))\n") % (ROOT, PKG)
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith('ok 1.0'), out.stderr[-2000:]
