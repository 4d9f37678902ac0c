"""The 256x256 convolution tile with its PACKED weight image (round 5: csrc/conv_igemm256.hip pack256_kernel,
emp_conv256_pack_weights, variant bit 20): every 1 KiB piece an LDS-DMA instruction moves is contiguous in the image, in the
kernel's K-walk order, so the instruction fetches whole 128-byte lines (profiles/r05_conv256_requests.txt).  It is a re-layout:
the results must be bit-identical to the plain weights on the same tile, for the shapes the network sends there -- 1x1
(one K-walk group), dilated 3x3 (groups of 8 slabs, tap-major inside a group), residual, per-image bias, the K-concatenated
second source (conv3 + projection shortcut).  The same tile form also takes the plain launches of the network:
tests/test_gpu_parity_fullsize.py teacher-forces every layer at 1024^2 with the packed images on (the default) and
tests/test_gpu_model.py compares whole forwards.
Reference semantics: nn.Conv2d + folded BatchNorm + ReLU (+ skip add), /root/reference/empanada/models/encoders/resnet.py:109-129."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TILE256 = 4 << 4
PACKED = 1 << 20


def _conv(lib, _abi, x, w, b, out, k, pad, dil, relu, variant, res=None, bias_n=None):
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    _abi.check(lib.emp_conv2d_nhwc_f16(_abi.ptr(x), B, H, W, Cin, Cin, _abi.ptr(w), _abi.ptr(b),
                                       _abi.ptr(bias_n) if bias_n is not None else None,
                                       _abi.ptr(res) if res is not None else None, Cout if res is not None else 0,
                                       _abi.ptr(out), Cout, Cout, k, k, 1, pad, dil, relu, variant,
                                       _abi.stream_ptr(x.device)), 'conv')


@pytest.mark.parametrize('name,H,Cin,Cout,k,dil,res', [
    ('l3.conv1', 32, 1024, 256, 1, 1, False),
    ('l4.conv3 + skip', 32, 512, 2048, 1, 1, True),
    ('l3.conv2', 32, 256, 256, 3, 1, False),
    ('aspp d4', 32, 2048, 256, 3, 4, False),
    ('l4.conv2 d2', 32, 512, 512, 3, 2, False),
    ('odd slab count', 32, 320, 256, 3, 1, False),       # CB = 10: not a multiple of the walk group -> one group
])
def test_packed_weights_give_the_same_bits(name, H, Cin, Cout, k, dil, res):
    from empanada_napari_amd import _abi
    lib = _abi.load()
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(hash(name) % 1000)
    B = 24                                                    # 24 x 32 x 32 = 96 pixel tiles x Cout / 256
    x = torch.randn((B, H, H, Cin), generator=g).to(torch.float16).to(dev)
    w = (torch.randn((Cout, k * k, Cin), generator=g) / np.sqrt(Cin * k * k)).to(torch.float16).to(dev)
    b = torch.randn((Cout,), generator=g).to(dev)
    bn = torch.randn((B, Cout), generator=g).to(dev)
    r = torch.randn((B, H, H, Cout), generator=g).to(torch.float16).to(dev) if res else None
    pad = dil * (k // 2)
    plain = torch.empty((B, H, H, Cout), device=dev, dtype=torch.float16)
    packed_out = torch.full((B, H, H, Cout), float('nan'), device=dev, dtype=torch.float16)
    wp = torch.empty_like(w)
    _abi.check(lib.emp_conv256_pack_weights(_abi.ptr(w), _abi.ptr(wp), Cout, k * k, Cin, 0, _abi.stream_ptr(dev)), 'pack')
    _conv(lib, _abi, x, w, b, plain, k, pad, dil, 1, TILE256, res=r, bias_n=bn)
    _conv(lib, _abi, x, wp, b, packed_out, k, pad, dil, 1, TILE256 | PACKED, res=r, bias_n=bn)
    torch.cuda.synchronize()
    assert torch.isfinite(packed_out.float()).all()
    assert torch.equal(plain, packed_out), f'{name}: {(plain != packed_out).sum().item()} of {plain.numel()} values differ'
    # the image is a permutation of the weights: same multiset of values
    assert torch.equal(torch.sort(w.flatten().view(torch.int16))[0], torch.sort(wp.flatten().view(torch.int16))[0])
    # and the plain tile against the generic 128 x 128 tile (the anchor the other parity tests pin on the oracle)
    ref = torch.empty_like(plain)
    _conv(lib, _abi, x, w, b, ref, k, pad, dil, 1, (1 << 4) | 3, res=r, bias_n=bn)
    torch.cuda.synchronize()
    assert torch.equal(plain, ref)


def test_packed_flag_is_refused_where_the_tile_cannot_run():
    from empanada_napari_amd import _abi
    lib = _abi.load()
    dev = torch.device('cuda:0')
    x = torch.zeros((1, 16, 16, 64), device=dev, dtype=torch.float16)
    w = torch.zeros((64, 1, 64), device=dev, dtype=torch.float16)
    out = torch.empty((1, 16, 16, 64), device=dev, dtype=torch.float16)
    rc = lib.emp_conv2d_nhwc_f16(_abi.ptr(x), 1, 16, 16, 64, 64, _abi.ptr(w), None, None, None, 0, _abi.ptr(out), 64, 64,
                                 1, 1, 1, 0, 1, 0, PACKED, _abi.stream_ptr(dev))
    assert rc != 0 and b'packed' in lib.emp_last_error()
    rc = lib.emp_conv256_pack_weights(_abi.ptr(w), _abi.ptr(w), 64, 1, 64, 0, _abi.stream_ptr(dev))
    assert rc != 0


@pytest.mark.parametrize('N,H,Cin,H2,Cin2,s2,Cout', [
    (24, 32, 512, 64, 1024, 2, 2048),      # layer4.0 conv3 + projection shortcut (stride-2 second source)
    (24, 32, 128, 32, 256, 1, 512),        # layer2.0's shape: 4 + 8 K-tiles
    (24, 32, 64, 32, 64, 1, 256),          # 4 K-tiles: too short for the pair pipeline -> the 64-byte-row kernel, packed weights all the same
])
def test_packed_weights_with_a_second_source(N, H, Cin, H2, Cin2, s2, Cout):
    """conv3 + projection shortcut as one K-concatenated 1x1 conv (ConvParams::in2): the packed image carries the second
    source's channels behind the main source's, the whole-line kernel switches source at a pair boundary."""
    from empanada_napari_amd import _abi
    lib = _abi.load()
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(Cin + Cin2)
    x = torch.randn((N, H, H, Cin), generator=g).to(torch.float16).to(dev)
    x2 = torch.randn((N, H2, H2, Cin2), generator=g).to(torch.float16).to(dev)
    w = (torch.randn((Cout, Cin + Cin2), generator=g) / np.sqrt(Cin + Cin2)).to(torch.float16).to(dev)
    b = torch.randn((Cout,), generator=g).to(dev)
    wp = torch.empty_like(w)
    _abi.check(lib.emp_conv256_pack_weights(_abi.ptr(w), _abi.ptr(wp), Cout, 1, Cin, Cin2, _abi.stream_ptr(dev)), 'pack')

    def run(wt, variant):
        out = torch.full((N, H, H, Cout), float('nan'), device=dev, dtype=torch.float16)
        _abi.check(lib.emp_conv1x1_dual_nhwc_f16(_abi.ptr(x), N, H, H, Cin, Cin, _abi.ptr(x2), H2, H2, Cin2, Cin2, s2,
                                                 _abi.ptr(wt), _abi.ptr(b), _abi.ptr(out), Cout, Cout, 1, variant,
                                                 _abi.stream_ptr(dev)), 'dual')
        torch.cuda.synchronize()
        return out

    plain = run(w, TILE256)
    assert torch.isfinite(plain.float()).all()
    assert torch.equal(plain, run(wp, TILE256 | PACKED))
    assert torch.equal(plain, run(w, (1 << 4) | 3))          # the 128 x 128 tile
