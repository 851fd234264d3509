"""CPU check of the fp8 mode's host-side weight quantiser (ivr_quantize_e4m3_host, used by ivr_tower_finalize) against
torch's float8_e4m3fn conversion: both implement OCP e4m3 with round-to-nearest-even; the build's quantiser saturates
where torch produces NaN for out-of-range inputs, so the comparison is made on clamped values and saturation is checked apart."""
import ctypes as C

import numpy as np
import torch

from ivr_amd import _ffi


def _quant(x):
    lib = _ffi.load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape, dtype=np.uint8)
    _ffi.check(lib.ivr_quantize_e4m3_host(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), x.size), "quantize")
    return out


def test_every_e4m3_value_round_trips():
    codes = np.arange(256, dtype=np.uint8)
    vals = torch.from_numpy(codes).view(torch.float8_e4m3fn).float().numpy()
    finite = np.isfinite(vals)
    got = _quant(vals[finite])
    want = codes[finite].copy()
    want[want == 0x80] = 0x80                       # -0 keeps its sign
    assert np.array_equal(got, want)


def test_matches_torch_rounding_on_dense_and_tie_inputs():
    rng = np.random.default_rng(0)
    x = np.concatenate([
        rng.standard_normal(200_000).astype(np.float32) * 3,
        (rng.standard_normal(100_000) * 1e-2).astype(np.float32),               # subnormal range (< 2^-6)
        rng.uniform(-448, 448, 100_000).astype(np.float32),
    ])
    # exact ties between neighbouring e4m3 values (midpoints): round-to-even must agree
    codes = np.arange(0, 0x7e, dtype=np.uint8)
    lo = torch.from_numpy(codes).view(torch.float8_e4m3fn).float().numpy()
    hi = torch.from_numpy(codes + 1).view(torch.float8_e4m3fn).float().numpy()
    mid = ((lo.astype(np.float64) + hi.astype(np.float64)) / 2).astype(np.float32)
    x = np.concatenate([x, mid, -mid, np.nextafter(mid, np.float32(np.inf)), np.nextafter(mid, np.float32(-np.inf))])
    want = torch.from_numpy(np.clip(x, -448, 448)).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    got = _quant(x)
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]


def test_saturation_and_nan():
    got = _quant(np.array([449, 1e9, np.inf, -449, -np.inf, 464, 479.9], dtype=np.float32))
    assert got.tolist() == [0x7e, 0x7e, 0x7e, 0xfe, 0xfe, 0x7e, 0x7e]
    assert (_quant(np.array([np.nan], dtype=np.float32))[0] & 0x7f) == 0x7f
    assert _quant(np.array([0.0, -0.0, 2.0 ** -10, 2.0 ** -10 * 1.0001, 2.0 ** -9], dtype=np.float32)).tolist() == [0, 0x80, 0, 1, 1]
