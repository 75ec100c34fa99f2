"""Reference restatement of the f16q operand format (chatts_amd/csrc/f16q.h) in torch - TEST INFRASTRUCTURE.

x = hi + lo,  hi = f16_rne(clamp(x, +-65504)),  lo = x - hi (exact in float32), stored as e4m3 codes q = rne(lo * 2^-E) with one e8m0 byte
(E + 127) per row and 128 consecutive K-values: the smallest power of two with max|lo| / 2^E <= 448 (127 for an all-zero block).  Weights:
f16 copy + e4m3 copy with ONE such scale per row.  Every function works on CPU or GPU tensors; comparisons against the HIP producers are
BIT-exact (codes, scale bytes, f16 bits), the GEMM is compared against the float64 product of the dequantised operands.
No reference twin: the format is this library's own (NetManAIOps/ChatTS multiplies fp16 x fp16 inside vLLM kernels, NOT IN REFERENCE)."""
import torch

BLOCK = 128


def scale_exp(amax):
    """E of the block scale 2^E for largest |residual| amax (float32 tensor): f16q_exp of f16q.h, by the bits"""
    b = amax.contiguous().view(torch.int32)
    e = (b >> 23) - 127
    E = e - 8 + ((b & 0x7FFFFF) > 0x600000).to(torch.int32)
    E = torch.where(b == 0, torch.zeros_like(E), E)
    return E.clamp(min=-127)


def e4m3_codes(y):
    """RNE, saturating (|y| <= 448 here) float32 -> OCP e4m3fn codes (uint8)"""
    return y.to(torch.float8_e4m3fn).view(torch.uint8)


def e4m3_values(codes):
    return codes.view(torch.float8_e4m3fn).to(torch.float32)


def split(x):
    """float32 [M, K] -> (hi float16 [M, K], lo8 uint8 [M, K], scale uint8 [M, K / 128])"""
    M, K = x.shape
    assert K % BLOCK == 0
    hi = x.clamp(-65504.0, 65504.0).to(torch.float16)
    lo = x - hi.float()
    amax = lo.abs().view(M, K // BLOCK, BLOCK).amax(-1)
    E = scale_exp(amax)
    inv = torch.exp2(-E.float())
    q = e4m3_codes((lo.view(M, K // BLOCK, BLOCK) * inv[:, :, None]).reshape(M, K))
    return hi, q, (E + 127).to(torch.uint8)


def weights(w):
    """bf16 [N, K] -> (w16 float16, w8 uint8, w8_exp uint8 [N])"""
    wf = w.float()
    E = scale_exp(wf.abs().amax(-1))
    q = e4m3_codes(wf * torch.exp2(-E.float())[:, None])
    return wf.to(torch.float16), q, (E + 127).to(torch.uint8)


def dequant_planes(hi, q, sc):
    """the float64 value the planes encode"""
    M, K = hi.shape
    s = torch.exp2(sc.to(torch.float64) - 127.0)
    lo = (e4m3_values(q).to(torch.float64).view(M, K // BLOCK, BLOCK) * s[:, :, None]).reshape(M, K)
    return hi.to(torch.float64), lo


def gemm(hi, q, sc, w16, w8, w8e):
    """float64 product of the dequantised operands: hi . w16^T + lo . w8^T (what gemm_f16q accumulates, without its float32 rounding)"""
    h, lo = dequant_planes(hi, q, sc)
    w8v = e4m3_values(w8).to(torch.float64) * torch.exp2(w8e.to(torch.float64) - 127.0)[:, None]
    return h @ w16.to(torch.float64).t() + lo @ w8v.t()
