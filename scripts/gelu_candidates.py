"""VERDICT r04 item 5 — the exact-erf GELU epilogue of fc1 (csrc/gemm.hip.h gelu_erf_core: ~100 issue cycles per output pair, ~18 % of an fc1
launch) against cheaper formulations, EXHAUSTIVELY: the epilogue's input is one of the 65 536 bf16 values and its output a bf16, so a candidate
is acceptable iff it reproduces torch's GELU (the reference's arithmetic under CPU autocast: F.gelu on a bf16 tensor = erf form evaluated in
fp32, rounded to bf16) on EVERY input.  Each candidate is emulated in fp32 (fma = exactly rounded a * b + c; v_exp_f32 / v_rcp_f32 taken as
correctly rounded — the hardware's are within 1 ulp, so a candidate that passes here still has to pass the exhaustive GPU test,
tests/test_gpu_parity.py::test_gelu_epilogue_is_exact_on_every_bf16_value) and priced in VALU issue cycles per PAIR of outputs
(MI355X_MICROARCH.md: v_exp / v_rcp 8, fma / mul / add / max 4 — packed f32 forms 4 for the pair —, v_cvt_pk_bf16_f32 ~4.5, bit ops 4).

  python3 scripts/gelu_candidates.py  ->  a table on stdout (profiles/r05_i_gelu_candidates.txt)"""
import numpy as np
import torch

f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + np.asarray(c, dtype=np.float64)).astype(f32)


def all_bf16():
    bits = np.arange(65536, dtype=np.uint32) << 16
    x = bits.view(f32)
    return x


def to_bf16_bits(y):
    t = torch.from_numpy(np.ascontiguousarray(y)).to(torch.bfloat16)
    return t.view(torch.int16).numpy().astype(np.int32) & 0xffff


def reference_bits(x):
    t = torch.from_numpy(x.copy()).to(torch.bfloat16)
    return torch.nn.functional.gelu(t).view(torch.int16).numpy().astype(np.int32) & 0xffff


# ---------------------------------------------------------------------------------------------- candidates (x: fp32 array of bf16 values)
def cand_current(x):
    """csrc/gemm.hip.h gelu_erf_core: erfc by Abramowitz & Stegun 7.1.26 on u = |x| sqrt(log2 e / 2), halved coefficients."""
    ax = np.abs(x)
    u = (ax * f32(0.84932180028801904272)).astype(f32)
    d = fma(u, f32(0.27273748087922250), f32(1.0))
    t = (f32(1.0) / d).astype(f32)
    p = fma(t, f32(0.5307027145), f32(-0.7265760135))
    p = fma(t, p, f32(0.7107068705)); p = fma(t, p, f32(-0.142248368)); p = fma(t, p, f32(0.127414796))
    p = (p * t).astype(f32)
    a = (u * u).astype(f32)
    e = np.exp2(-a.astype(np.float64)).astype(f32)
    h = (p * e).astype(f32)
    return fma(-ax, h, np.maximum(x, f32(0)))


COST_CURRENT = 10 * 4 + 2 * 8 + 2 * 8 + 2 * 4 + 4 * 4 + 9          # 10 packed ops, 2 rcp, 2 exp, 2 max, 4 unpack bit ops, 2 cvt_pk


def cand_exp_of_poly(x, deg):
    """Phi(-|x|) = exp2(P(|x|)) with P a least-squares polynomial of log2 Phi(-t) on [0, 6.5] (one transcendental instead of two, no division)."""
    from math import erfc, log2, sqrt
    ts = np.linspace(0, 6.5, 4001)
    ys = np.array([log2(0.5 * erfc(t / sqrt(2))) for t in ts])
    coef = np.polynomial.chebyshev.Chebyshev.fit(ts, ys, deg).convert(kind=np.polynomial.Polynomial).coef.astype(f32)
    ax = np.minimum(np.abs(x), f32(6.5))
    p = np.full_like(ax, coef[-1])
    for c in coef[-2::-1]:
        p = fma(p, ax, f32(c))
    h = np.exp2(p.astype(np.float64)).astype(f32)
    return fma(-np.abs(x), h, np.maximum(x, f32(0)))


def cost_exp_of_poly(deg):
    return deg * 4 + 2 * 8 + 2 * 4 + 4 + 4 * 4 + 9 + 4            # Horner (packed), 2 exp, 2 max, final fma, unpack, cvt, clamp


def cand_erf_poly_core(x, deg, cut):
    """erf(z) by an odd polynomial for |z| <= cut (no transcendental there), the exp form beyond: both branches are evaluated for a whole
    wave when its lanes disagree, so the cost is the SUM unless a tile's values all fall on one side — priced as the sum."""
    z = (x * f32(0.70710678)).astype(f32)
    zs = np.linspace(0, cut, 2001)
    from math import erf
    ys = np.array([erf(v) / v if v > 0 else 2 / np.sqrt(np.pi) for v in zs])
    coef = np.polynomial.chebyshev.Chebyshev.fit(zs * zs, ys, deg).convert(kind=np.polynomial.Polynomial).coef.astype(f32)
    z2 = (z * z).astype(f32)
    p = np.full_like(z, coef[-1])
    for c in coef[-2::-1]:
        p = fma(p, z2, f32(c))
    erf_core = (p * z).astype(f32)
    core = fma((x * f32(0.5)).astype(f32), erf_core, (x * f32(0.5)).astype(f32))
    return np.where(np.abs(z) <= f32(cut), core, cand_current(x))


def cand_tanh_form(x):
    """The cheap form AMD's library fuses (tanh approximation) — for the record: it is a different function."""
    y = (f32(0.7978845608) * (x + f32(0.044715) * x * x * x)).astype(f32)
    return (f32(0.5) * x * (f32(1.0) + np.tanh(y.astype(np.float64)).astype(f32))).astype(f32)


def cand_fp32_erf(x):
    """What torch computes: x/2 (1 + erf(x / sqrt 2)) in fp32 with a correctly rounded erf (the yardstick of the emulation itself)."""
    from scipy.special import erf
    return (f32(0.5) * x * (f32(1.0) + erf((x * f32(0.7071067811865476)).astype(np.float64)).astype(f32))).astype(f32)


def main():
    x = all_bf16()
    finite = np.isfinite(x)
    ref = reference_bits(x)
    rows = []
    def add(name, y, cost):
        got = to_bf16_bits(y)
        bad = (got != ref) & finite
        # -0.0 vs +0.0 and the NaN payload are not arithmetic differences the pipeline can see? they ARE bits: counted separately
        zero_sign = bad & ((got & 0x7fff) == 0) & ((ref & 0x7fff) == 0)
        real = bad & ~zero_sign
        worst = ""
        if real.any():
            i = np.nonzero(real)[0]
            j = i[np.argmax(np.abs(x[i]))]
            worst = f"e.g. x = {x[j]:.6g}: got {got[j]:#06x}, torch {ref[j]:#06x}"
        rows.append((name, cost, int(real.sum()), int(zero_sign.sum()), worst))
    add("current (A&S 7.1.26 erfc, rcp + exp2)", cand_current(x), COST_CURRENT)
    add("fp32 erf form with a correctly rounded erf (emulation check)", cand_fp32_erf(x), 0)
    for deg in (5, 6, 7, 8, 9, 10, 12):
        add(f"exp2(poly_{deg}(|x|)) — one transcendental", cand_exp_of_poly(x, deg), cost_exp_of_poly(deg))
    for deg, cut in ((5, 1.0), (6, 1.5), (7, 2.0), (9, 2.5)):
        add(f"odd erf polynomial deg {2 * deg + 1} for |z| <= {cut}, exp form beyond (both evaluated)", cand_erf_poly_core(x, deg, cut),
            COST_CURRENT + (deg + 3) * 4)
    add("tanh approximation (what hipBLASLt fuses)", cand_tanh_form(x), 6 * 4 + 2 * 16 + 16 + 9)
    print(f"{'candidate':92s} {'cycles/pair':>11s} {'vs today':>8s} {'wrong of 65536':>14s} {'(+/-0 only)':>11s}")
    for name, cost, bad, zs, worst in rows:
        rel = f"{cost / COST_CURRENT:.2f}" if cost else "-"
        print(f"{name:92s} {cost:11d} {rel:>8s} {bad:14d} {zs:11d}  {worst}")
    print(f"\nacceptance (VERDICT r04 item 5): 0 wrong AND <= 0.60 of today's {COST_CURRENT} cycles per pair")


if __name__ == "__main__":
    main()
